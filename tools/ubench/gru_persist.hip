// Persistent K-split GRU step on gfx950: is one resident kernel for all T steps faster than two launches per step?
// (the round-3 verdict's item 5; the product's Scan runs the C5 step as two generated product+epilogue launches,
//  13.9 us/step, `profiles/r4z_c5_kernel_stats.md`).
//
// The recurrence (pytensor/scan/scan_perform.pyx:74-603 drives it one step at a time on the host):
//     z = sigmoid(pz_t + h Uz)   r = sigmoid(pr_t + h Ur)   c = tanh(pc_t + (r*h) Uc)   h' = (1-z) h + z c
// B = 64, H = 1024, fp32; p*_t are the hoisted input products (x_t W* + b*), read as data.
//
// Decomposition: 256 workgroups = 8 K-slices (128 rows of U) x 32 N-slices (32 columns), XCD-aware (workgroup id % 8 is
// the XCD, all 8 K-slices of one N-slice sit on ONE XCD).  A workgroup keeps its three 128x32 weight blocks in REGISTERS
// for all T steps (96 VGPRs per lane as MFMA B-fragments); per phase it reads a 64x128 slice of the left operand (32 KB),
// multiplies on v_mfma_f32_32x32x2_f32, folds its two K-halves through LDS and writes a 64x32 partial; the workgroup is
// also the OWNER of 8 batch rows x 32 columns: it sums the 8 partials of its N-slice in a fixed order, applies the gate
// and publishes.  Four hand-overs per step (h -> partial -> r*h -> partial -> h'), none with a flag: every exchanged word
// is born as the bit pattern 0xFFFFFFFF and the reader retries until none is left (agent-scope loads / stores); readers
// that are the only reader put the pattern back, the r*h ring is reset by its writer one step later.
//
// build: hipcc --offload-arch=gfx950 -O3 -o gru_persist gru_persist.hip ; run: ./gru_persist [T]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int B = 64, H = 1024, KS = 8, NS = 32, KW = H / KS, NW = H / NS;  // KW = 128, NW = 32
constexpr unsigned EMPTY = 0xFFFFFFFFu;
constexpr long long POLL_TICKS = 5000000;  // 50 ms of the 100 MHz wall clock: a hand-over that never arrives
constexpr int NPROF = 12;
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned ld32(const float* p) {
  return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld64(const float* p) {
  return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st32(float* p, float v) {
  __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st32u(float* p, unsigned v) {
  __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// the lane's 32 left-operand values of one phase: row `row` of a [64][1024] matrix, 32 consecutive columns from k0
__device__ __forceinline__ bool load_lhs(const float* M, int row, int k0, float (&a)[32], long long t_end, int* status) {
  const float* p = M + (long long)row * H + k0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const unsigned long long w = ld64(p + 2 * q);
      const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
      ok = ok && lo != EMPTY && hi != EMPTY;
      a[2 * q] = __uint_as_float(lo);
      a[2 * q + 1] = __uint_as_float(hi);
    }
    if (__all(ok)) return true;
    if ((long long)wall_clock64() > t_end) { atomicOr(status, 1); return false; }
  }
}

// the owner's sum of its N-slice's 8 partials, in producer order
__device__ __forceinline__ bool load_partials(const float* P, float& sum, long long t_end, int* status) {
  for (;;) {
    bool ok = true;
    unsigned w[KS];
#pragma unroll
    for (int kp = 0; kp < KS; kp++) { w[kp] = ld32(P + kp * 256); ok = ok && w[kp] != EMPTY; }
    if (__all(ok)) {
      float s = 0.f;
#pragma unroll
      for (int kp = 0; kp < KS; kp++) s += __uint_as_float(w[kp]);
      sum = s;
      return true;
    }
    if ((long long)wall_clock64() > t_end) { atomicOr(status, 2); return false; }
  }
}

// ---- variant 2: partials stay inside the XCD (its L2 is the meeting point: plain stores write through the CU's L1,
// the reader invalidates its L1 and loads without scope bits), 16-byte stores and loads, a third of the store count ----
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool load_lhs4(const float* M, int row, int k0, float (&a)[32], long long t_end, int* status) {
  const float* p = M + (long long)row * H + k0;
  for (;;) {
    f4v v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
        "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
        "global_load_dwordx4 %4, %8, off offset:64 sc1\n\t"
        "global_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
        "global_load_dwordx4 %6, %8, off offset:96 sc1\n\t"
        "global_load_dwordx4 %7, %8, off offset:112 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
        : "v"(p)
        : "memory");
    const f4v vs[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 8; q++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        a[4 * q + c] = vs[q][c];
        ok = ok && __float_as_uint(vs[q][c]) != EMPTY;
      }
    if (__all(ok)) return true;
    if ((long long)wall_clock64() > t_end) { atomicOr(status, 1); return false; }
  }
}
// ---- selective re-poll: only what came back empty is asked for again (a retry of the whole 32 KB slice by 256 workgroups is
// 8 MB of fabric reads per round) ----
__device__ __forceinline__ void issue_lhs8(const float* p, f4v (&v)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %8, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %8, off offset:48 sc1\n\t"
      "global_load_dwordx4 %4, %8, off offset:64 sc1\n\t"
      "global_load_dwordx4 %5, %8, off offset:80 sc1\n\t"
      "global_load_dwordx4 %6, %8, off offset:96 sc1\n\t"
      "global_load_dwordx4 %7, %8, off offset:112 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p)
      : "memory");
}
__device__ __forceinline__ bool quad_empty(const f4v& v) {
  return __float_as_uint(v[0]) == EMPTY || __float_as_uint(v[1]) == EMPTY || __float_as_uint(v[2]) == EMPTY || __float_as_uint(v[3]) == EMPTY;
}
__device__ __forceinline__ bool ensure_quad(const float* p, f4v& v, long long t_end, int* status) {
  for (;;) {
    const bool bad = quad_empty(v);
    if (!__any(bad)) return true;
    if (bad) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "+v"(v) : "v"(p) : "memory");
    if ((long long)wall_clock64() > t_end) { atomicOr(status, 1); return false; }
  }
}
__device__ __forceinline__ bool load_partials_sel(const float* P, float& sum, long long t_end, int* status) {
  unsigned w[KS];
#pragma unroll
  for (int kp = 0; kp < KS; kp++) w[kp] = ld32(P + kp * 256);
  for (;;) {
    bool bad = false;
#pragma unroll
    for (int kp = 0; kp < KS; kp++)
      if (w[kp] == EMPTY) { w[kp] = ld32(P + kp * 256); bad = bad || w[kp] == EMPTY; }
    if (!__any(bad)) break;
    if ((long long)wall_clock64() > t_end) { atomicOr(status, 2); return false; }
  }
  float s = 0.f;
#pragma unroll
  for (int kp = 0; kp < KS; kp++) s += __uint_as_float(w[kp]);
  sum = s;
  return true;
}

template <int PLD> __device__ __forceinline__ bool load_partials_l2(const float* P, float& sum, long long t_end, int* status) {
  if constexpr (PLD == 3) return false;
  const float* P2 = P + 4 * 256;
  for (;;) {
    unsigned w0, w1, w2, w3, w4, w5, w6, w7;
    if constexpr (PLD == 2)
      asm volatile(
        "global_load_dword %0, %8, off sc0\n\t"
        "global_load_dword %1, %8, off offset:1024 sc0\n\t"
        "global_load_dword %2, %8, off offset:2048 sc0\n\t"
        "global_load_dword %3, %8, off offset:3072 sc0\n\t"
        "global_load_dword %4, %9, off sc0\n\t"
        "global_load_dword %5, %9, off offset:1024 sc0\n\t"
        "global_load_dword %6, %9, off offset:2048 sc0\n\t"
        "global_load_dword %7, %9, off offset:3072 sc0\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "=&v"(w7)
        : "v"(P), "v"(P2)
        : "memory");
    else
    asm volatile(
        "buffer_inv sc0\n\t"
        "global_load_dword %0, %8, off\n\t"
        "global_load_dword %1, %8, off offset:1024\n\t"
        "global_load_dword %2, %8, off offset:2048\n\t"
        "global_load_dword %3, %8, off offset:3072\n\t"
        "global_load_dword %4, %9, off\n\t"
        "global_load_dword %5, %9, off offset:1024\n\t"
        "global_load_dword %6, %9, off offset:2048\n\t"
        "global_load_dword %7, %9, off offset:3072\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "=&v"(w7)
        : "v"(P), "v"(P2)
        : "memory");
    const bool ok = w0 != EMPTY && w1 != EMPTY && w2 != EMPTY && w3 != EMPTY && w4 != EMPTY && w5 != EMPTY && w6 != EMPTY && w7 != EMPTY;
    if (__all(ok)) {
      sum = ((((((__uint_as_float(w0) + __uint_as_float(w1)) + __uint_as_float(w2)) + __uint_as_float(w3)) + __uint_as_float(w4)) +
              __uint_as_float(w5)) + __uint_as_float(w6)) + __uint_as_float(w7);
      return true;
    }
    if ((long long)wall_clock64() > t_end) { atomicOr(status, 2); return false; }
  }
}
// the workgroup's folded 64x32 partial out of LDS: thread -> 4 consecutive columns of 2 rows, one 16-byte store each
template <int PST> __device__ __forceinline__ void st16(float* p, f4v v) {
  if constexpr (PST == 1) *(f4v*)p = v;                                                              // plain: lands in the XCD's L2
  else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(v) : "memory");  // agent scope (the nop: the
  // compiler's hazard recognizer does not see the 128-bit store inside the asm and may overwrite its data registers next cycle)
}
template <int PST> __device__ __forceinline__ void store_partial4(float* Pm, int ks, const float (*lo)[NW], const float (*hi)[NW], int tid) {
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int idx = tid + 256 * q, row = idx >> 3, c4 = (idx & 7) * 4;
    f4v v;
#pragma unroll
    for (int c = 0; c < 4; c++) v[c] = lo[row][c4 + c] + hi[row][c4 + c];
    st16<PST>(Pm + ((long long)(row >> 3) * 8 + ks) * 256 + (row & 7) * 32 + c4, v);
  }
}
// the owner puts the empty pattern back over its 8 x 256 partials: two 16-byte stores per thread
template <int PST> __device__ __forceinline__ void reset_partials4(float* Pown, int tid) {
  const f4v e = {__uint_as_float(EMPTY), __uint_as_float(EMPTY), __uint_as_float(EMPTY), __uint_as_float(EMPTY)};
  st16<PST>(Pown + 4 * tid, e);
  st16<PST>(Pown + 1024 + 4 * tid, e);
}

// P: [2 parity][3 matrix][NS][8 owner][8 producer][256]   rh: [2][64][1024]   hs: [T+1][64][1024]   pre: [T][3][64][1024]
template <int PST, int PLD, int LHS>
__global__ __launch_bounds__(256, 1) void gru_persist(const float* __restrict__ U, const float* __restrict__ pre, float* hs, float* rh,
                                                       float* P, int T, int* status, long long* prof) {
  __shared__ float s1[2][2][B][NW];  // phase 1: [matrix z/r][K-half][batch row][column]
  __shared__ float s2[2][B][NW];     // phase 2: [K-half][batch row][column]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int ns = xcd * 4 + (slot & 3), ks = slot >> 2;
  const int mh = wid & 1, kh = wid >> 1, j = lane & 31, half = lane >> 5;
  const int arow = 32 * mh + j;                         // the batch row this lane feeds to the MFMA
  const int ak0 = KW * ks + 64 * kh + 32 * half;        // ... and its 32 consecutive K indices
  // weight fragments, resident for the whole loop: B[k = ak0 + s][n = 32 ns + j]
  float bz[32], br[32], bc[32];
#pragma unroll
  for (int s = 0; s < 32; s++) {
    const long long o = (long long)(ak0 + s) * H + NW * ns + j;
    bz[s] = U[o];
    br[s] = U[(long long)H * H + o];
    bc[s] = U[2LL * H * H + o];
  }
  // owner role: batch row ob, column on
  const int ob = 8 * ks + (tid >> 5), on = NW * ns + (tid & 31);
  float hcur = hs[(long long)ob * H + on];
  long long pt[NPROF] = {0};
  long long tq = wall_clock64();
  const long long t_start = tq;
  bool alive = true;
  for (int t = 0; t < T && alive; t++) {
    const int par = t & 1;
    const long long t_end = (long long)wall_clock64() + POLL_TICKS;
    const float* pre_t = pre + (long long)t * 3 * B * H + (long long)ob * H + on;
    const float pz = pre_t[0], pr = pre_t[(long long)B * H], pc = pre_t[2LL * B * H];
    float* Pz = P + ((((long long)par * 3 + 0) * NS + ns) * 8) * 8 * 256;
    float* Pr = P + ((((long long)par * 3 + 1) * NS + ns) * 8) * 8 * 256;
    float* Pc = P + ((((long long)par * 3 + 2) * NS + ns) * 8) * 8 * 256;
    // ---------------- phase 1: gates ----------------
    float a[32];
    f16v az = {0}, ar = {0};
    if constexpr (LHS == 2) {  // products start on the first 16 bytes that are there
      const float* lp = hs + (long long)t * B * H + (long long)arow * H + ak0;
      f4v v[8];
      issue_lhs8(lp, v);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (!ensure_quad(lp + 4 * q, v[q], t_end, status)) { alive = false; break; }
#pragma unroll
        for (int c = 0; c < 4; c++) {
          az = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q][c], bz[4 * q + c], az, 0, 0, 0);
          ar = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q][c], br[4 * q + c], ar, 0, 0, 0);
        }
      }
      if (!alive) break;
    } else {
    if (!(LHS == 1 ? load_lhs4(hs + (long long)t * B * H, arow, ak0, a, t_end, status)
                 : load_lhs(hs + (long long)t * B * H, arow, ak0, a, t_end, status))) { alive = false; break; }
    if (tid == 0) { const long long c = wall_clock64(); pt[0] += c - tq; tq = c; }
#pragma unroll
    for (int s = 0; s < 32; s++) {
      az = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bz[s], az, 0, 0, 0);
      ar = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], br[s], ar, 0, 0, 0);
    }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = 32 * mh + 8 * (r >> 2) + 4 * half + (r & 3);
      s1[0][kh][row][j] = az[r];
      s1[1][kh][row][j] = ar[r];
    }
    __syncthreads();
    if constexpr (PST != 0) {
      if (t > 0) reset_partials4<PST>(P + ((((long long)(par ^ 1) * 3 + 2) * NS + ns) * 8) * 8 * 256 + (long long)ks * 8 * 256, tid);
    }
    if (tid == 0) { const long long c = wall_clock64(); pt[1] += c - tq; tq = c; }
    if constexpr (PST != 0) {
      store_partial4<PST>(Pz, ks, s1[0][0], s1[0][1], tid);
      store_partial4<PST>(Pr, ks, s1[1][0], s1[1][1], tid);
    } else {
#pragma unroll
      for (int o = 0; o < 8; o++) {
        const int row = 8 * o + (tid >> 5), col = tid & 31;
        st32(Pz + ((long long)o * 8 + ks) * 256 + tid, s1[0][0][row][col] + s1[0][1][row][col]);
        st32(Pr + ((long long)o * 8 + ks) * 256 + tid, s1[1][0][row][col] + s1[1][1][row][col]);
      }
    }
    if (tid == 0) { const long long c = wall_clock64(); pt[2] += c - tq; tq = c; }
    float sz, sr;
    if (!(PLD == 3 ? load_partials_sel(Pz + (long long)ks * 8 * 256 + tid, sz, t_end, status) : PLD != 0 ? load_partials_l2<PLD>(Pz + (long long)ks * 8 * 256 + tid, sz, t_end, status) : load_partials(Pz + (long long)ks * 8 * 256 + tid, sz, t_end, status)) ||
        !(PLD == 3 ? load_partials_sel(Pr + (long long)ks * 8 * 256 + tid, sr, t_end, status) : PLD != 0 ? load_partials_l2<PLD>(Pr + (long long)ks * 8 * 256 + tid, sr, t_end, status) : load_partials(Pr + (long long)ks * 8 * 256 + tid, sr, t_end, status))) { alive = false; break; }
    if (tid == 0) { const long long c = wall_clock64(); pt[3] += c - tq; tq = c; }
    const float z = sigmoidf_(sz + pz), rg = sigmoidf_(sr + pr);
    st32(rh + ((long long)par * B + ob) * H + on, rg * hcur);
    st32u(rh + ((long long)(par ^ 1) * B + ob) * H + on, EMPTY);  // (r*h of step t-1: every reader is past it)
    if constexpr (PST != 0) {
      // (16-byte resets cross the threads' own elements: they wait for the next barrier, below)
    } else {
#pragma unroll
      for (int kp = 0; kp < KS; kp++) {
        st32u(Pz + ((long long)ks * 8 + kp) * 256 + tid, EMPTY);
        st32u(Pr + ((long long)ks * 8 + kp) * 256 + tid, EMPTY);
      }
    }
    if (tid == 0) { const long long c = wall_clock64(); pt[4] += c - tq; tq = c; }
    // ---------------- phase 2: candidate and blend ----------------
    f16v ac = {0};
    if constexpr (LHS == 2) {
      const float* lp = rh + (long long)par * B * H + (long long)arow * H + ak0;
      f4v v[8];
      issue_lhs8(lp, v);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (!ensure_quad(lp + 4 * q, v[q], t_end, status)) { alive = false; break; }
#pragma unroll
        for (int c = 0; c < 4; c++) ac = __builtin_amdgcn_mfma_f32_32x32x2f32(v[q][c], bc[4 * q + c], ac, 0, 0, 0);
      }
      if (!alive) break;
    } else {
    if (!(LHS == 1 ? load_lhs4(rh + (long long)par * B * H, arow, ak0, a, t_end, status)
                 : load_lhs(rh + (long long)par * B * H, arow, ak0, a, t_end, status))) { alive = false; break; }
    if (tid == 0) { const long long c = wall_clock64(); pt[5] += c - tq; tq = c; }
#pragma unroll
    for (int s = 0; s < 32; s++) ac = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bc[s], ac, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) s2[kh][32 * mh + 8 * (r >> 2) + 4 * half + (r & 3)][j] = ac[r];
    __syncthreads();
    if constexpr (PST != 0) {  // every wave is past its gate epilogue: the z/r partials of this step can be emptied
      reset_partials4<PST>(Pz + (long long)ks * 8 * 256, tid);
      reset_partials4<PST>(Pr + (long long)ks * 8 * 256, tid);
    }
    if (tid == 0) { const long long c = wall_clock64(); pt[6] += c - tq; tq = c; }
    if constexpr (PST != 0) {
      store_partial4<PST>(Pc, ks, s2[0], s2[1], tid);
    } else {
#pragma unroll
      for (int o = 0; o < 8; o++) {
        const int row = 8 * o + (tid >> 5), col = tid & 31;
        st32(Pc + ((long long)o * 8 + ks) * 256 + tid, s2[0][row][col] + s2[1][row][col]);
      }
    }
    if (tid == 0) { const long long c = wall_clock64(); pt[7] += c - tq; tq = c; }
    float sc;
    if (!(PLD == 3 ? load_partials_sel(Pc + (long long)ks * 8 * 256 + tid, sc, t_end, status) : PLD != 0 ? load_partials_l2<PLD>(Pc + (long long)ks * 8 * 256 + tid, sc, t_end, status) : load_partials(Pc + (long long)ks * 8 * 256 + tid, sc, t_end, status))) { alive = false; break; }
    if (tid == 0) { const long long c = wall_clock64(); pt[8] += c - tq; tq = c; }
    const float cand = tanhf(sc + pc);
    hcur = (1.f - z) * hcur + z * cand;
    st32(hs + ((long long)(t + 1) * B + ob) * H + on, hcur);
    if constexpr (PST != 0) {
      // (emptied after the next step's first barrier)
    } else {
#pragma unroll
      for (int kp = 0; kp < KS; kp++) st32u(Pc + ((long long)ks * 8 + kp) * 256 + tid, EMPTY);
    }
    if (tid == 0) { const long long c = wall_clock64(); pt[9] += c - tq; tq = c; }
  }
  if (tid == 0 && prof != nullptr) {
    pt[10] = wall_clock64() - t_start;
    for (int i = 0; i < NPROF; i++) prof[(long long)blockIdx.x * NPROF + i] = pt[i];
  }
}

// ---- the plain form, one thread per output element: the reference the persistent kernel is compared with ----
__global__ void ref_gates(const float* __restrict__ U, const float* __restrict__ pre_t, const float* __restrict__ h, float* z, float* rhv) {
  const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  float sz = 0.f, sr = 0.f;
  for (int k = 0; k < H; k++) {
    const float hv = h[b * H + k];
    sz += hv * U[(long long)k * H + n];
    sr += hv * U[(long long)H * H + (long long)k * H + n];
  }
  z[b * H + n] = sigmoidf_(sz + pre_t[b * H + n]);
  rhv[b * H + n] = sigmoidf_(sr + pre_t[B * H + b * H + n]) * h[b * H + n];
}
__global__ void ref_blend(const float* __restrict__ U, const float* __restrict__ pre_t, const float* __restrict__ h, const float* __restrict__ z,
                          const float* __restrict__ rhv, float* hn) {
  const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  float sc = 0.f;
  for (int k = 0; k < H; k++) sc += rhv[b * H + k] * U[2LL * H * H + (long long)k * H + n];
  const float zz = z[b * H + n];
  hn[b * H + n] = (1.f - zz) * h[b * H + n] + zz * tanhf(sc + pre_t[2 * B * H + b * H + n]);
}
__global__ void fill_hash(float* p, long long n, unsigned seed, float scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed ^ (unsigned)(i >> 32) * 40503u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((x >> 8) * (1.f / 8388608.f) - 1.f) * scale;
  }
}

static void cpu_steps(const std::vector<float>& U, const std::vector<float>& pre, std::vector<double>& h, int T) {
  std::vector<double> sz(B * H), sr(B * H), sc(B * H), rhv(B * H), z(B * H);
  for (int t = 0; t < T; t++) {
    std::fill(sz.begin(), sz.end(), 0.0); std::fill(sr.begin(), sr.end(), 0.0); std::fill(sc.begin(), sc.end(), 0.0);
    for (int b = 0; b < B; b++)
      for (int k = 0; k < H; k++) {
        const double hv = h[b * H + k];
        const float* uz = &U[(size_t)k * H]; const float* ur = &U[(size_t)H * H + (size_t)k * H];
        for (int n = 0; n < H; n++) { sz[b * H + n] += hv * uz[n]; sr[b * H + n] += hv * ur[n]; }
      }
    const float* p = &pre[(size_t)t * 3 * B * H];
    for (int i = 0; i < B * H; i++) {
      z[i] = 1.0 / (1.0 + std::exp(-(sz[i] + p[i])));
      rhv[i] = h[i] / (1.0 + std::exp(-(sr[i] + p[B * H + i])));
    }
    for (int b = 0; b < B; b++)
      for (int k = 0; k < H; k++) {
        const double v = rhv[b * H + k];
        const float* uc = &U[2 * (size_t)H * H + (size_t)k * H];
        for (int n = 0; n < H; n++) sc[b * H + n] += v * uc[n];
      }
    for (int i = 0; i < B * H; i++) h[i] = (1.0 - z[i]) * h[i] + z[i] * std::tanh(sc[i] + p[2 * B * H + i]);
  }
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 1000;
  const int TC = 6;  // steps checked against the fp64 host recurrence
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  float *U, *pre, *hs, *rh, *P, *z, *rhv, *href;
  int* status; long long* prof;
  const size_t nP = 2ull * 3 * NS * 8 * 8 * 256;
  CK(hipMalloc(&U, 3ull * H * H * 4));
  CK(hipMalloc(&pre, (size_t)T * 3 * B * H * 4));
  CK(hipMalloc(&hs, (size_t)(T + 1) * B * H * 4));
  CK(hipMalloc(&rh, 2ull * B * H * 4));
  CK(hipMalloc(&P, nP * 4));
  CK(hipMalloc(&z, B * H * 4)); CK(hipMalloc(&rhv, B * H * 4)); CK(hipMalloc(&href, 2ull * B * H * 4));
  CK(hipMalloc(&status, 4)); CK(hipMalloc(&prof, 256 * NPROF * 8));
  fill_hash<<<1024, 256>>>(U, 3ll * H * H, 11u, 0.05f);
  fill_hash<<<4096, 256>>>(pre, (long long)T * 3 * B * H, 23u, 1.0f);
  std::vector<float> h0(B * H);
  { unsigned x = 12345u; for (auto& v : h0) { x = x * 1664525u + 1013904223u; v = ((x >> 8) * (1.f / 8388608.f) - 1.f); } }
  CK(hipDeviceSynchronize());

  int variant = 1;
  std::vector<float> hlast1;
  auto run = [&](int steps, float* ms) -> int {
    CK(hipMemset(hs, 0xFF, (size_t)(T + 1) * B * H * 4));
    CK(hipMemcpy(hs, h0.data(), B * H * 4, hipMemcpyHostToDevice));
    CK(hipMemset(rh, 0xFF, 2ull * B * H * 4));
    CK(hipMemset(P, 0xFF, nP * 4));
    CK(hipMemset(status, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    if (variant == 2) gru_persist<2, 0, 1><<<256, 256>>>(U, pre, hs, rh, P, steps, status, prof);
    else if (variant == 3) gru_persist<1, 0, 1><<<256, 256>>>(U, pre, hs, rh, P, steps, status, prof);
    else if (variant == 4) gru_persist<1, 2, 1><<<256, 256>>>(U, pre, hs, rh, P, steps, status, prof);
    else if (variant == 5) gru_persist<1, 1, 1><<<256, 256>>>(U, pre, hs, rh, P, steps, status, prof);
    else if (variant == 6) gru_persist<1, 3, 2><<<256, 256>>>(U, pre, hs, rh, P, steps, status, prof);
    else if (variant == 7) gru_persist<2, 3, 2><<<256, 256>>>(U, pre, hs, rh, P, steps, status, prof);
    else gru_persist<0, 0, 0><<<256, 256>>>(U, pre, hs, rh, P, steps, status, prof);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(ms, e0, e1));
    int st; CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    return st;
  };

  float ms;
  std::vector<float> hU(3ull * H * H), hpre((size_t)TC * 3 * B * H);
  CK(hipMemcpy(hU.data(), U, hU.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hpre.data(), pre, hpre.size() * 4, hipMemcpyDeviceToHost));
  std::vector<double> hwant(h0.begin(), h0.end());
  cpu_steps(hU, hpre, hwant, TC);
  for (variant = 1; variant <= 7; variant++) {
  printf("==== variant %d: %s\n", variant, variant == 1 ? "every hand-over through agent-scope loads/stores, 4-byte stores"
                                     : variant == 2 ? "agent scope, 16-byte stores and left-operand loads"
                                     : variant == 3 ? "partials: PLAIN 16-byte stores, agent-scope loads"
                                     : variant == 6 ? "variant 3 + only the empty words are asked for again, products start on the first 16 bytes (wait rows include the products)"
                                     : variant == 7 ? "variant 6 with agent-scope partial stores"
                                     : variant == 4 ? "partials through the XCD's L2: plain stores, sc0 loads"
                                                    : "partials through the XCD's L2: plain stores, buffer_inv sc0 + plain loads");
  // ---- correctness: TC steps against the host recurrence in fp64 ----
  int st = run(TC, &ms);
  printf("persistent kernel, %d steps: status %d, %.3f ms\n", TC, st, ms);
  if (st != 0) { printf("hand-over timed out (status bits: 1 = left operand, 2 = partials)\n"); continue; }
  {
    std::vector<float> got(B * H);
    CK(hipMemcpy(got.data(), hs + (size_t)TC * B * H, B * H * 4, hipMemcpyDeviceToHost));
    const std::vector<double>& h = hwant;
    double worst = 0;
    for (int i = 0; i < B * H; i++) worst = std::fmax(worst, std::fabs(got[i] - h[i]) / (1e-5 + 1e-5 * std::fabs(h[i])));
    printf("after %d steps vs fp64 host recurrence: worst error %.3f of (1e-5 abs + 1e-5 rel)  -> %s\n", TC, worst, worst <= 1.0 ? "ok" : "MISMATCH");
    if (!(worst <= 1.0)) continue;
  }
  // ---- timing: T steps, three runs ----
  for (int rep = 0; rep < 3; rep++) {
    st = run(T, &ms);
    printf("persistent kernel, %d steps: status %d, %.3f ms = %.2f us/step\n", T, st, ms, ms * 1e3 / T);
    if (st != 0) break;
  }
  if (st != 0) continue;
  std::vector<long long> hp(256 * NPROF);
  CK(hipMemcpy(hp.data(), prof, hp.size() * 8, hipMemcpyDeviceToHost));
  const char* names[] = {"wait h slice", "mfma gates + LDS fold", "store partials z,r", "wait partials z,r", "gate epilogue + publish r*h + resets",
                         "wait r*h slice", "mfma candidate + LDS fold", "store partial c", "wait partial c", "blend + publish h' + resets", "whole loop"};
  for (int wg : {0}) {
    printf("workgroup %d (per step, wall clock of thread 0):\n", wg);
    for (int i = 0; i < 11; i++) printf("   %-40s %7.3f us\n", names[i], hp[wg * NPROF + i] * 0.01 / T);
  }
  double mean[11] = {0};
  for (int wg = 0; wg < 256; wg++) for (int i = 0; i < 11; i++) mean[i] += hp[wg * NPROF + i] * 0.01 / T / 256;
  printf("mean over the 256 workgroups:\n");
  for (int i = 0; i < 11; i++) printf("   %-40s %7.3f us\n", names[i], mean[i]);
  {
    std::vector<float> b(B * H);
    CK(hipMemcpy(b.data(), hs + (size_t)T * B * H, B * H * 4, hipMemcpyDeviceToHost));
    if (variant == 1) hlast1 = b;
    else if (!hlast1.empty()) printf("final state after %d steps: %s variant 1\n", T, memcmp(b.data(), hlast1.data(), b.size() * 4) == 0 ? "bit-identical to" : "DIFFERS from");
  }
  }  // variants
  // ---- the same T steps in the plain form (two launches per step), for the final-state comparison ----
  CK(hipMemcpy(href, h0.data(), B * H * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int t = 0; t < T; t++) {
    float* hc = href + (size_t)(t & 1) * B * H; float* hn = href + (size_t)((t + 1) & 1) * B * H;
    ref_gates<<<dim3(H / 256, B), 256>>>(U, pre + (size_t)t * 3 * B * H, hc, z, rhv);
    ref_blend<<<dim3(H / 256, B), 256>>>(U, pre + (size_t)t * 3 * B * H, hc, z, rhv, hn);
  }
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> a(B * H), b(B * H);
  CK(hipMemcpy(a.data(), href + (size_t)(T & 1) * B * H, B * H * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), hs + (size_t)T * B * H, B * H * 4, hipMemcpyDeviceToHost));
  double worst = 0, rms = 0;
  for (int i = 0; i < B * H; i++) { const double d = std::fabs((double)a[i] - b[i]); worst = std::fmax(worst, d); rms += d * d; }
  printf("plain form (one thread per element, two launches per step): %.2f us/step; final state after %d steps differs by max %.3g, rms %.3g\n",
         ms * 1e3 / T, T, worst, std::sqrt(rms / (B * H)));
  return 0;
}
