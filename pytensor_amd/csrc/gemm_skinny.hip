// gemm_skinny.hip — tall-and-skinny products: one operand has ~1e6 rows, the other side of the product is <= 16 wide.
//
// Reference ops: Gemm / Dot22 (pytensor/tensor/blas/gemm.py:76, dot22.py) as they appear in a multi-response regression
// logp + gradient graph (oracle/ref_graphs.build_wide200_gemm):
//     forward   Z = beta*C + alpha * X @ B        X (N_obs, K) row-major,  B (K, R),   R <= 16
//     backward  G = beta*C + alpha * X.T @ W      X.T a transposed VIEW of X,  W (N_obs, R) row-major
// Both stream X once and do 2*R flops per element read: ~2 flop/byte at R = 8 — HBM-bound by a factor of five on this
// chip, so the MFMA tiles of gemm.hip (64 or 128 columns of C per workgroup, split-K slabs for the backward product) buy
// nothing and waste 8x the matrix-pipe work on padding: round 6 measured 2.2 ms per product there (profiles/r7_*).
// Here: the long operand goes through LDS in 64-row tiles with coalesced 16-byte loads, the short operand sits in LDS,
// plain v_fma_f64 / v_fma_f32 do the arithmetic.  The backward product leaves one (K, R) slab per workgroup and a
// second launch adds the slabs in workgroup order: deterministic, no atomics.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int TR = 64;   // rows of the long operand per tile (forward)
constexpr int KC = 128;  // its columns per LDS chunk (forward)

template <class T, int V> struct __attribute__((aligned(sizeof(T) * V))) sk_pack { T v[V]; };

// ---- forward: out[m, n] = beta*C[m, n] + alpha * sum_k A[m, k] B[k, n];  A rows have unit column stride ------------------
// thread (row = t % 64, ng = t / 64): one row of the tile, columns [ng*NC, ng*NC + NC) of the <= 4*NC wide result
template <class T, int V, int NC>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(1, 2))) void skinny_fwd_kernel(T* __restrict__ out, const T* __restrict__ A, long long lda, const T* __restrict__ B,
                                                          long long sB0, long long sB1, const T* __restrict__ C, long long sC0, long long sC1,
                                                          long long M, int N, int K, T alpha, T beta, long long tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_lds_[];
  constexpr int PITCH = KC + 1;  // odd pitch: the 64 rows a wave reads at one k fall into different banks
  constexpr int NW = 4 * NC;
  T* As = reinterpret_cast<T*>(sk_lds_);  // [TR][PITCH]
  T* Bs = As + TR * PITCH;                // [KC][NW], zero beyond N
  typedef sk_pack<T, V> P;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const bool one_chunk = K <= KC;
  if (one_chunk) {
    for (int idx = t; idx < KC * NW; idx += BLOCK) {
      const int k = idx / NW, n = idx - k * NW;
      Bs[idx] = (k < K && n < N) ? B[(long long)k * sB0 + (long long)n * sB1] : T(0);
    }
  }
  // Steps = (tile, K chunk) pairs in order.  The operand of step s + 1 is requested into registers BEFORE step s is
  // computed out of LDS: every workgroup does the same work per step, so without it all of them loaded, then all of them
  // computed, and HBM idled through every compute phase (8 us per tile instead of ~3: 520 -> 245 us per GB, r7 profiles).
  // Loads are unconditional at clamped addresses (a pack that starts below kc may run up to V - 1 elements past it: still
  // inside the row pitch, a multiple of the pack whenever V > 1 — launch_fwd's alignment test); only elements below kc
  // are stored.
  constexpr int CPR = KC / (64 * V) > 0 ? KC / (64 * V) : 1;  // packs of one row chunk per lane
  constexpr int NP = (TR / 4) * CPR;
  const int nchunk = (K + KC - 1) / KC;
  const long long last_row = M - 1;
  auto request = [&](long long tile, int k0, P (&buf)[NP]) {
    const long long m0 = tile * TR;
    const int kc = K - k0 < KC ? K - k0 : KC;
#pragma unroll
    for (int q = 0; q < NP; q++) {
      const int r = wv + 4 * (q / CPR), kk = (lane + 64 * (q % CPR)) * V;
      const long long m = m0 + r < M ? m0 + r : last_row;
      buf[q] = *reinterpret_cast<const P*>(A + m * lda + k0 + (kk < kc ? kk : 0));
    }
  };
  P buf[NP];
  long long tile = blockIdx.x;
  int chunk = 0;
  if (tile < tiles) request(tile, 0, buf);
  T acc[NC];
  while (tile < tiles) {
    const long long m0 = tile * TR;
    const int k0 = chunk * KC;
    const int kc = K - k0 < KC ? K - k0 : KC;
    if (chunk == 0) {
#pragma unroll
      for (int c = 0; c < NC; c++) acc[c] = T(0);
    }
    __syncthreads();  // the previous step has been consumed
    if (!one_chunk) {
      for (int idx = t; idx < KC * NW; idx += BLOCK) {
        const int k = idx / NW, n = idx - k * NW;
        Bs[idx] = (k < kc && n < N) ? B[(long long)(k0 + k) * sB0 + (long long)n * sB1] : T(0);
      }
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {
      const int r = wv + 4 * (q / CPR), kk = (lane + 64 * (q % CPR)) * V;
#pragma unroll
      for (int e = 0; e < V; e++)
        if (kk + e < kc) As[r * PITCH + kk + e] = buf[q].v[e];
    }
    __syncthreads();
    // the next step's operand: in flight during this step's arithmetic
    long long ntile = tile;
    int nchk = chunk + 1;
    if (nchk == nchunk) { nchk = 0; ntile = tile + gridDim.x; }
    if (ntile < tiles) request(ntile, nchk * KC, buf);
    const T* ar = As + lane * PITCH;
    const T* br = Bs + wv * NC;
#pragma unroll 8
    for (int k = 0; k < kc; k++) {
      const T a = ar[k];
#pragma unroll
      for (int c = 0; c < NC; c++) acc[c] = __builtin_fma(a, br[k * NW + c], acc[c]);
    }
    if (nchk == 0) {
      const long long m = m0 + lane;
      if (m < M) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
          const int n = wv * NC + c;
          if (n < N) {
            T r = alpha * acc[c];
            if (beta != T(0)) r += beta * C[m * sC0 + (long long)n * sC1];
            out[m * N + n] = r;
          }
        }
      }
    }
    tile = ntile;
    chunk = nchk;
  }
}

// ---- backward: slab[b][m][n] = sum over the workgroup's rows i of X[i, m] * W[i, n];  X rows have unit column stride ----
// thread (ml = t % 128, nh = t / 128): result rows m = ml + 128*j (j < MJ), columns [nh*NC, nh*NC + NC) of the <= 2*NC wide W
template <class T, int V, int MJ, int NC>
// (two workgroups per CU at most — the LDS tile decides that — so a wave may use up to 256 VGPRs: without the hint the
//  register allocator kept the 16 prefetched packs in scratch memory)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(1, 2))) void skinny_atb_kernel(T* __restrict__ slab, const T* __restrict__ X, long long ldx, const T* __restrict__ W,
                                                          long long ldw, long long rows, int M, int N, int tr, long long rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_lds_[];
  constexpr int NW = 2 * NC;
  const int MP = (M + V - 1) / V * V;        // row pitch of the staged X tile
  T* Xs = reinterpret_cast<T*>(sk_lds_);     // [tr][MP]
  T* Ws = Xs + (long long)tr * MP;           // [tr][NW], zero beyond N
  typedef sk_pack<T, V> P;
  const int t = threadIdx.x, ml = t & 127, nh = t >> 7;
  T acc[MJ][NC];
#pragma unroll
  for (int j = 0; j < MJ; j++)
#pragma unroll
    for (int c = 0; c < NC; c++) acc[j][c] = T(0);
  const long long r_begin = (long long)blockIdx.x * rows_per_block;
  long long r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  const int packs_per_row = MP / V;
  // The next tile (up to SG packs of X and two values of W per thread) is requested into registers before the current one
  // is consumed out of LDS — see skinny_fwd_kernel.  Unconditional loads, clamped to the workgroup's last row; a pack may
  // run past M inside the row pitch.  (tr * packs_per_row <= SG * BLOCK and tr * NW <= 2 * BLOCK by the host's choice of tr.)
  constexpr int SG = 16;
  P xb[SG];
  T wb[2];
  const long long r_last = r_end - 1;
  // (the buffers are PARAMETERS of the lambda: captured by reference they stayed in scratch memory — 272 bytes of
  //  private segment, every access a scratch load behind vmcnt(0), 409 -> 803 us)
  auto request = [&](long long r0, P (&xb)[SG], T (&wb)[2]) {
#pragma unroll
    for (int u = 0; u < SG; u++) {
      const int idx = u * BLOCK + t;
      const int i = idx / packs_per_row, q = (idx - i * packs_per_row) * V;
      const long long rr = r0 + i < r_end ? r0 + i : r_last;
      xb[u] = *reinterpret_cast<const P*>(X + rr * ldx + q);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int idx = u * BLOCK + t;
      const int i = idx / NW, n = idx - i * NW;
      const long long rr = r0 + i < r_end ? r0 + i : r_last;
      wb[u] = W[rr * ldw + (n < N ? n : 0)];
    }
  };
  if (r_begin < r_end) request(r_begin, xb, wb);
  for (long long r0 = r_begin; r0 < r_end; r0 += tr) {
    const int nr = r_end - r0 < tr ? (int)(r_end - r0) : tr;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SG; u++) {
      const int idx = u * BLOCK + t;
      if (idx < nr * packs_per_row) {
        const int i = idx / packs_per_row, q = (idx - i * packs_per_row) * V;
        T* d = Xs + (long long)i * MP + q;  // (element by element: an aggregate copy out of the register array kept it in scratch)
#pragma unroll
        for (int e = 0; e < V; e++) d[e] = xb[u].v[e];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int idx = u * BLOCK + t;
      if (idx < nr * NW) Ws[idx] = (idx % NW) < N ? wb[u] : T(0);
    }
    __syncthreads();
    if (r0 + tr < r_end) request(r0 + tr, xb, wb);
#pragma unroll 4
    for (int i = 0; i < nr; i++) {
      T w[NC];
#pragma unroll
      for (int c = 0; c < NC; c++) w[c] = Ws[i * NW + nh * NC + c];  // (one address per half-block: a broadcast)
#pragma unroll
      for (int j = 0; j < MJ; j++) {
        const int m = ml + 128 * j;
        const T x = m < M ? Xs[(long long)i * MP + m] : T(0);
#pragma unroll
        for (int c = 0; c < NC; c++) acc[j][c] = __builtin_fma(x, w[c], acc[j][c]);
      }
    }
  }
  T* dst = slab + (long long)blockIdx.x * M * N;
#pragma unroll
  for (int j = 0; j < MJ; j++) {
    const int m = ml + 128 * j;
    if (m < M) {
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const int n = nh * NC + c;
        if (n < N) dst[(long long)m * N + n] = acc[j][c];
      }
    }
  }
}

// out[m, n] = beta*C[m, n] + alpha * sum of the slabs at [m, n]: 16 entries per workgroup, 16 slab lanes each adding every
// 16th slab (four loads in flight), the lanes folded in lane order through LDS — a fixed order, whatever the schedule
template <class T>
__global__ __launch_bounds__(BLOCK) void skinny_finish_kernel(T* __restrict__ out, const T* __restrict__ slab, int nslab, int MN, int N,
                                                             const T* __restrict__ C, long long sC0, long long sC1, T alpha, T beta) {
  __shared__ T red[16][17];
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  const int ec = e < MN ? e : MN - 1;
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int s = sl;
  for (; s + 48 < nslab; s += 64) {
    const T v0 = slab[(long long)s * MN + ec], v1 = slab[(long long)(s + 16) * MN + ec];
    const T v2 = slab[(long long)(s + 32) * MN + ec], v3 = slab[(long long)(s + 48) * MN + ec];
    a0 += v0;
    a1 += v1;
    a2 += v2;
    a3 += v3;
  }
  for (; s < nslab; s += 16) a0 += slab[(long long)s * MN + ec];
  red[sl][el] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && e < MN) {
    T acc = red[0][el];
#pragma unroll
    for (int j = 1; j < 16; j++) acc += red[j][el];
    T r = alpha * acc;
    if (beta != T(0)) {
      const int m = e / N, n = e - m * N;
      r += beta * C[(long long)m * sC0 + (long long)n * sC1];
    }
    out[e] = r;
  }
}

template <class K> int raise_lds(K kernel, size_t bytes, size_t* have) {
  if (bytes > *have && bytes > 48 * 1024) {
    PTHIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    *have = bytes;
  }
  return 0;
}

template <class T, int V, int NC>
int launch_fwd(long long M, int N, int K, double alpha, const T* A, long long lda, const T* B, long long sB0, long long sB1, double beta,
               const T* C, long long sC0, long long sC1, T* out) {
  static size_t have = 0;
  const size_t lds = ((size_t)TR * (KC + 1) + (size_t)KC * 4 * NC) * sizeof(T);
  int r = raise_lds(skinny_fwd_kernel<T, V, NC>, lds, &have);
  if (r) return r;
  const long long tiles = (M + TR - 1) / TR;
  long long grid = tiles;
  const long long cap = (long long)pthip::kNumCU * 8;
  if (grid > cap) grid = cap;
  PTHIP_KLAUNCH((skinny_fwd_kernel<T, V, NC>), dim3((unsigned)grid), dim3(BLOCK), lds, pthip::ctx().stream, out, A, lda, B, sB0, sB1, C, sC0, sC1, M, N, K,
                (T)alpha, (T)(C ? beta : 0.0), tiles);
  return pthip::post_launch("gemm(skinny forward)");
}

template <class T, int V, int MJ, int NC>
int launch_atb(long long rows, int M, int N, const T* X, long long ldx, const T* W, long long ldw, T* slab, int nblocks, long long rows_per_block, int tr) {
  static size_t have = 0;
  const int MP = (M + V - 1) / V * V;
  const size_t lds = ((size_t)tr * MP + (size_t)tr * 2 * NC) * sizeof(T);
  int r = raise_lds(skinny_atb_kernel<T, V, MJ, NC>, lds, &have);
  if (r) return r;
  PTHIP_KLAUNCH((skinny_atb_kernel<T, V, MJ, NC>), dim3((unsigned)nblocks), dim3(BLOCK), lds, pthip::ctx().stream, slab, X, ldx, W, ldw, rows, M, N, tr,
                rows_per_block);
  return pthip::post_launch("gemm(skinny transposed, partial slabs)");
}

template <class T>
int skinny_typed(long long M, long long N, long long K, double alpha, const T* A, long long sA0, long long sA1, const T* B, long long sB0, long long sB1,
                 double beta, const T* C, long long sC0, long long sC1, T* out, bool* handled) {
  constexpr int VW = 16 / (int)sizeof(T);
  *handled = false;
  if (N < 1 || N > 16) return 0;
  // ---- forward: M long, A rows contiguous --------------------------------------------------------------------------------
  if (M >= 8192 && K >= 1 && K <= 4096 && (sA1 == 1 || K == 1) && sA0 >= K) {
    *handled = true;
    const bool al = ((uintptr_t)A % 16) == 0 && sA0 % VW == 0;
    const int NCn = N <= 4 ? 1 : (N <= 8 ? 2 : 4);
#define FWD(V, NC) return launch_fwd<T, V, NC>(M, (int)N, (int)K, alpha, A, sA0, B, sB0, sB1, beta, C, sC0, sC1, out)
    if (al) {
      if (NCn == 1) FWD(VW, 1);
      if (NCn == 2) FWD(VW, 2);
      FWD(VW, 4);
    }
    if (NCn == 1) FWD(1, 1);
    if (NCn == 2) FWD(1, 2);
    FWD(1, 4);
#undef FWD
  }
  // ---- backward: the reduced dimension K long; A is the transposed view of a row-major (K, M) array; B rows contiguous -----
  if (K >= 8192 && M >= 1 && M <= 512 && sA0 == 1 && sA1 >= M && (sB1 == 1 || N == 1) && sB0 >= N) {
    *handled = true;
    const long long rows = K;
    const int Mi = (int)M, Ni = (int)N;
    const bool al = ((uintptr_t)A % 16) == 0 && sA1 % VW == 0;
    const int MJ = Mi <= 128 ? 1 : (Mi <= 256 ? 2 : 4);
    // rows per tile: the X tile within 56 KB of LDS, and within what a thread prefetches into registers
    // (16 packs of X, 2 values of W per thread: skinny_atb_kernel)
    const int Vsel = al ? VW : 1;
    const int MPh = (Mi + Vsel - 1) / Vsel * Vsel;
    const int NWh = 2 * (Ni <= 8 ? 4 : 8);
    int tr = (int)((56 * 1024) / ((size_t)MPh * sizeof(T)));
    if (tr > 64) tr = 64;
    if (tr > 16 * BLOCK / (MPh / Vsel)) tr = 16 * BLOCK / (MPh / Vsel);
    if (tr > 2 * BLOCK / NWh) tr = 2 * BLOCK / NWh;
    if (tr < 1) tr = 1;
    // ~2 workgroups per CU, each a whole number of tiles
    long long nblocks = (long long)pthip::kNumCU * 2;
    long long rpb = (rows + nblocks - 1) / nblocks;
    rpb = (rpb + tr - 1) / tr * tr;
    nblocks = (rows + rpb - 1) / rpb;
    void* scratch = nullptr;
    int r = pthip_alloc((size_t)nblocks * Mi * Ni * sizeof(T), &scratch);
    if (r) return r;
    T* slab = (T*)scratch;
#define ATB(V, MJ_, NC) r = launch_atb<T, V, MJ_, NC>(rows, Mi, Ni, A, sA1, B, sB0, slab, (int)nblocks, rpb, tr)
#define ATB_MJ(V, NC)                  \
  do {                                 \
    if (MJ == 1) ATB(V, 1, NC);        \
    else if (MJ == 2) ATB(V, 2, NC);   \
    else ATB(V, 4, NC);                \
  } while (0)
    if (al) {
      if (Ni <= 8) ATB_MJ(VW, 4); else ATB_MJ(VW, 8);
    } else {
      if (Ni <= 8) ATB_MJ(1, 4); else ATB_MJ(1, 8);
    }
#undef ATB_MJ
#undef ATB
    if (!r) {
      const int MN = Mi * Ni;
      PTHIP_KLAUNCH((skinny_finish_kernel<T>), dim3((unsigned)((MN + 15) / 16)), dim3(BLOCK), 0, pthip::ctx().stream, out, (const T*)slab, (int)nblocks,
                    MN, Ni, C, sC0, sC1, (T)alpha, (T)(C ? beta : 0.0));
      r = pthip::post_launch("gemm(skinny transposed, finish)");
    }
    const int rf = pthip_free(scratch);  // stream-ordered reuse keeps this safe
    return r ? r : rf;
  }
  return 0;
}

}  // namespace

namespace pthip {

// called first by pthip_gemm for unbatched products (gemm.hip): *handled says whether this file took the product
int gemm_skinny(int dtype, long long M, long long N, long long K, double alpha, const void* A, long long sA0, long long sA1, const void* B,
                long long sB0, long long sB1, double beta, const void* C, long long sC0, long long sC1, void* out, bool* handled) {
  *handled = false;
  static const bool off = [] { const char* e = getenv("PTHIP_GEMM_SKINNY"); return e && e[0] == '0'; }();
  if (off || out == C) return 0;
  if (dtype == PTHIP_F64)
    return skinny_typed<double>(M, N, K, alpha, (const double*)A, sA0, sA1, (const double*)B, sB0, sB1, beta, (const double*)C, sC0, sC1, (double*)out, handled);
  if (dtype == PTHIP_F32)
    return skinny_typed<float>(M, N, K, alpha, (const float*)A, sA0, sA1, (const float*)B, sB0, sB1, beta, (const float*)C, sC0, sC1, (float*)out, handled);
  return 0;
}

}  // namespace pthip
