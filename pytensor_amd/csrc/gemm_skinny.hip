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
__global__ __launch_bounds__(BLOCK) void skinny_fwd_kernel(T* __restrict__ out, const T* __restrict__ A, long long lda, const T* __restrict__ B,
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
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long m0 = tile * TR;
    T acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = T(0);
    for (int k0 = 0; k0 < K; k0 += KC) {
      const int kc = K - k0 < KC ? K - k0 : KC;
      __syncthreads();  // the previous chunk / tile has been consumed
      if (!one_chunk) {
        for (int idx = t; idx < KC * NW; idx += BLOCK) {
          const int k = idx / NW, n = idx - k * NW;
          Bs[idx] = (k < kc && n < N) ? B[(long long)(k0 + k) * sB0 + (long long)n * sB1] : T(0);
        }
      }
      // wave w stages rows w, w + 4, ...: one row chunk (kc elements, contiguous) per load instruction
#pragma unroll 4
      for (int r = wv; r < TR; r += 4) {
        const long long m = m0 + r;
        for (int kk = lane * V; kk < KC; kk += 64 * V) {
          P v;
          if (m < M && kk + V <= kc) {
            v = *reinterpret_cast<const P*>(A + m * lda + k0 + kk);
          } else {
#pragma unroll
            for (int e = 0; e < V; e++) v.v[e] = (m < M && kk + e < kc) ? A[m * lda + k0 + kk + e] : T(0);
          }
#pragma unroll
          for (int e = 0; e < V; e++) As[r * PITCH + kk + e] = v.v[e];
        }
      }
      __syncthreads();
      const T* ar = As + lane * PITCH;
      const T* br = Bs + wv * NC;
#pragma unroll 8
      for (int k = 0; k < kc; k++) {
        const T a = ar[k];
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = __builtin_fma(a, br[k * NW + c], acc[c]);
      }
    }
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
}

// ---- backward: slab[b][m][n] = sum over the workgroup's rows i of X[i, m] * W[i, n];  X rows have unit column stride ----
// thread (ml = t % 128, nh = t / 128): result rows m = ml + 128*j (j < MJ), columns [nh*NC, nh*NC + NC) of the <= 2*NC wide W
template <class T, int V, int MJ, int NC>
__global__ __launch_bounds__(BLOCK) void skinny_atb_kernel(T* __restrict__ slab, const T* __restrict__ X, long long ldx, const T* __restrict__ W,
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
  for (long long r0 = r_begin; r0 < r_end; r0 += tr) {
    const int nr = r_end - r0 < tr ? (int)(r_end - r0) : tr;
    __syncthreads();
    for (int idx = t; idx < nr * packs_per_row; idx += BLOCK) {
      const int i = idx / packs_per_row, q = (idx - i * packs_per_row) * V;
      P v;
      if (q + V <= M) {
        v = *reinterpret_cast<const P*>(X + (r0 + i) * ldx + q);
      } else {
#pragma unroll
        for (int e = 0; e < V; e++) v.v[e] = q + e < M ? X[(r0 + i) * ldx + q + e] : T(0);
      }
      *reinterpret_cast<P*>(Xs + (long long)i * MP + q) = v;
    }
    for (int idx = t; idx < nr * NW; idx += BLOCK) {
      const int i = idx / NW, n = idx - i * NW;
      Ws[idx] = n < N ? W[(r0 + i) * ldw + n] : T(0);
    }
    __syncthreads();
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

// out[m, n] = beta*C[m, n] + alpha * (slab[0] + slab[1] + ...)[m, n], slabs in workgroup order
template <class T>
__global__ __launch_bounds__(BLOCK) void skinny_finish_kernel(T* __restrict__ out, const T* __restrict__ slab, int nslab, int MN, int N,
                                                             const T* __restrict__ C, long long sC0, long long sC1, T alpha, T beta) {
  const int e = blockIdx.x * BLOCK + threadIdx.x;
  if (e >= MN) return;
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int s = 0;
  for (; s + 3 < nslab; s += 4) {
    a0 += slab[(long long)s * MN + e];
    a1 += slab[(long long)(s + 1) * MN + e];
    a2 += slab[(long long)(s + 2) * MN + e];
    a3 += slab[(long long)(s + 3) * MN + e];
  }
  for (; s < nslab; s++) a0 += slab[(long long)s * MN + e];
  T r = alpha * ((a0 + a1) + (a2 + a3));
  if (beta != T(0)) {
    const int m = e / N, n = e - m * N;
    r += beta * C[(long long)m * sC0 + (long long)n * sC1];
  }
  out[e] = r;
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
    int tr = (int)((56 * 1024) / ((size_t)((Mi + VW - 1) / VW * VW) * sizeof(T)));
    if (tr > 64) tr = 64;
    if (tr < 4) tr = 4;
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
      PTHIP_KLAUNCH((skinny_finish_kernel<T>), dim3((unsigned)((MN + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, pthip::ctx().stream, out, (const T*)slab, (int)nblocks,
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
