// linalg.hip — Cholesky (potrf) and triangular solves (trtrs/potrs building block).
//
// Reference: Cholesky.perform (pytensor/tensor/linalg/decomposition/cholesky.py:48-83:
// LAPACK potrf, clean=True zeroes the other triangle, info != 0 => all-NaN result);
// SolveTriangular.perform (solvers/triangular.py:32-71, trtrs, NaN on info != 0);
// CholeskySolve.perform (solvers/psd.py:35-53, potrs = two triangular solves).
//
// MI355X mapping: on the hot path these are latency-bound (n = 128: 0.7 MFLOP), so the
// design goal is "one launch, everything on-chip".  A 128x128 fp64 matrix (128 KiB)
// fits the 160 KiB LDS of one CU:
//   potrf : one workgroup, blocked right-looking with 16-column panels:
//           (a) the 16x16 diagonal block is factored in REGISTERS by every wave
//               redundantly (lane i holds row i; cross-lane reads are v_readlane), so no
//               barrier or LDS round trip separates it from
//           (b) the panel solve X = A21 L11^-T, one matrix row per thread, L11 entries
//               broadcast from the wave's own registers, and
//           (c) the trailing update A22 -= X X^T as 16x16 tiles on the matrix cores
//               (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32), operands read from LDS.
//           (A variant that moved the L11 broadcasts of (a)/(b) from v_readlane to LDS was
//            3% faster alone and 2x slower next to the streaming kernel, whose butterfly
//            reductions keep the CU's LDS crossbar busy: the register broadcasts stay.)
//           Two barriers per panel (8 panels at n = 128) instead of three per column.
//   trsv  : matrix staged to LDS by the whole workgroup (coalesced), then ONE wave does the
//           substitution wave-synchronously (two rows per lane, pivots via v_readlane):
//           no barriers inside the n-step dependency chain.
// Larger matrices: blocked right-looking factorisation over HBM (chol_blocked: LDS-resident
// diagonal blocks, multi-workgroup panel solves, trailing updates on the MFMA GEMM).
// Batches (Blockwise) map to grid.x.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int NB = 16;  // panel width

// 16x16x4 matrix-core step, D += A B.  Operands: lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; result register r of lane l is D[drow(l, r)][l&15].
template <class T> struct Mfma16;
template <> struct Mfma16<double> {
  typedef double v4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ v4 run(double a, double b, v4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mfma16<float> {
  typedef float v4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ v4 run(float a, float b, v4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

__device__ __forceinline__ double hw_rcp(double x) { return __builtin_amdgcn_rcp(x); }  // v_rcp_f64: a seed, refine
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <class T> __device__ __forceinline__ T bcast_lane(T v, int src) {
  // wave-uniform broadcast of lane `src` (compile-time constant after unrolling)
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u;
    u.t = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
    return u.t;
  } else {
    union { T t; int i; } u;
    u.t = v;
    u.i = __builtin_amdgcn_readlane(u.i, src);
    return u.t;
  }
}

// Stage a dense n x n matrix (fast axis contiguous, slow stride n) from HBM into LDS through
// `put(a, c, v)` (a = slow index, c = fast index).  One workgroup cannot hide HBM latency with
// occupancy, so it keeps 16 x 16-byte loads per thread in flight: 64 KiB per round trip, two
// round trips for a 128 x 128 fp64 matrix (8-byte loads, 8 deep, needed eight and took ~40 us
// next to a kernel that saturates HBM).  `wide` = 16-byte path usable (n % VEC == 0, aligned).
template <class T, class F>
__device__ __forceinline__ void stage_dense(const T* __restrict__ src, int n, bool wide, F&& put,
                                            long long ld = 0) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int UN = 16;
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const int tid = threadIdx.x;
  const int total = n * n;
  if (ld == 0) ld = n;  // (a block of a larger matrix: slow stride ld > n)
  if (wide) {
    const int nv = total / VEC;
    for (int v0 = 0; v0 < nv; v0 += BLOCK * UN) {
      vec_t v[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        int idx = v0 + u * BLOCK + tid;
        if (idx >= nv) idx = nv - 1;
        const int e = idx * VEC;
        const int a = e / n;
        v[u] = *(const vec_t*)(src + (long long)a * ld + (e - a * n));
      }
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int idx = v0 + u * BLOCK + tid;
        if (idx < nv) {
          const int e = idx * VEC;
          const int a = e / n, c = e - a * n;
#pragma unroll
          for (int w = 0; w < VEC; w++) put(a, c + w, v[u][w]);
        }
      }
    }
  } else {
    for (int e0 = 0; e0 < total; e0 += BLOCK * 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int e = e0 + u * BLOCK + tid;
        if (e >= total) e = total - 1;
        const int a = e / n;
        v[u] = src[(long long)a * ld + (e - a * n)];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + u * BLOCK + tid;
        if (e < total) {
          const int a = e / n, c = e - a * n;
          put(a, c, v[u]);
        }
      }
    }
  }
}

// Wave-synchronous substitution T x = b with T resident in LDS (W, leading dimension ld).
// Called by ONE wave (64 lanes, RPL rows per lane, n <= 64*RPL); no barriers, no branches in
// the step.  The system is row-scaled once, in parallel (r'_i = b_i / t_ii, t'_ik = t_ik / t_ii),
// so the serial chain per column is only  v_readlane(r'_k) -> fma : x_k IS the scaled residual
// of row k.  Columns are fetched from LDS eight at a time, one block ahead of the block being
// eliminated, already masked to the strict triangle (select, not multiply: the unreferenced
// triangle may hold anything).
template <class T, int RPL, bool LOWER>
__device__ __forceinline__ void wave_trsv_dir(const T* __restrict__ W, int ld, int n,
                                              const T* __restrict__ b, T* __restrict__ x, int unit,
                                              bool poisoned) {
  constexpr int U = 8;
  const int lane = threadIdx.x & 63;
  T r[RPL], rd[RPL];
  bool sing = false;
#pragma unroll
  for (int q = 0; q < RPL; q++) {
    const int i = lane + 64 * q;
    const int ic = i < n ? i : n - 1;
    const T d = unit ? T(1) : W[ic * ld + ic];
    if (i < n && d == T(0)) sing = true;  // trtrs: exact singularity
    rd[q] = i < n ? T(1) / d : T(0);
    r[q] = i < n ? b[i] * rd[q] : T(0);
  }
  const bool fail = poisoned || __ballot(sing) != 0ull;
#pragma unroll
  for (int qq = 0; qq < RPL; qq++) {
    const int qb = LOWER ? qq : RPL - 1 - qq;  // 64-row block holding the pivots (compile time)
    const int kbeg = 64 * qb;
    const int kend = (kbeg + 64) < n ? (kbeg + 64) : n;
    const int cnt = kend - kbeg;  // pivots in this block (wave-uniform)
    if (cnt <= 0) continue;
    T cur[U][RPL], nxt[U][RPL];
    auto fetch = [&](T (&dst)[U][RPL], int kk0) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int kk = kk0 + u;
        const int kkc = kk < cnt ? kk : cnt - 1;
        const int k = LOWER ? kbeg + kkc : kend - 1 - kkc;
#pragma unroll
        for (int q = 0; q < RPL; q++) {
          if (LOWER ? q < qb : q > qb) { dst[u][q] = T(0); continue; }
          const int i = lane + 64 * q;
          const int ic = i < n ? i : n - 1;
          const T v = W[ic * ld + k] * rd[q];
          // only the pivot block straddles the diagonal; rows >= n hold rd = 0 and are never
          // broadcast or stored; pivots past the end of the block broadcast x_k = 0 instead
          dst[u][q] = (q != qb || (LOWER ? i > k : i < k)) ? v : T(0);
        }
      }
    };
    fetch(cur, 0);
    for (int kk0 = 0; kk0 < cnt; kk0 += U) {
      fetch(nxt, kk0 + U);  // (clamped and fully masked past the end of the block)
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int kk = kk0 + u;
        const int kkc = kk < cnt ? kk : cnt - 1;
        const int kl = LOWER ? kkc : cnt - 1 - kkc;  // pivot lane within the block
        T xk = bcast_lane(r[qb], kl);                 // wave-uniform lane: v_readlane
        if (kk >= cnt) xk = T(0);                      // (scalar select)
#pragma unroll
        for (int q = 0; q < RPL; q++)
          if (LOWER ? q >= qb : q <= qb) r[q] -= cur[u][q] * xk;
      }
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int q = 0; q < RPL; q++) cur[u][q] = nxt[u][q];
    }
  }
  const T nanv = (T)__builtin_nan("");
#pragma unroll
  for (int q = 0; q < RPL; q++) {
    const int i = lane + 64 * q;
    if (i < n) x[i] = fail ? nanv : r[q];
  }
}

template <class T, int RPL>
__device__ __forceinline__ void wave_trsv(const T* __restrict__ W, int ld, int n,
                                          const T* __restrict__ b, T* __restrict__ x, int lower,
                                          int unit, bool poisoned) {
  if (lower) wave_trsv_dir<T, RPL, true>(W, ld, n, b, x, unit, poisoned);
  else wave_trsv_dir<T, RPL, false>(W, ld, n, b, x, unit, poisoned);
}

// The factorisation proper on a lower-triangular working matrix W[i * ld + j] (i >= j) resident in
// LDS, by all BLOCK threads of the workgroup.  Ends on a workgroup barrier; returns "a pivot failed" as
// seen by this thread (every wave that takes part sees the same pivots).
// Per 16-column panel: (a) the elimination, one dependent chain per wave with NO barrier inside — lanes
// 0-15 of every wave hold the 16 diagonal rows (redundantly: each wave needs the pivots and the u_ck
// itself), lanes 16-63 hold 48 of the rows below, so the scaling of column k is at once the factorisation
// step and the panel solve; square-root-free (pivot by v_readlane, v_rcp + two Newton steps; columns scaled
// by 1/sqrt(pivot) afterwards), the next column's u_ck by v_readlane before the reciprocal is known, the
// others through a per-wave LDS column (see potrf64_wave below, the same chain: 2.0-2.4 us per panel
// against 4 us for the register factorisation + one-thread-per-row solve it replaces).  (b) the trailing
// update on the matrix cores.
template <class T>
__device__ __forceinline__ bool potrf_lds_core(T* __restrict__ W, const int ld, const int n) {
  __shared__ T s_pcol[BLOCK / 64][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  T* colbuf = s_pcol[wv];
  bool fail = false;
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jb = (n - j0) < NB ? (n - j0) : NB;
    const int m = n - j0 - jb;  // rows below the panel
    {
      const int below = 48 * wv + lane - NB;
      const bool diagl = lane < NB;
      const bool live = diagl ? (lane < jb) : (below < m);
      int row = diagl ? (j0 + lane) : (j0 + jb + below);
      row = row < n ? row : n - 1;
      if (wv == 0 || 48 * wv < m) {  // (wave-uniform: this wave has rows of the panel)
        T d[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) {
          d[c] = (live && c < jb && (!diagl || c <= lane)) ? W[row * ld + j0 + c] : T(0);
          if (diagl && !live && c == lane) d[c] = T(1);  // a short last panel continues with the identity
        }
        T cur[NB], lprev = T(0), pown = T(1);
#pragma unroll
        for (int c = 0; c < NB; c++) cur[c] = T(0);
#pragma unroll
        for (int k = 0; k < NB; k++) {
          if (k + 2 < NB) colbuf[lane] = d[k];
          const T pk = bcast_lane(d[k], k);
          T u1 = T(0);
          if (k + 1 < NB) u1 = bcast_lane(d[k], k + 1);
          T nxt[NB];
#pragma unroll
          for (int c = k + 2; c < NB; c++) nxt[c] = colbuf[c];
          if (!(pk > T(0))) fail = true;  // dpotf2: non-positive or NaN pivot (wave-uniform)
          pown = (lane == k) ? pk : pown;
          T r = hw_rcp(pk);
          r = __builtin_fma(r, __builtin_fma(-pk, r, T(1)), r);
          r = __builtin_fma(r, __builtin_fma(-pk, r, T(1)), r);
          const T lk = d[k] * r;
          if (k + 1 < NB) d[k + 1] -= lk * u1;
          if (k >= 1) {
#pragma unroll
            for (int c = k + 1; c < NB; c++) d[c] -= lprev * cur[c];
          }
#pragma unroll
          for (int c = k + 2; c < NB; c++) cur[c] = nxt[c];
          lprev = lk;
        }
        colbuf[lane] = rsqrt(pown);  // lane k < 16 owns pivot k; L = (unscaled columns) * diag(1/sqrt(p))
#pragma unroll
        for (int c = 0; c < NB; c++) d[c] *= colbuf[c];
        if (live && (!diagl || wv == 0)) {
#pragma unroll
          for (int c = 0; c < NB; c++)
            if (c < jb && (!diagl || c <= lane)) W[row * ld + j0 + c] = d[c];
        }
      }
    }
    __syncthreads();
    // ---- (c) trailing update A22 -= X X^T on the matrix cores: one 16x16 output tile of
    //      the lower triangle per wave and step, K = 16 = four 16x16x4 MFMAs.  (m > 0 only
    //      after a full panel, so K is always NB.)  Both operands are rows of the panel X.
    if (m > 0) {
      const int mt = (m + 15) >> 4;           // 16-row tiles per side
      const int ntiles = mt * (mt + 1) / 2;   // lower-triangular tile count
      const int base = j0 + jb;
      const int wave = tid >> 6;
      for (int tt = wave; tt < ntiles; tt += BLOCK / 64) {
        // unrank tt -> (ti >= tj)   (wave-uniform)
        int ti = (int)((sqrtf(8.0f * tt + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= tt) ti++;
        while (ti * (ti + 1) / 2 > tt) ti--;
        const int tj = tt - ti * (ti + 1) / 2;
        const int i0 = base + ti * 16, c0 = base + tj * 16;
        const int ri = (i0 + (lane & 15)) < n ? (i0 + (lane & 15)) : (n - 1);
        const int rj = (c0 + (lane & 15)) < n ? (c0 + (lane & 15)) : (n - 1);
        T af[4], bf[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          af[kk] = W[ri * ld + j0 + kk * 4 + (lane >> 4)];
          bf[kk] = W[rj * ld + j0 + kk * 4 + (lane >> 4)];
        }
        typename Mfma16<T>::v4 acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = Mfma16<T>::run(af[kk], bf[kk], acc);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = i0 + Mfma16<T>::drow(lane, r), j = c0 + (lane & 15);
          if (i < n && j < n && j <= i) W[i * ld + j] -= acc[r];
        }
      }
    }
    __syncthreads();
  }
  return fail;
}

// ---------------------------------------------------------------------------------
// blocked LDS-resident Cholesky (optionally followed by the first triangular solve with
// the factor still in LDS: rhs != NULL  =>  xout = L^-1 rhs, lower factor only)
// ---------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void potrf_lds_kernel(T* __restrict__ Lout,
                                                         const T* __restrict__ Ain, int n,
                                                         int lower, const T* __restrict__ rhs,
                                                         T* __restrict__ xout, long long ldio,
                                                         int* __restrict__ failflag) {
  // ldio != 0: the matrix is a diagonal block of a larger row-major working matrix (row stride
  // ldio) factored IN PLACE as one step of the blocked algorithm (chol_blocked below): only the
  // lower triangle is written back, a failure is reported through *failflag.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_fail;
  T* W = (T*)smem_raw;
  const int ld = n | 1;  // odd leading dimension: column walks are bank-conflict free
  const long long mat = blockIdx.x;
  const long long gld = ldio ? ldio : n;
  const T* A = Ain + mat * (long long)n * n;
  T* Lo = Lout + mat * (long long)n * n;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) s_fail = 0;
  // load the referenced triangle as a lower-triangular working matrix W[i][j], i >= j
  // (for `upper` the strict upper triangle is read transposed: LAPACK reads only `uplo`)
  {
    const bool wide = (n % (16 / (int)sizeof(T))) == 0 && (((size_t)A) & 15) == 0 && (gld % (16 / (int)sizeof(T))) == 0;
    if (lower)
      stage_dense<T>(A, n, wide, [&](int i, int j, T v) { if (i >= j) W[i * ld + j] = v; }, gld);
    else
      stage_dense<T>(A, n, wide, [&](int i, int j, T v) { if (j >= i) W[j * ld + i] = v; }, gld);
  }
  __syncthreads();
  const bool fail = potrf_lds_core<T>(W, ld, n);
  if (fail) s_fail = 1;  // benign race: every writer stores 1
  __syncthreads();
  const bool failed = s_fail != 0;
  const T nanv = (T)__builtin_nan("");
  if (ldio) {
    if (failed && tid == 0) atomicOr(failflag, 1);
#pragma unroll 8
    for (int e = tid; e < n * n; e += BLOCK) {
      const int i = e / n, j = e - i * n;
      if (i >= j) Lo[(long long)i * gld + j] = failed ? nanv : W[i * ld + j];
    }
    return;
  }
#pragma unroll 8
  for (int e = tid; e < n * n; e += BLOCK) {
    const int i = e / n, j = e - i * n;
    T v;
    if (failed) v = nanv;
    else if (lower) v = (i >= j) ? W[i * ld + j] : T(0);
    else v = (j >= i) ? W[j * ld + i] : T(0);
    Lo[e] = v;
  }
  // fused SolveTriangular(L, b): the lower factor is still in LDS — no second launch, no
  // re-staging of the 128 KiB matrix (the solve alone is ~5 us of a ~50-70 us trsv launch)
  if (rhs != nullptr && tid < 64)
    wave_trsv<T, 4>(W, ld, n, rhs + mat * (long long)n, xout + mat * (long long)n, 1, 0, failed);
}

// ---------------------------------------------------------------------------------
// Blocked Cholesky for matrices beyond one CU's LDS (n > 141 fp64 / 200 fp32): right-looking,
// NBK-column panels over a row-major working matrix Wk (lower triangle, row stride n) in HBM.
// Per panel k:  (1) the NBK x NBK diagonal block is factored in place by the LDS-resident kernel
// above (one CU; the serial column chain is the critical path of the whole factorisation);
// (2) the panel below it, X = A21 L11^-T, by chol_trsm_kernel — every workgroup takes CT_ROWS rows,
// one thread per row, L11 and the slab in LDS; (3) the trailing matrix A22 -= X X^T on the MFMA
// GEMM (gemm.hip, in place), by block columns so that only blocks touching the lower triangle
// are computed.  A failed pivot anywhere sets a device flag; the finishing kernel writes the
// requested triangle (transposed for `upper`), zeroes the other one, or NaN-fills everything
// (cholesky.py:78-80).  No host synchronisation: capturable into a hipGraph.
// ---------------------------------------------------------------------------------
constexpr int NBK = 64;       // panel width
constexpr int NBO = 512;      // outer panel: the trailing matrix is updated once per NBO columns
constexpr int CT_ROWS = 192;  // rows of the panel per workgroup (= threads of chol_trsm_kernel)

// W[i][j] (i >= j) <- the referenced triangle of A; W[i][j] (i < j) <- 0.  32x32 tiles through LDS:
// `upper` reads the tile of A transposed, both sides coalesced.
template <class T>
__global__ __launch_bounds__(BLOCK) void chol_stage_kernel(T* __restrict__ W, const T* __restrict__ A,
                                                          int n, int lower, int np) {
  // np >= n: edge (= row stride) of the working matrix; rows / columns past n continue the diagonal
  // with ones (the factor of diag(A, I) is diag(L, I): whole tiles for chol_dag_kernel)
  __shared__ T tile[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) {  // strictly upper block of W
    for (int e = threadIdx.x; e < 32 * 32; e += BLOCK) {
      const int i = bi * 32 + (e >> 5), j = bj * 32 + (e & 31);
      if (i < np && j < np) W[(long long)i * np + j] = T(0);
    }
    return;
  }
  // source tile: lower -> A[bi][bj]; upper -> A[bj][bi] read row-wise, used transposed
  const int si = lower ? bi : bj, sj = lower ? bj : bi;
  for (int e = threadIdx.x; e < 32 * 32; e += BLOCK) {
    const int r = e >> 5, c = e & 31;
    const int i = si * 32 + r, j = sj * 32 + c;
    tile[r][c] = (i < n && j < n) ? A[(long long)i * n + j] : T(0);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += BLOCK) {
    const int r = e >> 5, c = e & 31;
    const int i = bi * 32 + r, j = bj * 32 + c;
    if (i < n && j < n) W[(long long)i * np + j] = (i >= j) ? (lower ? tile[r][c] : tile[c][r]) : T(0);
    else if (i < np && j < np) W[(long long)i * np + j] = (i == j) ? T(1) : T(0);
  }
}

// X = A21 L11^-T for the rows below diagonal block k: W[r][k..k+nb) <- solution, r in [k+nb, n).
// L11 (lower, nb <= NBK) and this workgroup's CT_ROWS x nb slab live in LDS (slab transposed:
// thread t walks column t of Xs conflict-free, the entries of L11 are uniform-address reads);
// eight solution entries at a time in registers, as in trsm_lds_kernel.
template <class T>
__global__ __launch_bounds__(CT_ROWS) void chol_trsm_kernel(T* __restrict__ W, long long ld, int k, int nb, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int LL = NBK + 2;       // row stride of Ls: pairs of entries stay 16-byte aligned
  constexpr int XL = CT_ROWS + 1;   // row stride of Xs
  T* Ls = (T*)smem_raw;             // [NBK][LL]
  T* Xs = Ls + NBK * LL;            // [NBK][XL]
  T* Rd = Xs + NBK * XL;            // [NBK] reciprocal pivots
  const int tid = threadIdx.x;
  const long long r0 = (long long)k + nb + (long long)blockIdx.x * CT_ROWS;
  const int rows = (int)((n - r0) < CT_ROWS ? (n - r0) : CT_ROWS);
  // staging: one workgroup cannot hide memory latency with occupancy (three waves on the CU), so the
  // loads are issued in batches of 16 per thread before the first LDS store (a one-load-per-iteration
  // loop measured 52 us per panel at n = 2048 fp64 — 64 dependent round trips, profiles/r3f_chol2048_timeline.md)
  constexpr int UN = 16;
  for (int e0 = 0; e0 < NBK * NBK; e0 += CT_ROWS * UN) {  // (the pad rows/columns of a short last panel: zeros)
    T v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * CT_ROWS + tid;
      const int i = e / NBK, j = e - i * NBK;
      v[u] = (e < NBK * NBK && i < nb && j <= i) ? W[(long long)(k + i) * ld + k + j] : T(0);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * CT_ROWS + tid;
      if (e < NBK * NBK) Ls[(e / NBK) * LL + (e % NBK)] = v[u];
    }
  }
  if (tid < NBK) Rd[tid] = tid < nb ? T(1) / W[(long long)(k + tid) * ld + k + tid] : T(0);
  const int tot = rows * nb;
  for (int e0 = 0; e0 < tot; e0 += CT_ROWS * UN) {
    T v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * CT_ROWS + tid;
      const int r = e / nb, c = e - r * nb;
      v[u] = (e < tot) ? W[(r0 + r) * ld + k + c] : T(0);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * CT_ROWS + tid;
      const int r = e / nb, c = e - r * nb;
      if (e < tot) Xs[c * XL + r] = v[u];
    }
  }
  __syncthreads();
  if (tid < rows) {
    constexpr int RB = 8;
    typedef T T2 __attribute__((ext_vector_type(2)));
    for (int s0 = 0; s0 < nb; s0 += RB) {
      T acc[RB];
#pragma unroll
      for (int r = 0; r < RB; r++) acc[r] = (s0 + r < nb) ? Xs[(s0 + r) * XL + tid] : T(0);
      // columns already solved, four at a time (s0 is a multiple of 8): 4 own reads + 16 two-entry
      // reads of L feed 32 FMAs — one thread per row means the chain is latency-, not LDS-bound,
      // so the reads of a whole group are requested before its first FMA
      for (int c = 0; c < s0; c += 4) {
        const T x0 = Xs[c * XL + tid], x1 = Xs[(c + 1) * XL + tid], x2 = Xs[(c + 2) * XL + tid], x3 = Xs[(c + 3) * XL + tid];
        T2 la[RB], lb[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) {
          la[r] = *(const T2*)&Ls[(s0 + r) * LL + c];
          lb[r] = *(const T2*)&Ls[(s0 + r) * LL + c + 2];
        }
#pragma unroll
        for (int r = 0; r < RB; r++) {
          acc[r] -= x0 * la[r].x;
          acc[r] -= x1 * la[r].y;
          acc[r] -= x2 * lb[r].x;
          acc[r] -= x3 * lb[r].y;
        }
      }
#pragma unroll
      for (int r = 0; r < RB; r++) {
        if (s0 + r < nb) {
#pragma unroll
          for (int r2 = 0; r2 < r; r2++) acc[r] -= acc[r2] * Ls[(s0 + r) * LL + s0 + r2];
          acc[r] = acc[r] * Rd[s0 + r];  // reciprocal of the pivot, formed once per workgroup
          Xs[(s0 + r) * XL + tid] = acc[r];
        }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < rows * nb; e += CT_ROWS) {
    const int r = e / nb, c = e - r * nb;
    W[(r0 + r) * ld + k + c] = Xs[c * XL + r];
  }
}

// The same panel solve on the matrix cores (full panels, nb == NBK): blocked substitution over the
// four 16-column blocks of the panel,  X_j = (A_j - sum_{c<j} X_c L_jc^T) inv(L_jj)^T.  The sums and the
// product with the 16x16 inverse are v_mfma_*_16x16x4 steps on operands read from LDS; the inverses of
// the four diagonal blocks are formed once per workgroup by one wave (lane = block x column, forward
// substitution in registers).  Every 16-row tile of the slab belongs to ONE wave from the first barrier
// to the write-back, so the blocks need no further synchronisation (LDS operations of a wave complete
// in order).  The one-thread-per-row kernel above stays as the general path (short panels) and as
// the A/B reference (PTHIP_CHOL_TRSM=scalar).
constexpr int CM_ROWS = 192;  // rows per workgroup: 12 tiles, three per wave
template <class T>
__global__ __launch_bounds__(BLOCK) void chol_trsm_mfma_kernel(T* __restrict__ W, long long ld, int k, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int LLs = NBK + 1, XLs = NBK + 1;  // odd row strides: operand columns walk distinct banks
  T* Ls = (T*)smem_raw;              // [NBK][LLs]     L11
  T* Xs = Ls + NBK * LLs;            // [CM_ROWS][XLs] the slab, row-major
  T* Dv = Xs + CM_ROWS * XLs;        // [4][16][17]    inverses of the diagonal 16x16 blocks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  const long long r0 = (long long)k + NBK + (long long)blockIdx.x * CM_ROWS;
  const int rows = (int)((n - r0) < CM_ROWS ? (n - r0) : CM_ROWS);
  constexpr int UN = 16;
  for (int e0 = 0; e0 < NBK * NBK; e0 += BLOCK * UN) {
    T v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * BLOCK + tid;
      const int i = e / NBK, j = e - i * NBK;
      v[u] = (e < NBK * NBK && j <= i) ? W[(long long)(k + i) * ld + k + j] : T(0);
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * BLOCK + tid;
      if (e < NBK * NBK) Ls[(e / NBK) * LLs + (e % NBK)] = v[u];
    }
  }
  for (int e0 = 0; e0 < CM_ROWS * NBK; e0 += BLOCK * UN) {
    T v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * BLOCK + tid;
      const int r = e / NBK, c = e - r * NBK;
      v[u] = (r < rows) ? W[(r0 + r) * ld + k + c] : T(0);  // (rows past the matrix: zeros, never written back)
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int e = e0 + u * BLOCK + tid;
      const int r = e / NBK, c = e - r * NBK;
      if (e < CM_ROWS * NBK) Xs[r * XLs + c] = v[u];
    }
  }
  __syncthreads();
  if (wave == 0) {
    // lane = 16 * block + column: column `li` of inv(L_jj), j = lq, by forward substitution
    const T* Lj = Ls + (lq * 16) * LLs + lq * 16;
    T x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      T s = (i == li) ? T(1) : T(0);
#pragma unroll
      for (int q = 0; q < i; q++) s -= Lj[i * LLs + q] * x[q];
      x[i] = (i < li) ? T(0) : s / Lj[i * LLs + i];
    }
#pragma unroll
    for (int i = 0; i < 16; i++) Dv[(lq * 16 + i) * 17 + li] = x[i];  // Dinv_j[i][li]
  }
  __syncthreads();
  typedef typename Mfma16<T>::v4 v4;
  for (int j = 0; j < NBK / 16; j++) {
    for (int q = 0; q < CM_ROWS / 16 / 4; q++) {
      const int rt = wave + 4 * q;
      const T* xrow = Xs + (rt * 16 + li) * XLs;
      v4 acc = {T(0), T(0), T(0), T(0)};
      for (int c = 0; c < j; c++) {
#pragma unroll
        for (int st = 0; st < 4; st++)
          acc = Mfma16<T>::run(xrow[c * 16 + 4 * st + lq], Ls[(j * 16 + li) * LLs + c * 16 + 4 * st + lq], acc);
      }
      // R = A_j - sum: every lane updates the four entries of the tile it holds in the D layout
#pragma unroll
      for (int r = 0; r < 4; r++) Xs[(rt * 16 + Mfma16<T>::drow(lane, r)) * XLs + j * 16 + li] -= acc[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      v4 xo = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int st = 0; st < 4; st++)
        xo = Mfma16<T>::run(xrow[j * 16 + 4 * st + lq], Dv[(j * 16 + li) * 17 + 4 * st + lq], xo);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; r++) Xs[(rt * 16 + Mfma16<T>::drow(lane, r)) * XLs + j * 16 + li] = xo[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  for (int e = tid; e < rows * NBK; e += BLOCK) {
    const int r = e / NBK, c = e - r * NBK;
    W[(r0 + r) * ld + k + c] = Xs[r * XLs + c];
  }
}

// out <- the factor in the requested triangle (W holds it lower), zeros elsewhere; all-NaN when a
// pivot failed.  32x32 tiles through LDS so that the transposed write of `upper` is coalesced.
template <class T>
__global__ __launch_bounds__(BLOCK) void chol_finish_kernel(T* __restrict__ out, const T* __restrict__ W,
                                                           int n, int lower, const int* __restrict__ failflag, int np) {
  __shared__ T tile[32][33];
  const bool failed = *failflag != 0;
  const int bi = blockIdx.y, bj = blockIdx.x;
  const T nanv = (T)__builtin_nan("");
  const bool needs = lower ? (bj <= bi) : (bi <= bj);  // does this output block touch the triangle?
  if (failed || !needs) {
    for (int e = threadIdx.x; e < 32 * 32; e += BLOCK) {
      const int i = bi * 32 + (e >> 5), j = bj * 32 + (e & 31);
      if (i < n && j < n) out[(long long)i * n + j] = failed ? nanv : T(0);
    }
    return;
  }
  const int si = lower ? bi : bj, sj = lower ? bj : bi;  // block of W (lower storage) feeding this one
  for (int e = threadIdx.x; e < 32 * 32; e += BLOCK) {
    const int r = e >> 5, c = e & 31;
    const int i = si * 32 + r, j = sj * 32 + c;
    tile[r][c] = (i < n && j < n && i >= j) ? W[(long long)i * np + j] : T(0);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += BLOCK) {
    const int r = e >> 5, c = e & 31;
    const int i = bi * 32 + r, j = bj * 32 + c;
    if (i < n && j < n) out[(long long)i * n + j] = lower ? tile[r][c] : tile[c][r];
  }
}

// ---------------------------------------------------------------------------------
// The same factorisation as ONE persistent kernel over a graph of 64 x 64 tile tasks (default).
// The launch-per-step form below spends its time on the serial chain diagonal block -> panel solve ->
// update, three launches per 64 columns at 28 + 15 + 38 us (n = 4096 fp64: 5.2 of 6.0 ms,
// profiles/r3i_chol4096_kernel_stats.md): streams and graphs cannot express "the next diagonal block
// only needs ITS column updated".  Here task (i, j), i >= j, owns tile (i, j) of the lower triangle:
//   acc = sum_{k<j} L(i,k) L(j,k)^T       as the tiles of row i and row j become final (left-looking,
//                                          accumulators in registers, operands staged through LDS)
//   R   = A(i,j) - acc
//   i == j:  L(j,j) = chol(R) in LDS (potrf_lds_core) + the inverses of its four 16x16 diagonal blocks
//   i >  j:  L(i,j) = R L(j,j)^-T         (blocked substitution on the matrix cores, as chol_trsm_mfma_kernel)
// and publishes `done[i][j]` (release, agent scope).  Tasks are numbered column by column and dealt
// round-robin to the resident workgroups; each workgroup runs its tasks in that order, so every
// dependency of a task belongs to an earlier task and the oldest unfinished task can always run: no
// deadlock as long as all workgroups are resident (grid <= CUs x occupancy, checked by the host).
// A workgroup holding a task of column j+1 accumulates everything that is already final and then
// waits for the one tile the critical path still owes it — the look-ahead that the launch-per-step
// form lacks.  Waits are bounded: a wait that expires sets bit 4 of the device status word (loud
// RuntimeError on the host) instead of hanging the device.
// ---------------------------------------------------------------------------------
constexpr int DT = 64;         // tile edge
constexpr int DLS = DT + 1;    // LDS row stride of a staged tile
constexpr int DAG_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ int dag_flag(const int* f) {
  return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until tiles (i, k) and (j, k) are final; returns how many consecutive columns from k on are
// final already (>= 1), or -1 when the wait expired / another workgroup gave up.  Wave 0 polls 64
// columns per round trip; ends on a barrier + acquire so that every wave reads the published tiles.
__device__ __forceinline__ int dag_wait(const int* __restrict__ done, int nT, int i, int j, int k, int kend,
                                        int* __restrict__ abortflag, int* s_box) {
  if (threadIdx.x < 64) {
    const int kk = k + (int)threadIdx.x;
    int spins = 0, nready;
    for (;;) {
      int ok = 1;
      if (kk < kend) ok = dag_flag(done + (long long)i * nT + kk) & dag_flag(done + (long long)j * nT + kk);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(ok != 0);
      nready = (~m == 0ull) ? 64 : __builtin_ctzll(~m);
      if (nready > 0) break;
      if (++spins > DAG_SPIN_LIMIT || ((spins & 255) == 0 && dag_flag(abortflag))) { nready = -1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (threadIdx.x == 0) *s_box = nready;
  }
  __syncthreads();
  const int nr = *s_box;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return nr;
}

// profiling hook (PTHIP_CHOL_TRACE): thread 0 stamps the 100 MHz clock into slot `slot` of its task
__device__ __forceinline__ void dag_stamp(long long* tr, int slot) {
  if (tr && threadIdx.x == 0) tr[slot] = (long long)wall_clock64();
}

template <class T> struct DagTile {
  static constexpr int VEC = 16 / (int)sizeof(T);
  static constexpr int NV = DT * DT / VEC / BLOCK;  // 16-byte loads per thread and tile
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  vec_t v[NV];
  __device__ __forceinline__ void load(const T* __restrict__ src, long long ld) {
#pragma unroll
    for (int u = 0; u < NV; u++) {
      const int idx = u * BLOCK + (int)threadIdx.x;
      const int r = idx / (DT / VEC), c = (idx % (DT / VEC)) * VEC;
      v[u] = *(const vec_t*)(src + (long long)r * ld + c);
    }
  }
  __device__ __forceinline__ void store(T* __restrict__ dst) const {
#pragma unroll
    for (int u = 0; u < NV; u++) {
      const int idx = u * BLOCK + (int)threadIdx.x;
      const int r = idx / (DT / VEC), c = (idx % (DT / VEC)) * VEC;
#pragma unroll
      for (int w = 0; w < VEC; w++) dst[r * DLS + c + w] = v[u][w];
    }
  }
};

// acc1 += L(ra,k) L(rb,k)^T  (HEAD: also acc2 += L(ra,k) L(ra,k)^T) for k in [0, kend), each product as
// soon as both tiles are final; the next pair of tiles is in flight under the current product whenever it
// is already known to be final.  Returns false when a wait expired.
template <class T, bool HEAD>
__device__ __forceinline__ bool dag_accumulate(typename Mfma16<T>::v4 (&acc1)[2][2], typename Mfma16<T>::v4 (&acc2)[2][2],
                                               const T* __restrict__ W, long long ld, int nT, int ra, int rb, int kend,
                                               const int* __restrict__ done, int* __restrict__ abortflag, int* s_box,
                                               T* __restrict__ As, T* __restrict__ Bs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;  // this wave's 32 x 32 quadrant of the tile
  int kready = 0;
  bool have = false;
  DagTile<T> ta, tb;
  for (int k = 0; k < kend; k++) {
    if (!have) {
      if (k >= kready) {
        const int nr = dag_wait(done, nT, ra, rb, k, kend, abortflag, s_box);
        if (nr < 0) return false;
        kready = (k + nr) < kend ? (k + nr) : kend;
      }
      ta.load(W + (long long)ra * DT * ld + (long long)k * DT, ld);
      tb.load(W + (long long)rb * DT * ld + (long long)k * DT, ld);
    }
    ta.store(As);
    tb.store(Bs);
    __syncthreads();
    have = (k + 1) < kready;
    if (have) {
      ta.load(W + (long long)ra * DT * ld + (long long)(k + 1) * DT, ld);
      tb.load(W + (long long)rb * DT * ld + (long long)(k + 1) * DT, ld);
    }
#pragma unroll 4
    for (int st = 0; st < DT / 4; st++) {
      const T a0 = As[(r0 + li) * DLS + 4 * st + lq], a1 = As[(r0 + 16 + li) * DLS + 4 * st + lq];
      const T b0 = Bs[(c0 + li) * DLS + 4 * st + lq], b1 = Bs[(c0 + 16 + li) * DLS + 4 * st + lq];
      acc1[0][0] = Mfma16<T>::run(a0, b0, acc1[0][0]);
      acc1[0][1] = Mfma16<T>::run(a0, b1, acc1[0][1]);
      acc1[1][0] = Mfma16<T>::run(a1, b0, acc1[1][0]);
      acc1[1][1] = Mfma16<T>::run(a1, b1, acc1[1][1]);
      if constexpr (HEAD) {
        const T d0 = As[(c0 + li) * DLS + 4 * st + lq], d1 = As[(c0 + 16 + li) * DLS + 4 * st + lq];
        acc2[0][0] = Mfma16<T>::run(a0, d0, acc2[0][0]);
        acc2[0][1] = Mfma16<T>::run(a0, d1, acc2[0][1]);
        acc2[1][0] = Mfma16<T>::run(a1, d0, acc2[1][0]);
        acc2[1][1] = Mfma16<T>::run(a1, d1, acc2[1][1]);
      }
    }
    __syncthreads();
  }
  return true;
}

// X = R L^-T in place on the 64 x 64 tile As (L in Bs, the inverses of its diagonal 16 x 16 blocks in
// Dv), by 16-column blocks on the matrix cores; wave w owns rows 16w .. 16w+15 (no cross-wave traffic,
// LDS operations of a wave complete in order).  Caller brackets it with barriers.
template <class T>
__device__ __forceinline__ void dag_substitute(T* __restrict__ As, const T* __restrict__ Bs, const T* __restrict__ Dv) {
  typedef typename Mfma16<T>::v4 v4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const T* xrow = As + (wave * 16 + li) * DLS;
  for (int jb = 0; jb < DT / 16; jb++) {
    v4 s = {T(0), T(0), T(0), T(0)};
    for (int c = 0; c < jb; c++) {
#pragma unroll
      for (int st = 0; st < 4; st++)
        s = Mfma16<T>::run(xrow[c * 16 + 4 * st + lq], Bs[(jb * 16 + li) * DLS + c * 16 + 4 * st + lq], s);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) As[(wave * 16 + Mfma16<T>::drow(lane, r)) * DLS + jb * 16 + li] -= s[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    v4 xo = {T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int st = 0; st < 4; st++)
      xo = Mfma16<T>::run(xrow[jb * 16 + 4 * st + lq], Dv[(jb * 16 + li) * 17 + 4 * st + lq], xo);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; r++) As[(wave * 16 + Mfma16<T>::drow(lane, r)) * DLS + jb * 16 + li] = xo[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// this thread's 16 entries of tile (i, j) of the working matrix, in the accumulator layout
template <class T>
__device__ __forceinline__ void dag_load_acc_layout(T (&a)[2][2][4], const T* __restrict__ Aij, long long ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15;
  const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
#pragma unroll
  for (int x = 0; x < 2; x++)
#pragma unroll
    for (int y = 0; y < 2; y++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        a[x][y][r] = Aij[(long long)(r0 + x * 16 + Mfma16<T>::drow(lane, r)) * ld + c0 + y * 16 + li];
}

template <class T>
__device__ __forceinline__ void dag_residual_to_lds(T* __restrict__ As, const T (&a)[2][2][4],
                                                    const typename Mfma16<T>::v4 (&acc)[2][2]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15;
  const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
#pragma unroll
  for (int x = 0; x < 2; x++)
#pragma unroll
    for (int y = 0; y < 2; y++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        As[(r0 + x * 16 + Mfma16<T>::drow(lane, r)) * DLS + c0 + y * 16 + li] = a[x][y][r] - acc[x][y][r];
}

// wait for diagonal tile jd, then its factor -> Bs and the inverses of its diagonal blocks -> Dv
template <class T>
__device__ __forceinline__ bool dag_fetch_factor(const T* __restrict__ W, long long ld, int nT, int jd,
                                                 const T* __restrict__ Dinv, const int* __restrict__ done,
                                                 int* __restrict__ abortflag, int* s_box, T* __restrict__ Bs,
                                                 T* __restrict__ Dv) {
  if (dag_wait(done, nT, jd, jd, jd, jd + 1, abortflag, s_box) < 0) return false;
  const int tid = threadIdx.x;
  DagTile<T> tb;
  tb.load(W + (long long)jd * DT * ld + (long long)jd * DT, ld);
  const T* Dg = Dinv + (long long)jd * (4 * 16 * 16);
  T dv[4];
#pragma unroll
  for (int u = 0; u < 4; u++) dv[u] = Dg[u * BLOCK + tid];
  tb.store(Bs);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int e = u * BLOCK + tid;
    Dv[(e >> 4) * 17 + (e & 15)] = dv[u];
  }
  __syncthreads();
  return true;
}

// Tiles other workgroups will read are stored write-through (agent-scope stores): the release fence of
// dag_publish then only waits for these stores — a plain store leaves a dirty L2 line and the fence has to
// write the L2 back (3-4 us per publication with 32 workgroups of the same XCD writing tiles).
template <class T> __device__ __forceinline__ void dag_store(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class T>
__device__ __forceinline__ void dag_store_tile(T* __restrict__ Oij, long long ld, const T* __restrict__ As) {
#pragma unroll 4
  for (int e = threadIdx.x; e < DT * DT; e += BLOCK) {
    const int r = e >> 6, c = e & 63;
    dag_store<T>(Oij + (long long)r * ld + c, As[r * DLS + c]);
  }
}

// tile (rows i0.., columns c0..) of the diagonal tile takes the update of the 16-column panel at j0:
// Wt[i0.., c0..] -= X[i0..] X[c0..]^T, one wave, operands and result in LDS
template <class T>
__device__ __forceinline__ void potrf64_tile_update(T* __restrict__ Wt, int j0, int i0, int c0) {
  typedef typename Mfma16<T>::v4 v4;
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  v4 acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int kk = 0; kk < 4; kk++)
    acc = Mfma16<T>::run(Wt[(i0 + li) * DLS + j0 + 4 * kk + lq], Wt[(c0 + li) * DLS + j0 + 4 * kk + lq], acc);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = i0 + Mfma16<T>::drow(lane, r), jx = c0 + li;
    if (jx <= i) Wt[i * DLS + jx] -= acc[r];
  }
}

// Waves 2 and 3 of a head task, before their write-back duty: the trailing tiles that are NOT in the
// column block the next elimination reads — wave 2: (2,2) and (3,2) from panel 0; wave 3: (3,3) from
// panel 0, then from panel 1 — each as soon as its panel is final (`prog`), counted in upd[wave].
template <class T>
__device__ __forceinline__ void potrf64_trailing_helper(T* __restrict__ Wt, volatile int* prog, volatile int* upd, int wave) {
  const int lane = threadIdx.x & 63;
  while (*prog <= 0) __builtin_amdgcn_s_sleep(2);
  if (wave == 2) {
    potrf64_tile_update<T>(Wt, 0, 32, 32);
    potrf64_tile_update<T>(Wt, 0, 48, 32);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) upd[2] = 2;
  } else {
    potrf64_tile_update<T>(Wt, 0, 48, 48);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) upd[3] = 1;
    while (*prog <= 1) __builtin_amdgcn_s_sleep(2);
    potrf64_tile_update<T>(Wt, 16, 48, 48);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) upd[3] = 2;
  }
}

// The 64 x 64 diagonal tile in LDS (lower triangle of Wt[r * DLS + c]) factored by ONE wave, no barriers:
// lane r holds row j0 + r of the current 16-column panel — the 16 diagonal rows AND every row below —
// so that scaling column k by 1/sqrt(pivot) is at once the factorisation step and the panel solve, and
// the rank-1 update broadcasts each l_ck once (v_readlane) for all rows; the trailing 16 x 16 tiles are
// then updated on the matrix cores from LDS (at most six tiles: one wave's MFMA time is what two
// barriers would cost).  After each panel `*prog` advances: column block p of the factor (and, in rows
// DT .. DT+15, the transposed inverse of diagonal block p >= 1) is final, for the waves that write back.
template <class T>
__device__ __forceinline__ bool potrf64_wave(T* __restrict__ Wt, T* __restrict__ colbuf, volatile int* prog, volatile int* upd,
                                             long long* tr) {
  typedef typename Mfma16<T>::v4 v4;
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  bool fail = false;
#pragma unroll
  for (int p = 0; p < DT / 16; p++) {
    const int j0 = 16 * p;
    // lanes past the last row of the tile (panels 1-3): rows DT .. DT+15 of Wt, which hold the identity in
    // these 16 columns — "rows below" whose panel solve e_a L_pp^-T is row a of L_pp^-T: the inverse of the
    // diagonal block, which the substitution of every consumer needs, falls out of the elimination for free
    const int nreal = DT - j0;
    const int row = lane < nreal ? (j0 + lane) : (lane < nreal + 16 ? (DT + lane - nreal) : (DT - 1));
    const bool live = lane < nreal + (p > 0 ? 16 : 0);
    T d[16];
#pragma unroll
    for (int c = 0; c < 16; c++) d[c] = Wt[row * DLS + j0 + c];
#pragma unroll
    for (int c = 0; c < 16; c++) d[c] = (lane >= 16 || c <= lane) ? d[c] : T(0);
    // Square-root-free elimination (L D L^T; the columns are scaled by 1/sqrt(pivot) afterwards, all 16
    // at once): the loop is one dependent chain executed by a single wave, every instruction on it costs
    // its full latency, so the chain is kept to: pivot p = row k's entry k (v_readlane), r = 1/p
    // (v_rcp + two Newton steps), l = column * r, and the update d[c] -= l * u_ck of the later columns with
    // u_ck = entry k of row j0 + c — for the next column by v_readlane BEFORE the reciprocal is known, the others
    // through LDS (one 8-byte store of the unscaled column, uniform-address loads — the thirty v_readlane +
    // SGPR pairs per column of an all-register version made the loop instruction-bound), applied one
    // column later, under the next column's chain (a dependent fp64 op is 3.8 ns, an LDS round trip 53 ns,
    // tools/ubench/lat.hip).
    T cur[16], lprev = T(0), pown = T(1);
#pragma unroll
    for (int c = 0; c < 16; c++) cur[c] = T(0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k + 2 < 16) colbuf[lane] = d[k];
      const T pk = bcast_lane(d[k], k);
      T u1 = T(0);
      if (k + 1 < 16) u1 = bcast_lane(d[k], k + 1);
      T nxt[16];
#pragma unroll
      for (int c = k + 2; c < 16; c++) nxt[c] = colbuf[c];
      if (!(pk > T(0))) fail = true;  // dpotf2: non-positive or NaN pivot (wave-uniform)
      pown = (lane == k) ? pk : pown;
      T r = hw_rcp(pk);
      r = __builtin_fma(r, __builtin_fma(-pk, r, T(1)), r);
      r = __builtin_fma(r, __builtin_fma(-pk, r, T(1)), r);
      const T lk = d[k] * r;
      if (k + 1 < 16) d[k + 1] -= lk * u1;
      if (k >= 1) {
#pragma unroll
        for (int c = k + 1; c < 16; c++) d[c] -= lprev * cur[c];
      }
#pragma unroll
      for (int c = k + 2; c < 16; c++) cur[c] = nxt[c];
      lprev = lk;
    }
    {
      // L = (unscaled columns) * diag(1/sqrt(p)): lane k < 16 owns pivot k
      colbuf[lane] = rsqrt(pown);
#pragma unroll
      for (int c = 0; c < 16; c++) d[c] *= colbuf[c];
    }
    if (p == 0) dag_stamp(tr, 9);
    if (live) {  // (entries right of the diagonal in the 16 diagonal rows: never read again)
#pragma unroll
      for (int c = 0; c < 16; c++) Wt[row * DLS + j0 + c] = d[c];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *prog = p + 1;
    // trailing update, this wave's share: the tiles of the NEXT column block only (the one the next
    // elimination reads); the tiles further right belong to waves 2 and 3 (potrf64_trailing_helper), whose
    // earlier updates of the same tiles must have landed first
    if (p == 1) while (upd[2] < 2) __builtin_amdgcn_s_sleep(1);
    if (p == 2) while (upd[3] < 2) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int ti = 0; ti < DT / 16 - 1 - p; ti++) potrf64_tile_update<T>(Wt, j0, j0 + 16 + 16 * ti, j0 + 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (p < 3) dag_stamp(tr, 10 + p);
  }
  return fail;
}

// every thread's stores of the tile are visible device-wide before the flag is
__device__ __forceinline__ void dag_publish(int* flag, long long* tr = nullptr, int slot = 0) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  dag_stamp(tr, slot);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Tasks, in dependency order: for every tile column j the HEAD task H(j) — tiles (j, j-1) AND (j, j):
// the solve of the first tile below diagonal block j-1, the update of diagonal tile j with it straight
// from LDS, and the factorisation of that tile (the whole critical path of one column in ONE workgroup:
// separate tasks cost a store + publish + poll + reload + product = 11 of 37 us per column,
// profiles/r3t_chol_trace.txt) — followed by the ordinary tasks (i, j), i >= j + 2.
template <class T>
__global__ __launch_bounds__(BLOCK) void chol_dag_kernel(T* W, long long ld, const T* Asrc, long long lda, int nT,
                                                        int ntasks,
                                                        int* __restrict__ done, T* __restrict__ Dinv,
                                                        int* __restrict__ failflag, int* __restrict__ abortflag,
                                                        int* __restrict__ status, long long* __restrict__ trace) {
  // Asrc: where the tiles of A are read from — the working matrix itself (staged copy: Asrc == W, hence no
  // __restrict__ on either), or the caller's A when W is the output and needs no staging (lower factor, n a
  // multiple of the tile edge).
  // trace != NULL (PTHIP_CHOL_TRACE=<file>): sixteen 100 MHz timestamps per task, see tools/chol_trace.py
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_box, s_prog, s_arrive[4], s_upd[4];
  T* As = (T*)smem_raw;     // [DT + 16][DLS]  L(ra,k); then the tile being finished (+ 16 identity rows, potrf64_wave)
  T* Bs = As + (DT + 16) * DLS;    // [DT][DLS]  L(rb,k); then the factor of the diagonal tile above
  T* Dv = Bs + DT * DLS;    // [4][16][17] inverses of the diagonal 16x16 blocks of that factor
  T* s_rd = Dv + 4 * 16 * 17;  // [DT] reciprocals of that factor's diagonal
  T* s_col = s_rd + DT;        // [64] the column being eliminated (potrf64_wave)
  typedef typename Mfma16<T>::v4 v4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
#define DAG_STAMP(slot) dag_stamp(tr, slot)
#define DAG_GIVE_UP() do { if (tid == 0) { atomicOr(status, 16); __hip_atomic_store(abortflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } return; } while (0)
  // Tasks are CLAIMED, in order, through a ticket (flags[2], zeroed with the flags): a workgroup that is running takes
  // the lowest unclaimed task, and every task waits only on lower-numbered ones — which are therefore finished or in
  // the hands of a running workgroup.  Progress no longer depends on all gridDim.x workgroups being resident at once
  // (another process's kernels on the same device — PyMC's chains — can hold CUs for as long as they like); with the
  // static stride t = blockIdx.x + k*gridDim.x a workgroup the dispatcher had not started yet owned tasks the running
  // ones waited for until their spin limit.  The next ticket is requested when the current task is about to store its
  // tile: the round trip (an L2 atomic) hides behind the stores, and a task is not held by a workgroup that is still
  // busy (requested at the START of the current task, the head task of the next column — the critical path — could sit
  // behind a long update while other workgroups idled: n = 4096 went from 1.43 to 1.57 ms).
  __shared__ int s_task;
  int* const next_task = failflag + 2;
  int t_next = 0;
  if (tid == 0) t_next = atomicAdd(next_task, 1);
  for (;;) {
    if (tid == 0) s_task = t_next;
    __syncthreads();
    const int t = s_task;
    __syncthreads();  // (s_task is rewritten at the top of the next round)
    if (t >= ntasks) break;
    int j = 0, rem = t;
    for (;;) {
      const int cj = 1 + ((nT - j - 2) > 0 ? (nT - j - 2) : 0);
      if (rem < cj) break;
      rem -= cj;
      j++;
    }
    const bool head = rem == 0;
    const int i = head ? j : j + 1 + rem;
    long long* tr = trace ? trace + (long long)t * 16 : nullptr;
    DAG_STAMP(0);
    v4 acc1[2][2], acc2[2][2];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
      for (int y = 0; y < 2; y++) acc1[x][y] = acc2[x][y] = v4{T(0), T(0), T(0), T(0)};
    if (!head) {
      // ---- ordinary task: L(i,j) = (A(i,j) - sum_k L(i,k) L(j,k)^T) L(j,j)^-T
      T a[2][2][4];
      dag_load_acc_layout<T>(a, Asrc + (long long)i * DT * lda + (long long)j * DT, lda);
      if (!dag_accumulate<T, false>(acc1, acc2, W, ld, nT, i, j, j, done, abortflag, &s_box, As, Bs)) DAG_GIVE_UP();
      DAG_STAMP(1);
      dag_residual_to_lds<T>(As, a, acc1);
      if (!dag_fetch_factor<T>(W, ld, nT, j, Dinv, done, abortflag, &s_box, Bs, Dv)) DAG_GIVE_UP();
      DAG_STAMP(2);
      dag_substitute<T>(As, Bs, Dv);
      __syncthreads();
      DAG_STAMP(3);
      if (tid == 0) t_next = atomicAdd(next_task, 1);
      dag_store_tile<T>(W + (long long)i * DT * ld + (long long)j * DT, ld, As);
      dag_publish(done + (long long)i * nT + j, tr, 6);
      DAG_STAMP(15);
      continue;
    }
    // ---- head task of column j
    T ad[2][2][4];
    dag_load_acc_layout<T>(ad, Asrc + (long long)j * DT * lda + (long long)j * DT, lda);
    if (j > 0) {
      T as_[2][2][4];
      dag_load_acc_layout<T>(as_, Asrc + (long long)j * DT * lda + (long long)(j - 1) * DT, lda);
      if (!dag_accumulate<T, true>(acc1, acc2, W, ld, nT, j, j - 1, j - 1, done, abortflag, &s_box, As, Bs)) DAG_GIVE_UP();
      DAG_STAMP(1);
      dag_residual_to_lds<T>(As, as_, acc1);
      if (!dag_fetch_factor<T>(W, ld, nT, j - 1, Dinv, done, abortflag, &s_box, Bs, Dv)) DAG_GIVE_UP();
      DAG_STAMP(2);
      dag_substitute<T>(As, Bs, Dv);
      __syncthreads();
      DAG_STAMP(3);
      // L(j,j-1) on its way to memory (waves 1-3; wave 0 keeps no store in flight: it goes straight on to the
      // factorisation, the others retire their stores and publish the tile behind its back); meanwhile the
      // diagonal tile takes its last update from LDS
      if (wave != 0) {
        T* Ox = W + (long long)j * DT * ld + (long long)(j - 1) * DT;
#pragma unroll 4
        for (int e = tid - 64; e < DT * DT; e += BLOCK - 64) dag_store<T>(Ox + (long long)(e >> 6) * ld + (e & 63), As[(e >> 6) * DLS + (e & 63)]);
      }
      DAG_STAMP(4);
#pragma unroll 4
      for (int st = 0; st < DT / 4; st++) {
        const T a0 = As[(r0 + li) * DLS + 4 * st + lq], a1 = As[(r0 + 16 + li) * DLS + 4 * st + lq];
        const T d0 = As[(c0 + li) * DLS + 4 * st + lq], d1 = As[(c0 + 16 + li) * DLS + 4 * st + lq];
        acc2[0][0] = Mfma16<T>::run(a0, d0, acc2[0][0]);
        acc2[0][1] = Mfma16<T>::run(a0, d1, acc2[0][1]);
        acc2[1][0] = Mfma16<T>::run(a1, d0, acc2[1][0]);
        acc2[1][1] = Mfma16<T>::run(a1, d1, acc2[1][1]);
      }
      DAG_STAMP(5);
      DAG_STAMP(6);
      __syncthreads();  // every wave is done reading As
      DAG_STAMP(7);
    }
    dag_residual_to_lds<T>(As, ad, acc2);
    for (int e = tid; e < 16 * DT; e += BLOCK) {  // rows DT .. DT+15: the identity under column blocks 1-3
      const int a = e >> 6, c = e & 63;
      As[(DT + a) * DLS + c] = (c >= 16 && (c & 15) == a) ? T(1) : T(0);
    }
    if (tid < 4) { s_prog = 0; s_arrive[tid] = 0; s_upd[tid] = 0; }
    __syncthreads();
    DAG_STAMP(8);
    if (j > 0 && wave != 0) {
      // (plain LDS words, no LDS atomic: an atomic through a generic pointer to LDS trips the gfx950 back end)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      volatile int* arrive = s_arrive;
      if (lane == 0) arrive[wave] = 1;
      if (wave == 1) {
        while (!(arrive[2] && arrive[3])) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(done + (long long)j * nT + (j - 1), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    {
      volatile int* prog = &s_prog;
      T* Ojj = W + (long long)j * DT * ld + (long long)j * DT;
      T* Dg = Dinv + (long long)j * (4 * 16 * 16);
      if (wave == 0) {
        const bool fail = potrf64_wave<T>(As, s_col, prog, s_upd, tr);
        if (fail && lane == 0) atomicOr(failflag, 1);
        DAG_STAMP(13);
      } else {
        if (wave == 1) {
          // diagonal block 0 has no spare lanes in the elimination: its inverse here, under panels 1-3
          // (column c of inv(L_00) by forward substitution, lane c < 16; right-looking: 16 steps of mul + fma)
          while (*prog <= 0) __builtin_amdgcn_s_sleep(8);
          const T dg = As[li * DLS + li];
          T rr = hw_rcp(dg);
          rr = __builtin_fma(rr, __builtin_fma(-dg, rr, T(1)), rr);
          rr = __builtin_fma(rr, __builtin_fma(-dg, rr, T(1)), rr);
          s_rd[li] = rr;  // (the four 16-lane groups store the same value)
          T x[16];
#pragma unroll
          for (int r = 0; r < 16; r++) x[r] = (r == li) ? T(1) : T(0);
#pragma unroll
          for (int q = 0; q < 16; q++) {
            x[q] *= s_rd[q];
#pragma unroll
            for (int r = q + 1; r < 16; r++) x[r] -= As[r * DLS + q] * x[q];
          }
          if (lane < 16) {
#pragma unroll
            for (int r = 0; r < 16; r++) dag_store<T>(Dg + r * 16 + li, x[r]);
          }
        }
        if (wave >= 2) potrf64_trailing_helper<T>(As, prog, s_upd, wave);
        // column block b of the factor, and the inverse of diagonal block b >= 1 (rows DT.. of As hold it
        // transposed), back to memory as soon as panel b is final
        const int t2 = tid - 64;
        for (int bq = 0; bq < DT / 16; bq++) {
          while (*prog <= bq) __builtin_amdgcn_s_sleep(8);
          const int nel = (DT - 16 * bq) * 16;
          for (int e = t2; e < nel; e += BLOCK - 64) {
            const int r = 16 * bq + (e >> 4), c = 16 * bq + (e & 15);
            if (c <= r) dag_store<T>(Ojj + (long long)r * ld + c, As[r * DLS + c]);
          }
          if (bq > 0) {
            for (int e = t2; e < 256; e += BLOCK - 64) {
              const int r = e >> 4, c = e & 15;  // Dinv_b[r][c] = (L_bb^-T)[c][r]
              dag_store<T>(Dg + (bq * 16 + r) * 16 + c, As[(DT + c) * DLS + 16 * bq + r]);
            }
          }
        }
      }
    }
    if (tid == 0) t_next = atomicAdd(next_task, 1);
    dag_publish(done + (long long)j * nT + j, tr, 14);
    DAG_STAMP(15);
  }
#undef DAG_STAMP
#undef DAG_GIVE_UP
}

// out[0 .. count) <- NaN when *flag is set (a failed pivot poisons the whole result)
template <class T> __global__ void nan_fill_if_kernel(T* __restrict__ out, long long count, const int* __restrict__ flag) {
  if (*flag == 0) return;
  const T nanv = (T)__builtin_nan("");
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long long)gridDim.x * blockDim.x) out[e] = nanv;
}

// strict upper triangle of an n x n row-major matrix <- 0 (32 x 32 tiles): the lower factor written in place
// by chol_dag_kernel never touches it
template <class T>
__global__ __launch_bounds__(BLOCK) void zero_upper_kernel(T* __restrict__ L, int n) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj < bi) return;
  for (int e = threadIdx.x; e < 32 * 32; e += BLOCK) {
    const int i = bi * 32 + (e >> 5), j = bj * 32 + (e & 31);
    if (i < n && j < n && j > i) L[(long long)i * n + j] = T(0);
  }
}

template <class T>
int chol_dag(int lower, long long n, const T* A, T* L) {
  hipStream_t st = pthip::ctx().stream;
  const long long np = (n + DT - 1) / DT * DT;
  // lower factor of a matrix of whole tiles: factor straight into the output (no staged copy, no finishing pass)
  static const bool no_direct = getenv("PTHIP_CHOL_STAGED") != nullptr;
  const bool direct = lower && np == n && !no_direct && (const void*)A != (const void*)L;
  const int nT = (int)(np / DT);
  const size_t wbytes = direct ? 0 : (size_t)np * np * sizeof(T);
  const size_t dbytes = (size_t)nT * 4 * 16 * 16 * sizeof(T);
  const size_t fbytes = ((size_t)nT * nT * sizeof(int) + 255) / 256 * 256 + 256;
  void* scratch = nullptr;
  int r = pthip_alloc(wbytes + dbytes + fbytes, &scratch);
  if (r) return r;
  T* W = direct ? L : (T*)scratch;
  T* Dinv = (T*)((char*)scratch + wbytes);
  int* flags = (int*)((char*)scratch + wbytes + dbytes);
  int* failflag = flags, *abortflag = flags + 1, *done = flags + 64;
  auto fail = [&](int rc) { pthip_free(scratch); return rc; };
  auto kk = chol_dag_kernel<T>;
  const size_t lds = (size_t)((2 * DT + 16) * DLS + 4 * 16 * 17 + DT + 64) * sizeof(T);
  static int resident = 0;  // workgroups of this kernel the device holds at once
  if (!resident) {
    if (lds > 64 * 1024)
      if (hipError_t e = hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess)
        return fail(pthip::check(e, "chol_dag attribute"));
    int per_cu = 0, dev = 0, cus = 0;
    if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kk, BLOCK, lds); e != hipSuccess)
      return fail(pthip::check(e, "chol_dag occupancy"));
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return fail(pthip::check(e, "chol_dag device"));
    if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess)
      return fail(pthip::check(e, "chol_dag device attribute"));
    if (per_cu > 2) per_cu = 2;
    resident = per_cu * cus;
    if (resident <= 0) return fail(pthip::check(hipErrorInvalidValue, "chol_dag: kernel does not fit the device"));
  }
  if (hipError_t e = pthip::memset_async(flags, 0, fbytes, st); e != hipSuccess) return fail(pthip::check(e, "chol flags memset"));
  const unsigned ntile = (unsigned)(np / 32);
  if (direct) {
    PTHIP_KLAUNCH((zero_upper_kernel<T>), dim3(ntile, ntile), dim3(BLOCK), 0, st, L, (int)n);
    if ((r = pthip::post_launch("chol zero upper"))) return fail(r);
  } else {
    PTHIP_KLAUNCH((chol_stage_kernel<T>), dim3(ntile, ntile), dim3(BLOCK), 0, st, W, A, (int)n, lower, (int)np);
    if ((r = pthip::post_launch("chol_stage"))) return fail(r);
  }
  int ntasks = 0;  // per column: the head task + the tiles from two below the diagonal down
  for (int j = 0; j < nT; j++) ntasks += 1 + ((nT - j - 2) > 0 ? (nT - j - 2) : 0);
  const unsigned grid = (unsigned)(ntasks < resident ? ntasks : resident);
  long long* trace = nullptr;
  static const char* trace_path = getenv("PTHIP_CHOL_TRACE");
  if (trace_path && hipMalloc((void**)&trace, (size_t)ntasks * 16 * sizeof(long long)) != hipSuccess) trace = nullptr;
  PTHIP_KLAUNCH(kk, dim3(grid), dim3(BLOCK), lds, st, W, np, direct ? A : (const T*)W, np, nT, ntasks, done, Dinv, failflag, abortflag, pthip::ctx().status_dev, trace);
  if ((r = pthip::post_launch("chol_dag"))) return fail(r);
  if (trace) {  // profiling hook only: synchronises
    std::vector<long long> h((size_t)ntasks * 16);
    if (hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(h.data(), trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
      if (FILE* f = fopen(trace_path, "wb")) {
        const long long hdr[2] = {nT, (long long)grid};
        fwrite(hdr, sizeof(long long), 2, f);
        fwrite(h.data(), sizeof(long long), h.size(), f);
        fclose(f);
      }
    }
    (void)hipFree(trace);
  }
  if (direct) {
    PTHIP_KLAUNCH((nan_fill_if_kernel<T>), dim3(256), dim3(BLOCK), 0, st, L, n * n, (const int*)failflag);
    r = pthip::post_launch("chol nan fill");
  } else {
    const unsigned nt = (unsigned)((n + 31) / 32);
    PTHIP_KLAUNCH((chol_finish_kernel<T>), dim3(nt, nt), dim3(BLOCK), 0, st, L, (const T*)W, (int)n, lower, (const int*)failflag, (int)np);
    r = pthip::post_launch("chol_finish");
  }
  pthip_free(scratch);  // stream-ordered reuse keeps this safe
  return r;
}

template <class T>
int chol_blocked(int lower, long long n, const T* A, T* L) {
  static const bool steps = getenv("PTHIP_CHOL") && !strcmp(getenv("PTHIP_CHOL"), "steps");
  if (!steps && !pthip::ctx().safe_mode) return chol_dag<T>(lower, n, A, L);
  hipStream_t st = pthip::ctx().stream;
  void* scratch = nullptr;
  const size_t wbytes = (size_t)n * n * sizeof(T);
  int r = pthip_alloc(wbytes + 256, &scratch);
  if (r) return r;
  T* W = (T*)scratch;
  int* flag = (int*)((char*)scratch + wbytes);
  const int dt = sizeof(T) == 8 ? PTHIP_F64 : PTHIP_F32;
  const unsigned nt = (unsigned)((n + 31) / 32);
  auto fail = [&](int rc) { pthip_free(scratch); return rc; };
  if (hipError_t e = pthip::memset_async(flag, 0, 256, st); e != hipSuccess) return fail(pthip::check(e, "chol flag memset"));
  PTHIP_KLAUNCH((chol_stage_kernel<T>), dim3(nt, nt), dim3(BLOCK), 0, st, W, A, (int)n, lower, (int)n);
  if ((r = pthip::post_launch("chol_stage"))) return fail(r);
  auto kd = potrf_lds_kernel<T>;
  auto kt = chol_trsm_kernel<T>;
  const size_t lds_d = (size_t)NBK * (NBK | 1) * sizeof(T);
  const size_t lds_t = (size_t)NBK * (NBK + 2 + CT_ROWS + 1 + 1) * sizeof(T);
  auto km = chol_trsm_mfma_kernel<T>;
  const size_t lds_m = (size_t)(NBK * (NBK + 1) + CM_ROWS * (NBK + 1) + 4 * 16 * 17) * sizeof(T);
  static const bool use_mfma = !(getenv("PTHIP_CHOL_TRSM") && !strcmp(getenv("PTHIP_CHOL_TRSM"), "scalar"));
  static bool attr_m = false;
  if (!attr_m && lds_m > 64 * 1024) {
    if (hipError_t e = hipFuncSetAttribute((const void*)km, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m); e != hipSuccess)
      return fail(pthip::check(e, "chol_trsm_mfma attribute"));
    attr_m = true;
  }
  static bool attr_t = false;
  if (!attr_t && lds_t > 64 * 1024) {
    if (hipError_t e = hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t); e != hipSuccess)
      return fail(pthip::check(e, "chol_trsm attribute"));
    attr_t = true;
  }
  // Two levels: NBO-column outer panels.  Inside one, every NBK-column step updates only the rest of
  // the outer panel (a tall m x <=NBO strip, K = NBK); the matrix to the right of the outer panel is
  // updated ONCE per outer panel with K = NBO — the GEMM runs at its large-K rate and the trailing
  // matrix crosses HBM n/NBO times instead of n/NBK times (n = 4096, one level: 220 launches of a
  // K = 64 update at 36 us each = 58 % of the factorisation, profiles/r3c_chol4096_kernel_stats.md).
  for (long long K0 = 0; K0 < n; K0 += NBO) {
    const long long Kend = (K0 + NBO < n) ? K0 + NBO : n;
    for (long long k = K0; k < Kend; k += NBK) {
      const int nb = (int)((Kend - k) < NBK ? (Kend - k) : NBK);
      T* D = W + k * n + k;
      PTHIP_KLAUNCH(kd, dim3(1), dim3(BLOCK), lds_d, st, D, (const T*)D, nb, 1, (const T*)nullptr, (T*)nullptr, (long long)n, flag);
      if ((r = pthip::post_launch("chol diag"))) return fail(r);
      const long long m = n - k - nb;
      if (m <= 0) break;
      if (nb == NBK && use_mfma) {
        PTHIP_KLAUNCH(km, dim3((unsigned)((m + CM_ROWS - 1) / CM_ROWS)), dim3(BLOCK), lds_m, st, W, n, (int)k, (int)n);
      } else {
        PTHIP_KLAUNCH(kt, dim3((unsigned)((m + CT_ROWS - 1) / CT_ROWS)), dim3(CT_ROWS), lds_t, st, W, n, (int)k, nb, (int)n);
      }
      if ((r = pthip::post_launch("chol trsm"))) return fail(r);
      const long long ncol = Kend - (k + nb);  // columns of the outer panel still to come
      if (ncol > 0) {
        const T* P = W + (k + nb) * n + k;  // the panel just solved: m x nb, row stride n
        r = pthip::gemm_inplace(dt, m, ncol, nb, -1.0, P, n, 1, P, 1, n, 1.0, W + (k + nb) * n + (k + nb), n);
        if (r) return fail(r);
      }
    }
    const long long mo = n - Kend;
    if (mo <= 0) break;
    // the matrix right of / below the outer panel, lower block triangle only: block columns of width cw
    const long long wo = Kend - K0;
    const T* Lp = W + Kend * n + K0;  // mo x wo, row stride n
    long long cw = (mo / 4 + 127) / 128 * 128;
    if (cw < 512) cw = 512;
    for (long long c0 = 0; c0 < mo; c0 += cw) {
      const long long wN = (mo - c0) < cw ? (mo - c0) : cw;
      r = pthip::gemm_inplace(dt, mo - c0, wN, wo, -1.0, Lp + c0 * n, n, 1, Lp + c0 * n, 1, n, 1.0,
                              W + (Kend + c0) * n + (Kend + c0), n);
      if (r) return fail(r);
    }
  }
  PTHIP_KLAUNCH((chol_finish_kernel<T>), dim3(nt, nt), dim3(BLOCK), 0, st, L, (const T*)W, (int)n, lower, (const int*)flag, (int)n);
  r = pthip::post_launch("chol_finish");
  pthip_free(scratch);  // stream-ordered reuse keeps this safe
  return r;
}

// ---------------------------------------------------------------------------------
// triangular solve, one right-hand side: wave-synchronous substitution from LDS
// ---------------------------------------------------------------------------------
// T accessed through element strides (sT0, sT1): a transposed solve = swapped strides +
// flipped `lower`.  RPL rows per lane (n <= 64*RPL).
template <class T, int RPL>
__global__ __launch_bounds__(BLOCK) void trsv_lds_kernel(T* __restrict__ Xout,
                                                        const T* __restrict__ Tm, long long sTb,
                                                        long long sT0, long long sT1,
                                                        const T* __restrict__ B, long long sBb,
                                                        int n, int lower, int unit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* W = (T*)smem_raw;
  const int ld = n | 1;
  const long long mat = blockIdx.x;
  const T* Tg = Tm + mat * sTb;
  const T* b = B + mat * sBb;
  T* x = Xout + mat * (long long)n;
  // stage T with the unit-stride axis across lanes (coalesced for either orientation)
  const bool rowmaj = (sT1 == 1 || sT0 != 1);
  const bool dense = rowmaj ? (sT1 == 1 && sT0 == n) : (sT0 == 1 && sT1 == n);
  if (dense || n == 1) {
    const bool wide = (n % (16 / (int)sizeof(T))) == 0 && (((size_t)Tg) & 15) == 0;
    if (rowmaj) stage_dense<T>(Tg, n, wide, [&](int i, int j, T v) { W[i * ld + j] = v; });
    else stage_dense<T>(Tg, n, wide, [&](int j, int i, T v) { W[i * ld + j] = v; });
  } else {
    for (int e0 = 0; e0 < n * n; e0 += BLOCK * 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + u * BLOCK + threadIdx.x;
        const int a = e / n, c = e - a * n;  // a: slow index, c: fast (unit-stride) index
        const int i = rowmaj ? a : c, j = rowmaj ? c : a;
        v[u] = e < n * n ? Tg[i * sT0 + j * sT1] : T(0);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int e = e0 + u * BLOCK + threadIdx.x;
        const int a = e / n, c = e - a * n;
        const int i = rowmaj ? a : c, j = rowmaj ? c : a;
        if (e < n * n) W[i * ld + j] = v[u];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;  // one wave solves; no barriers below
  wave_trsv<T, RPL>(W, ld, n, b, x, lower, unit, false);
}

// a FEW right-hand sides (<= 16: the matrix solve of a multi-response prior, `solve_triangular(L, B)` with B (K, R) —
// pytensor/tensor/linalg/solvers/triangular.py:41-66, and its gradient's solve), triangle resident in LDS: one WAVE per
// column with the wave-synchronous substitution of the vector kernel (wave_trsv: a 128-row solve is ~5 us of dependent
// v_readlane -> fma steps), the four waves taking columns w, w + 4, ...  One thread per column (trsm_lds_kernel below) is
// n^2 / 2 dependent steps on eight lanes: 160-230 us at n = 128, R = 8 (profiles/r7_wide200_gemm_*).
template <class T, int RPL>
__global__ __launch_bounds__(BLOCK) void trsm_few_lds_kernel(T* Xout, const T* __restrict__ Tm, long long sTb, long long sT0, long long sT1,
                                                            const T* B, long long sBb, int n, int nrhs, int lower, int unit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* W = (T*)smem_raw;
  const int ld = n | 1;
  T* cols = W + (size_t)n * ld;  // [BLOCK / 64][2][n]: a wave's right-hand side and solution, contiguous
  const long long mat = blockIdx.x;
  const T* Tg = Tm + mat * sTb;
  const T* b = B + mat * sBb;
  T* x = Xout + mat * (long long)n * nrhs;
  for (int e0 = 0; e0 < n * n; e0 += BLOCK * 8) {
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + u * BLOCK + threadIdx.x;
      const int ec = e < n * n ? e : n * n - 1;
      const int i = ec / n, j = ec - i * n;
      v[u] = Tg[i * sT0 + j * sT1];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int e = e0 + u * BLOCK + threadIdx.x;
      if (e < n * n) {
        const int i = e / n, j = e - i * n;
        W[i * ld + j] = v[u];
      }
    }
  }
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T* bw = cols + (size_t)wv * 2 * n;
  T* xw = bw + n;
  for (int c = wv; c < nrhs; c += BLOCK / 64) {  // (wave-uniform: no barriers below, LDS traffic of a wave is in order)
    for (int i = lane; i < n; i += 64) bw[i] = b[(long long)i * nrhs + c];
    wave_trsv<T, RPL>(W, ld, n, bw, xw, lower, unit, false);
    for (int i = lane; i < n; i += 64) x[(long long)i * nrhs + c] = xw[i];
  }
}

// many right-hand sides, triangle resident in LDS: one thread per rhs column, eight rows of the
// solution at a time in registers.  The triangle entries are uniform-address (broadcast) LDS
// reads, the already-solved x_j are coalesced loads across the columns and are reused for eight
// rows; the 8x8 diagonal block is substituted in registers.  (MatrixInverse / Solve with a
// matrix rhs: the generic kernel below re-read T and x from global memory for every term.)
template <class T>
__global__ __launch_bounds__(BLOCK) void trsm_lds_kernel(T* Xout,
                                                        const T* __restrict__ Tm, long long sTb,
                                                        long long sT0, long long sT1,
                                                        const T* B, long long sBb,
                                                        int n, int nrhs, int lower, int unit,
                                                        int* __restrict__ failflag) {
  // (B may be Xout: the blocked solve below runs in place; failflag != NULL: a zero pivot is reported
  // there and the caller poisons the whole result)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* W = (T*)smem_raw;
  const int ld = n | 1;
  const long long mat = blockIdx.y;
  const T* Tg = Tm + mat * sTb;
  const T* b = B + mat * sBb;
  T* x = Xout + mat * (long long)n * nrhs;
  for (int e = threadIdx.x; e < n * n; e += BLOCK) {
    const int i = e / n, j = e - i * n;
    W[i * ld + j] = Tg[i * sT0 + j * sT1];
  }
  __syncthreads();
  const int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= nrhs) return;
  constexpr int RB = 8;
  bool fail = false;
  for (int s0 = 0; s0 < n; s0 += RB) {
    T acc[RB];
    int row[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const int p = (s0 + r) < n ? (s0 + r) : (n - 1);
      row[r] = lower ? p : n - 1 - p;
      acc[r] = b[(long long)row[r] * nrhs + c];
    }
    for (int q = 0; q < s0; q++) {
      const int j = lower ? q : n - 1 - q;
      const T xj = x[(long long)j * nrhs + c];
#pragma unroll
      for (int r = 0; r < RB; r++) acc[r] -= W[row[r] * ld + j] * xj;
    }
#pragma unroll
    for (int r = 0; r < RB; r++) {
      if (s0 + r < n) {
#pragma unroll
        for (int r2 = 0; r2 < r; r2++) acc[r] -= W[row[r] * ld + row[r2]] * acc[r2];
        const T d = unit ? T(1) : W[row[r] * ld + row[r]];
        if (d == T(0)) fail = true;
        acc[r] = acc[r] / d;
        x[(long long)row[r] * nrhs + c] = acc[r];
      }
    }
  }
  // a zero pivot poisons the whole system (all columns share T): NaN-fill like the reference
  if (fail) {
    if (failflag) atomicOr(failflag, 1);
    for (int i = 0; i < n; i++) x[(long long)i * nrhs + c] = (T)__builtin_nan("");
  }
}

// generic (any n, nrhs): one thread per right-hand-side column, row-oriented substitution.
template <class T>
__global__ __launch_bounds__(BLOCK) void trsm_kernel(T* __restrict__ Xout,
                                                    const T* __restrict__ Tm, long long sTb,
                                                    long long sT0, long long sT1,
                                                    const T* __restrict__ B, long long sBb, int n,
                                                    int nrhs, int lower, int unit) {
  const long long mat = blockIdx.y;
  const int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= nrhs) return;
  const T* Tg = Tm + mat * sTb;
  const T* b = B + mat * sBb;
  T* x = Xout + mat * (long long)n * nrhs;
  bool fail = false;
  for (int s = 0; s < n; s++) {
    const int i = lower ? s : n - 1 - s;
    T acc = b[(long long)i * nrhs + c];
    if (lower) {
      for (int j = 0; j < i; j++) acc -= Tg[i * sT0 + j * sT1] * x[(long long)j * nrhs + c];
    } else {
      for (int j = n - 1; j > i; j--) acc -= Tg[i * sT0 + j * sT1] * x[(long long)j * nrhs + c];
    }
    const T d = unit ? T(1) : Tg[i * sT0 + i * sT1];
    if (d == T(0)) fail = true;
    x[(long long)i * nrhs + c] = acc / d;
  }
  // a zero pivot poisons the whole system (all columns share T): NaN-fill like the reference
  if (fail)
    for (int i = 0; i < n; i++) x[(long long)i * nrhs + c] = (T)__builtin_nan("");
}

template <class T>
int potrf_typed(int lower, long long batch, long long n, const void* A, void* L,
                const void* rhs = nullptr, void* xout = nullptr) {
  hipStream_t st = pthip::ctx().stream;
  if (batch == 0 || n == 0) return 0;
  if (rhs != nullptr && (!lower || n > 256))
    return pthip::set_error("pthip_potrf_trsv: fused solve needs the lower factor and n <= 256");
  const size_t ld = (size_t)(n | 1);
  const size_t need = (size_t)n * ld * sizeof(T);
  if (need <= 160 * 1024 - 2560) {  // (the kernel's static LDS: a 64-entry column per wave + flags)
    auto k = potrf_lds_kernel<T>;
    if (need > 64 * 1024)
      PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    PTHIP_KLAUNCH(k, dim3((unsigned)batch), dim3(BLOCK), need, st, (T*)L, (const T*)A, (int)n, lower,
                       (const T*)rhs, (T*)xout, 0LL, (int*)nullptr);
    return pthip::post_launch("potrf_lds");
  }
  if (rhs != nullptr) return pthip::set_error("pthip_potrf_trsv: matrix does not fit the LDS-resident kernel");
  // beyond one CU's LDS: the blocked multi-workgroup factorisation, one matrix after the other
  for (long long b = 0; b < batch; b++) {
    int r = chol_blocked<T>(lower, n, (const T*)A + b * n * n, (T*)L + b * n * n);
    if (r) return r;
  }
  return 0;
}

// ---------------------------------------------------------------------------------
// Triangular solves beyond the LDS (n > 141 fp64 / 200 fp32).
//
// Few right-hand sides (the GP / PyMC case: L^-1 y, L^-T z with vectors): memory-bound, n^2/2 entries
// of T read once, but n/64 dependent steps.  One persistent kernel, workgroup s owns rows 64s .. 64s+63
// (in "solve coordinates": an upper or transposed system is walked backwards through signed strides, so
// the kernel only ever sees a lower triangle): it inverts its 64 x 64 diagonal block in LDS up front
// (off the chain), then takes x_0 .. x_{s-1} as they appear, acc += T[s,k] x_k with the blocks of T
// prefetched (they are inputs), and finishes with x_s = Dinv (b_s - acc).  Hand-over without a fence: every
// solution entry is published as ONE 16-byte write-through store {bits(x), bits(x) ^ MAGIC} into a
// zeroed scratch array and consumers poll the pair itself — a torn or stale read fails the check unless
// it already equals the final pair, so only 8-byte atomicity is assumed — which takes the release fence
// (1.5-2 us per hop, the write-through latency) and the separate flag load off every hop.
//
// Many right-hand sides (trsm_blocked): the inverses of all 256 x 256 diagonal blocks in one launch
// (tri_inv256_kernel), then per block X_b = inv(T_bb) B_b and the update of the rows still to come, both on
// the MFMA GEMM with K = 256.
// A zero pivot anywhere NaN-fills the whole result, as the LDS-resident kernels do.
// ---------------------------------------------------------------------------------
constexpr int TV = 64;        // rows per workgroup
constexpr int TVS = TV + 1;   // LDS row stride
constexpr int TV_NR = 4;      // right-hand sides per launch
constexpr unsigned long long TV_MAGIC = 0x7ff4c0de5ea1ed01ull;

__device__ __forceinline__ unsigned long long tv_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned long long tv_bits(float v) { return (unsigned long long)__float_as_uint(v); }
__device__ __forceinline__ void tv_from_bits(unsigned long long b, double& v) { v = __longlong_as_double((long long)b); }
__device__ __forceinline__ void tv_from_bits(unsigned long long b, float& v) { v = __uint_as_float((unsigned)b); }

// Di <- inverse of the 64 x 64 lower-triangular Ds (both [TV][TVS] in LDS), all BLOCK threads; ends on a
// barrier.  The four diagonal 16 x 16 blocks by forward substitution (one wave, lane = 16 b + column), the
// blocks below them level by level on the matrix cores: Inv_ij = -Inv_ii (sum_{k=j..i-1} L_ik Inv_kj).
// (A thread-per-column substitution over the whole block is a 2016-step dependent chain of LDS reads:
// 95 us, on the critical path of the first row block.)
template <class T>
__device__ __forceinline__ void tri_inverse64(const T* __restrict__ Ds, T* __restrict__ Di, T* __restrict__ scratch) {
  typedef typename Mfma16<T>::v4 v4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  for (int e = tid; e < TV * TV; e += BLOCK) Di[(e >> 6) * TVS + (e & 63)] = T(0);
  __syncthreads();
  if (wave == 0) {
    const T* Lb = Ds + (lq * 16) * TVS + lq * 16;
    T x[16];
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = (r == li) ? T(1) : T(0);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      x[q] = x[q] / Lb[q * TVS + q];
#pragma unroll
      for (int r = q + 1; r < 16; r++) x[r] -= Lb[r * TVS + q] * x[q];
    }
#pragma unroll
    for (int r = 0; r < 16; r++) Di[(lq * 16 + r) * TVS + lq * 16 + li] = x[r];
  }
  __syncthreads();
  T* Mt = scratch + wave * (16 * 17);
  for (int d = 1; d < 4; d++) {
    const int j = wave, i = wave + d;  // block (i, j) of this level on wave j
    if (i < 4) {
      v4 m = {T(0), T(0), T(0), T(0)};
      for (int k = j; k < i; k++) {
#pragma unroll
        for (int st = 0; st < 4; st++)
          m = Mfma16<T>::run(Ds[(16 * i + li) * TVS + 16 * k + 4 * st + lq], Di[(16 * k + 4 * st + lq) * TVS + 16 * j + li], m);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) Mt[Mfma16<T>::drow(lane, r) * 17 + li] = m[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      v4 o = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int st = 0; st < 4; st++)
        o = Mfma16<T>::run(Di[(16 * i + li) * TVS + 16 * i + 4 * st + lq], Mt[(4 * st + lq) * 17 + li], o);
#pragma unroll
      for (int r = 0; r < 4; r++) Di[(16 * i + Mfma16<T>::drow(lane, r)) * TVS + 16 * j + li] = -o[r];
    }
    __syncthreads();
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void trsv_dag_kernel(T* __restrict__ Xp, long long x0, const T* __restrict__ Tp,
                                                        long long t0, long long t1, const T* __restrict__ Bp,
                                                        long long b0, int n, int nr, int unit,
                                                        unsigned long long* __restrict__ box, int* __restrict__ failflag,
                                                        int* __restrict__ abortflag, int* __restrict__ ticket, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int s_ok;
  T* Ds = (T*)smem_raw;        // [TV][TVS] diagonal block (lower, identity beyond n)
  T* Di = Ds + TV * TVS;       // [TV][TVS] its inverse
  T* Ts = Ds;                  // [TV][TVS] off-diagonal block T[s, k] (the diagonal block is dead once inverted)
  T* xs = Di + TV * TVS;       // [TV][TV_NR] x_k, then the right-hand side of the diagonal solve
  T* red = xs + TV * TV_NR;    // [4][TV][TV_NR] partial sums of the four column quarters
  const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
  // the row block is CLAIMED through a ticket, not read off blockIdx.x: block s waits only for blocks below s, and
  // those were claimed by workgroups that are running — no assumption about which workgroups are resident together
  // or in which order the dispatcher starts them (other processes' kernels may hold CUs)
  if (tid == 0) s_ok = atomicAdd(ticket, 1);
  __syncthreads();
  const int s = s_ok;
  __syncthreads();
  const long long rbase = (long long)s * TV;
  const int nb = (n - rbase) < TV ? (int)(n - rbase) : TV;
  const bool rowfast = (t0 == 1 || t0 == -1) && !(t1 == 1 || t1 == -1);  // which index walks memory
  constexpr int NL = TV * TV / BLOCK;  // 16 entries of a block per thread
  auto coord = [&](int e, int& i, int& j) {  // e -> (row, column), fast-in-memory index on adjacent threads
    const int a = e / TV, c = e - a * TV;
    i = rowfast ? c : a;
    j = rowfast ? a : c;
  };
  // ---- the diagonal block and its inverse (before any waiting)
  bool fail = false;
  {
    T v[NL];
#pragma unroll
    for (int u = 0; u < NL; u++) {
      int i, j;
      coord(u * BLOCK + tid, i, j);
      v[u] = (i < nb && j < i) ? Tp[(rbase + i) * t0 + (rbase + j) * t1] : T(0);
      if (i == j) {
        v[u] = (i < nb && !unit) ? Tp[(rbase + i) * t0 + (rbase + i) * t1] : T(1);
        if (v[u] == T(0)) fail = true;  // trtrs: exact singularity
      }
    }
#pragma unroll
    for (int u = 0; u < NL; u++) {
      int i, j;
      coord(u * BLOCK + tid, i, j);
      Ds[i * TVS + j] = v[u];
    }
  }
  if (fail) atomicOr(failflag, 1);
  __syncthreads();
  tri_inverse64<T>(Ds, Di, xs);  // (scratch: xs and red, contiguous, 4 x 16 x 17 entries)
  // ---- acc = sum_k T[s,k] x_k
  T acc[TV_NR];
#pragma unroll
  for (int j = 0; j < TV_NR; j++) acc[j] = T(0);
  T tv[NL];
  auto fetch = [&](int k) {
#pragma unroll
    for (int u = 0; u < NL; u++) {
      int i, j;
      coord(u * BLOCK + tid, i, j);
      tv[u] = (i < nb) ? Tp[(rbase + i) * t0 + ((long long)k * TV + j) * t1] : T(0);
    }
  };
  if (s > 0) fetch(0);
  for (int k = 0; k < s; k++) {
#pragma unroll
    for (int u = 0; u < NL; u++) {
      int i, j;
      coord(u * BLOCK + tid, i, j);
      Ts[i * TVS + j] = tv[u];
    }
    if (k + 1 < s) fetch(k + 1);  // in flight under the wait
    if (tid < TV) {
      const unsigned long long* bx = box + ((long long)k * TV + lane) * (TV_NR * 2);
      unsigned long long a[TV_NR];
      int spins = 0, ok;
      for (;;) {
        bool mine = true;
#pragma unroll
        for (int j = 0; j < TV_NR; j++) {
          if (j < nr) {
            a[j] = __hip_atomic_load(bx + 2 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long b = __hip_atomic_load(bx + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mine = mine && ((a[j] ^ b) == TV_MAGIC);
          }
        }
        ok = __builtin_amdgcn_ballot_w64(!mine) == 0ull;
        if (ok) break;
        if (++spins > DAG_SPIN_LIMIT || ((spins & 255) == 0 && dag_flag(abortflag))) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) s_ok = ok;
#pragma unroll
      for (int j = 0; j < TV_NR; j++) {
        T xv = T(0);
        if (j < nr) tv_from_bits(a[j], xv);
        xs[lane * TV_NR + j] = xv;
      }
    }
    __syncthreads();
    if (!s_ok) {
      if (tid == 0) { atomicOr(status, 16); __hip_atomic_store(abortflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      return;
    }
#pragma unroll
    for (int cc = 0; cc < TV / 4; cc++) {
      const int c = part * (TV / 4) + cc;
      const T t = Ts[lane * TVS + c];
#pragma unroll
      for (int j = 0; j < TV_NR; j++) acc[j] += t * xs[c * TV_NR + j];
    }
    __syncthreads();
  }
  // ---- x_s = Dinv (b_s - acc)
#pragma unroll
  for (int j = 0; j < TV_NR; j++) red[(part * TV + lane) * TV_NR + j] = acc[j];
  __syncthreads();
  {
    const int r = lane, j = part;  // (TV_NR == 4 == number of waves)
    T v = T(0);
    if (r < nb && j < nr) v = Bp[(rbase + r) * b0 + j];
#pragma unroll
    for (int q = 0; q < 4; q++) v -= red[(q * TV + r) * TV_NR + j];
    __syncthreads();
    xs[r * TV_NR + j] = (j < nr) ? v : T(0);
  }
  __syncthreads();
  {
    T px[TV_NR];
#pragma unroll
    for (int j = 0; j < TV_NR; j++) px[j] = T(0);
#pragma unroll
    for (int cc = 0; cc < TV / 4; cc++) {
      const int c = part * (TV / 4) + cc;
      const T t = Di[lane * TVS + c];
#pragma unroll
      for (int j = 0; j < TV_NR; j++) px[j] += t * xs[c * TV_NR + j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TV_NR; j++) red[(part * TV + lane) * TV_NR + j] = px[j];
  }
  __syncthreads();
  {
    const int r = lane, j = part;
    if (j < nr) {
      T v = T(0);
#pragma unroll
      for (int q = 0; q < 4; q++) v += red[(q * TV + r) * TV_NR + j];
      if (r < nb) Xp[(rbase + r) * x0 + j] = v;
      // the pair in one 16-byte write-through store
      typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
      const unsigned long long bits = tv_bits(v);
      u2 pr = {bits, bits ^ TV_MAGIC};
      u2* dst = (u2*)(box + ((long long)s * TV + r) * (TV_NR * 2) + 2 * j);
      // (s_nop: the store reads its upper data dwords after issue — the hazard the compiler pads for its own stores)
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"(dst), "v"(pr) : "memory");
    }
  }
}


// one matrix, nrhs <= 16: chunks of TV_NR right-hand sides, one persistent launch each
template <class T>
int trsv_dag(int lower, int unit, long long n, long long nrhs, const T* Tm, long long sT0, long long sT1, const T* B, T* out) {
  hipStream_t st = pthip::ctx().stream;
  const int nB = (int)((n + TV - 1) / TV);
  const int nchunk = (int)((nrhs + TV_NR - 1) / TV_NR);
  const size_t boxbytes = (size_t)nB * TV * TV_NR * 2 * sizeof(unsigned long long);
  void* scratch = nullptr;
  const size_t flagbytes = 256 + ((size_t)nchunk * sizeof(int) + 255) / 256 * 256;  // fail, abort; from word 64 on: one ticket per chunk
  int r = pthip_alloc(boxbytes * nchunk + flagbytes, &scratch);
  if (r) return r;
  auto fail = [&](int rc) { pthip_free(scratch); return rc; };
  int* flags = (int*)((char*)scratch + boxbytes * nchunk);
  auto kk = trsv_dag_kernel<T>;
  const size_t lds = (size_t)(2 * TV * TVS + TV * TV_NR + 4 * TV * TV_NR) * sizeof(T);
  static int resident = 0;
  if (!resident) {
    if (lds > 64 * 1024)
      if (hipError_t e = hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess)
        return fail(pthip::check(e, "trsv_dag attribute"));
    int per_cu = 0, dev = 0, cus = 0;
    if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kk, BLOCK, lds); e != hipSuccess)
      return fail(pthip::check(e, "trsv_dag occupancy"));
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return fail(pthip::check(e, "trsv_dag device"));
    if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess)
      return fail(pthip::check(e, "trsv_dag device attribute"));
    resident = per_cu * cus;
  }
  if (nB > resident) {
    // more row blocks than the device holds workgroups (a smaller part, or n beyond 64 x CUs x occupancy): the
    // persistent kernel would wait on blocks that are not running.  The blocked solve has no such requirement.
    pthip_free(scratch);
    return 1 << 30;  // (the caller falls back to trsm_blocked)
  }
  if (hipError_t e = pthip::memset_async(scratch, 0, boxbytes * nchunk + flagbytes, st); e != hipSuccess) return fail(pthip::check(e, "trsv box memset"));
  // solve coordinates: p = lower ? i : n-1-i
  const long long sg = lower ? 1 : -1;
  const T* Tp = lower ? Tm : Tm + (n - 1) * (sT0 + sT1);
  const T* Bp = lower ? B : B + (n - 1) * nrhs;
  T* Xp = lower ? out : out + (n - 1) * nrhs;
  for (int c = 0; c < nchunk; c++) {
    const int nr = (int)((nrhs - (long long)c * TV_NR) < TV_NR ? (nrhs - (long long)c * TV_NR) : TV_NR);
    PTHIP_KLAUNCH(kk, dim3((unsigned)nB), dim3(BLOCK), lds, st, Xp + (long long)c * TV_NR, sg * nrhs, Tp, sg * sT0, sg * sT1,
                  Bp + (long long)c * TV_NR, sg * nrhs, (int)n, nr, unit,
                  (unsigned long long*)((char*)scratch + boxbytes * c), flags, flags + 1, flags + 64 + c, pthip::ctx().status_dev);
    if ((r = pthip::post_launch("trsv_dag"))) return fail(r);
  }
  PTHIP_KLAUNCH((nan_fill_if_kernel<T>), dim3(256), dim3(BLOCK), 0, st, out, n * nrhs, (const int*)flags);
  r = pthip::post_launch("trsm nan fill");
  pthip_free(scratch);
  return r;
}

// Inverses of all 256 x 256 diagonal blocks of T in one launch (block b -> Tinv[b], dense 256 x 256
// row-major, identity beyond n), one workgroup per block.  Inside, 4 x 4 tiles of 64: the diagonal tiles by
// tri_inverse64, the tiles below them level by level, Inv(i,j) = -Inv(i,i) (sum_{k=j..i-1} T(i,k) Inv(k,j)),
// every 64^3 product on the matrix cores from LDS-staged operands.  An upper triangle is inverted through
// its transpose M = U^T (lower): inv(U) = inv(M)^T — all reads of T and all accesses to the result go
// through the same (i, j) -> address maps.
constexpr int TB = 256;  // rows per diagonal block of the many-right-hand-sides solve

template <class T>
__global__ __launch_bounds__(BLOCK) void tri_inv256_kernel(T* __restrict__ Tinv, const T* __restrict__ Tm, long long sT0,
                                                          long long sT1, int n, int lower, int unit,
                                                          int* __restrict__ failflag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* As = (T*)smem_raw;          // [TV][TVS] left operand / the diagonal tile
  T* Bs = As + TV * TVS;         // [TV][TVS] right operand / its inverse
  T* scratch = Bs + TV * TVS;    // [4][16][17] (tri_inverse64)
  typedef typename Mfma16<T>::v4 v4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
  const long long base = (long long)blockIdx.x * TB;
  T* out = Tinv + (long long)blockIdx.x * TB * TB;
  // entry (i, j) of the lower-triangular matrix being inverted, i, j relative to this block
  auto Mv = [&](int i, int j) -> T {
    const long long gi = base + i, gj = base + j;
    if (gi >= n || gj >= n) return (i == j) ? T(1) : T(0);
    if (i == j && unit) return T(1);
    return lower ? Tm[gi * sT0 + gj * sT1] : Tm[gj * sT0 + gi * sT1];
  };
  auto Oaddr = [&](int i, int j) -> T* { return lower ? out + (long long)i * TB + j : out + (long long)j * TB + i; };
  bool fail = false;
  // ---- diagonal tiles
  for (int d = 0; d < TB / TV; d++) {
    {
      T va[TV * TV / BLOCK];
#pragma unroll
      for (int u = 0; u < TV * TV / BLOCK; u++) {
        const int e = u * BLOCK + tid, i = e >> 6, j = e & 63;
        va[u] = (j <= i) ? Mv(TV * d + i, TV * d + j) : T(0);
        if (i == j && va[u] == T(0)) fail = true;  // trtrs: exact singularity
      }
#pragma unroll
      for (int u = 0; u < TV * TV / BLOCK; u++) {
        const int e = u * BLOCK + tid;
        As[(e >> 6) * TVS + (e & 63)] = va[u];
      }
    }
    __syncthreads();
    tri_inverse64<T>(As, Bs, scratch);  // (ends on a barrier)
    for (int e = tid; e < TV * TV; e += BLOCK) {
      const int i = e >> 6, j = e & 63;
      *Oaddr(TV * d + i, TV * d + j) = Bs[i * TVS + j];
    }
    __syncthreads();
  }
  if (fail) atomicOr(failflag, 1);
  __threadfence();
  __syncthreads();
  // ---- tiles below the diagonal, level by level (tile (i, j), i - j = d)
  auto product = [&](v4 (&acc)[2][2]) {  // acc += As * Bs (64 x 64 x 64), this wave's 32 x 32 quadrant
#pragma unroll 4
    for (int st = 0; st < TV / 4; st++) {
      const T a0 = As[(r0 + li) * TVS + 4 * st + lq], a1 = As[(r0 + 16 + li) * TVS + 4 * st + lq];
      const T b0 = Bs[(4 * st + lq) * TVS + c0 + li], b1 = Bs[(4 * st + lq) * TVS + c0 + 16 + li];
      acc[0][0] = Mfma16<T>::run(a0, b0, acc[0][0]);
      acc[0][1] = Mfma16<T>::run(a0, b1, acc[0][1]);
      acc[1][0] = Mfma16<T>::run(a1, b0, acc[1][0]);
      acc[1][1] = Mfma16<T>::run(a1, b1, acc[1][1]);
    }
  };
  for (int d = 1; d < TB / TV; d++) {
    for (int i = d; i < TB / TV; i++) {
      const int j = i - d;
      v4 acc[2][2];
#pragma unroll
      for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++) acc[x][y] = v4{T(0), T(0), T(0), T(0)};
      for (int k = j; k < i; k++) {
        {  // (all 32 loads of a thread in flight before the first LDS store: a load-store loop is 16 round trips)
          T va[TV * TV / BLOCK], vb[TV * TV / BLOCK];
#pragma unroll
          for (int u = 0; u < TV * TV / BLOCK; u++) {
            const int e = u * BLOCK + tid, r = e >> 6, c = e & 63;
            va[u] = Mv(TV * i + r, TV * k + c);
            vb[u] = *Oaddr(TV * k + r, TV * j + c);
          }
#pragma unroll
          for (int u = 0; u < TV * TV / BLOCK; u++) {
            const int e = u * BLOCK + tid, r = e >> 6, c = e & 63;
            As[r * TVS + c] = va[u];
            Bs[r * TVS + c] = vb[u];
          }
        }
        __syncthreads();
        product(acc);
        __syncthreads();
      }
      // S -> Bs, Inv(i,i) -> As, Inv(i,j) = -(As * Bs)
#pragma unroll
      for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++)
#pragma unroll
          for (int r = 0; r < 4; r++) Bs[(r0 + x * 16 + Mfma16<T>::drow(lane, r)) * TVS + c0 + y * 16 + li] = acc[x][y][r];
      {
        T va[TV * TV / BLOCK];
#pragma unroll
        for (int u = 0; u < TV * TV / BLOCK; u++) {
          const int e = u * BLOCK + tid;
          va[u] = *Oaddr(TV * i + (e >> 6), TV * i + (e & 63));
        }
#pragma unroll
        for (int u = 0; u < TV * TV / BLOCK; u++) {
          const int e = u * BLOCK + tid;
          As[(e >> 6) * TVS + (e & 63)] = va[u];
        }
      }
      __syncthreads();
      v4 res[2][2];
#pragma unroll
      for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++) res[x][y] = v4{T(0), T(0), T(0), T(0)};
      product(res);
#pragma unroll
      for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            *Oaddr(TV * i + r0 + x * 16 + Mfma16<T>::drow(lane, r), TV * j + c0 + y * 16 + li) = -res[x][y][r];
      __syncthreads();
    }
    __threadfence();  // this level's tiles are operands of the next
    __syncthreads();
  }
  // the other triangle of the dense block: zeros
  for (int e = tid; e < TB * TB; e += BLOCK) {
    const int i = e / TB, j = e - i * TB;
    if (j > i) *Oaddr(i, j) = T(0);
  }
}

// The 64 x 64 diagonal tiles alone, one workgroup each (n / 64 of them side by side; tri_inv256_kernel walks its four
// tiles and six off-diagonal tiles one after the other in a single workgroup: 205 us at any n, profiles/r3y).  The
// doubling below takes them to 128, 256, 512 with batched GEMMs.
template <class T>
__global__ __launch_bounds__(BLOCK) void tri_inv64_kernel(T* __restrict__ Tinv, const T* __restrict__ Tm, long long sT0, long long sT1, int n,
                                                         int lower, int unit, int* __restrict__ failflag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* As = (T*)smem_raw;
  T* Bs = As + TV * TVS;
  T* scratch = Bs + TV * TVS;
  const int tid = threadIdx.x;
  const long long base = (long long)blockIdx.x * TV;
  T* out = Tinv + (long long)blockIdx.x * TV * TV;
  bool fail = false;
  {
    T va[TV * TV / BLOCK];
#pragma unroll
    for (int u = 0; u < TV * TV / BLOCK; u++) {
      const int e = u * BLOCK + tid, i = e >> 6, j = e & 63;
      const long long gi = base + i, gj = base + j;
      T v = T(0);
      if (j <= i) {
        if (gi >= n) v = (i == j) ? T(1) : T(0);
        else if (i == j && unit) v = T(1);
        else v = lower ? Tm[gi * sT0 + gj * sT1] : Tm[gj * sT0 + gi * sT1];
      }
      va[u] = v;
      if (i == j && v == T(0)) fail = true;  // trtrs: exact singularity
    }
#pragma unroll
    for (int u = 0; u < TV * TV / BLOCK; u++) {
      const int e = u * BLOCK + tid;
      As[(e >> 6) * TVS + (e & 63)] = va[u];
    }
  }
  if (fail) atomicOr(failflag, 1);
  __syncthreads();
  tri_inverse64<T>(As, Bs, scratch);  // (ends on a barrier)
  for (int e = tid; e < TV * TV; e += BLOCK) {
    const int a = e >> 6, c = e & 63;  // memory order of `out`
    out[e] = lower ? Bs[a * TVS + c] : Bs[c * TVS + a];
  }
}

// Inverses of diagonal blocks of size 2s from those of size s:  inv([[A, 0], [C, D]]) = [[A^-1, 0], [-D^-1 C A^-1, D^-1]]
// (lower; the upper case mirrors it).  `off` holds the off-diagonal blocks (s x s each, computed by two batched GEMMs),
// a pair without a second block gets the identity there (rows beyond n carry the identity from the level below).
template <class T>
__global__ __launch_bounds__(BLOCK) void tri_inv_assemble_kernel(T* __restrict__ Inv2, const T* __restrict__ Inv, const T* __restrict__ off,
                                                                int s, int nb_s, int lower) {
  const long long p = blockIdx.y;
  const int b0 = 2 * (int)p, b1 = b0 + 1;
  const long long s2 = 2LL * s;
  T* dst = Inv2 + p * s2 * s2;
  const T* A = Inv + (long long)b0 * s * s;
  const T* D = b1 < nb_s ? Inv + (long long)b1 * s * s : nullptr;
  const T* O = off + p * (long long)s * s;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < s2 * s2; e += (long long)gridDim.x * BLOCK) {
    const int i = (int)(e / s2), j = (int)(e - (long long)i * s2);
    T v;
    if (i < s && j < s) v = A[(long long)i * s + j];
    else if (i >= s && j >= s) v = D ? D[(long long)(i - s) * s + (j - s)] : (i == j ? T(1) : T(0));
    else if (lower ? (i >= s) : (i < s)) v = D ? O[(long long)(lower ? i - s : i) * s + (lower ? j : j - s)] : T(0);
    else v = T(0);
    dst[e] = v;
  }
}

// one matrix, many right-hand sides: the inverses of the 256 x 256 diagonal blocks up front (one launch),
// then per block X_b = inv(T_bb) B_b and the update of all rows still to come, both on the MFMA GEMM at
// K = 256 — three launches per 256 rows (a version with 64-row inner steps and K = 64 updates inside the
// outer block: 110 launches at n = 2048, 1.6 ms; this one: 25 launches)
template <class T>
int trsm_blocked(int lower, int unit, long long n, long long nrhs, const T* Tm, long long sT0, long long sT1, const T* B, T* out) {
  hipStream_t st = pthip::ctx().stream;
  const int dt = sizeof(T) == 8 ? PTHIP_F64 : PTHIP_F32;
  // (clamped to 4*TB = 1024 rows: the step scratch Xt below is sized for steps of at most that many rows — ADVICE r4)
  static const int blk_env = std::min(getenv("PTHIP_TRSM_BLOCK") ? atoi(getenv("PTHIP_TRSM_BLOCK")) : 512, 4 * (int)TB);
  // start from 64 x 64 tiles when the doubling below will carry them to 256 rows or more (PTHIP_TRSM_BASE=256: the
  // round-3 single-workgroup inverse of each 256-row block)
  static const int base_env = getenv("PTHIP_TRSM_BASE") ? atoi(getenv("PTHIP_TRSM_BASE")) : 64;
  const bool base64 = base_env == 64 && blk_env >= 256 && nrhs >= 512 && n > TV;
  const long long IB = base64 ? TV : TB;  // rows of the blocks inverted inside a workgroup
  const long long nB = (n + IB - 1) / IB;
  const size_t invbytes = (size_t)nB * IB * IB * sizeof(T), tmpbytes = (size_t)4 * TB * nrhs * sizeof(T);  // (steps of up to 1024 rows)
  void* scratch = nullptr;
  int r = pthip_alloc(invbytes + tmpbytes + 256, &scratch);
  if (r) return r;
  auto fail = [&](int rc) { pthip_free(scratch); return rc; };
  T* Tinv = (T*)scratch;
  T* Xt = (T*)((char*)scratch + invbytes);
  int* flag = (int*)((char*)scratch + invbytes + tmpbytes);
  if (hipError_t e = pthip::memset_async(Xt, 0, tmpbytes + 256, st); e != hipSuccess) return fail(pthip::check(e, "trsm scratch memset"));
  if (hipError_t e = pthip::memcpy_async(out, B, (size_t)n * nrhs * sizeof(T), hipMemcpyDeviceToDevice, st); e != hipSuccess)
    return fail(pthip::check(e, "trsm rhs copy"));
  auto ki = tri_inv256_kernel<T>;
  const size_t lds = (size_t)(2 * TV * TVS + 4 * 16 * 17) * sizeof(T);
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipError_t e = hipFuncSetAttribute((const void*)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess)
      return fail(pthip::check(e, "tri_inv256 attribute"));
    if (hipError_t e = hipFuncSetAttribute((const void*)tri_inv64_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess)
      return fail(pthip::check(e, "tri_inv64 attribute"));
    attr = true;
  }
  if (base64) PTHIP_KLAUNCH((tri_inv64_kernel<T>), dim3((unsigned)nB), dim3(BLOCK), lds, st, Tinv, Tm, sT0, sT1, (int)n, lower, unit, flag);
  else PTHIP_KLAUNCH(ki, dim3((unsigned)nB), dim3(BLOCK), lds, st, Tinv, Tm, sT0, sT1, (int)n, lower, unit, flag);
  if ((r = pthip::post_launch(base64 ? "tri_inv64" : "tri_inv256"))) return fail(r);
  // Round 4: with many right-hand sides the 256-row steps are 25 dependent launches at n = 2048 whose solve GEMMs
  // (M = 256) fill an eighth of the chip.  The block inverses are doubled once (256 -> 512: per pair
  // inv = [[A^-1, 0], [-D^-1 C A^-1, D^-1]], two batched GEMMs and an assemble launch for all pairs), then the same
  // right-looking sweep runs with 512-row steps.  (Each doubling squares nothing but widens the block whose explicit
  // inverse is applied: PTHIP_TRSM_BLOCK=256 keeps the round-3 form, 1024 doubles twice.)
  long long SB = IB;  // rows per step
  T* InvCur = Tinv;
  void* lvl_scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  auto fail2 = [&](int rc) { for (void* q : lvl_scratch) if (q) pthip_free(q); return fail(rc); };
  for (int lvl = 0; lvl < 4 && SB * 2 <= blk_env && nrhs >= 2 * SB && n > SB; lvl++) {
    const long long sblk = SB, nb_s = (n + sblk - 1) / sblk, npairs = (nb_s + 1) / 2;
    const size_t inv2bytes = (size_t)npairs * 4 * sblk * sblk * sizeof(T), offbytes = (size_t)npairs * sblk * sblk * sizeof(T);
    void* q = nullptr;
    if ((r = pthip_alloc(inv2bytes + 2 * offbytes + 256, &q))) return fail2(r);
    lvl_scratch[lvl] = q;
    T* Inv2 = (T*)q;
    T* tmp = (T*)((char*)q + inv2bytes);
    T* off = (T*)((char*)q + inv2bytes + offbytes);
    if (hipError_t e = pthip::memset_async(tmp, 0, 2 * offbytes, st); e != hipSuccess) return fail2(pthip::check(e, "trsm level memset"));
    long long full = 0;  // pairs whose second block is complete
    while (full < npairs && (2 * full + 2) * sblk <= n) full++;
    const long long with_b1 = nb_s / 2;  // pairs that have a second block at all (complete or partial)
    const long long pstride = 2 * sblk * (sT0 + sT1);
    // C of pair p: lower -> rows of the second block x columns of the first; upper -> the mirror
    const T* C0 = lower ? Tm + sblk * sT0 : Tm + sblk * sT1;
    if (lower) {
      // tmp = C A^-1 (valid rows only), off = -D^-1 tmp
      if (full > 0)
        if ((r = pthip_gemm(dt, full, sblk, sblk, sblk, 1.0, C0, pstride, sT0, sT1, InvCur, 2 * sblk * sblk, sblk, 1, 0.0, nullptr, 0, 0, 0, tmp))) return fail2(r);
      if (with_b1 > full) {
        const long long pidx = full, rows = n - (2 * pidx + 1) * sblk;
        if ((r = pthip::gemm_inplace(dt, rows, sblk, sblk, 1.0, C0 + pidx * pstride, sT0, sT1, InvCur + 2 * pidx * sblk * sblk, sblk, 1, 0.0,
                                     tmp + pidx * sblk * sblk, sblk))) return fail2(r);
      }
      if (with_b1 > 0)
        if ((r = pthip_gemm(dt, with_b1, sblk, sblk, sblk, -1.0, InvCur + sblk * sblk, 2 * sblk * sblk, sblk, 1, tmp, sblk * sblk, sblk, 1, 0.0, nullptr, 0, 0,
                            0, off))) return fail2(r);
    } else {
      // tmp = A^-1 C (valid columns only), off = -tmp D^-1
      if (full > 0)
        if ((r = pthip_gemm(dt, full, sblk, sblk, sblk, 1.0, InvCur, 2 * sblk * sblk, sblk, 1, C0, pstride, sT0, sT1, 0.0, nullptr, 0, 0, 0, tmp))) return fail2(r);
      if (with_b1 > full) {
        const long long pidx = full, cols = n - (2 * pidx + 1) * sblk;
        if ((r = pthip::gemm_inplace(dt, sblk, cols, sblk, 1.0, InvCur + 2 * pidx * sblk * sblk, sblk, 1, C0 + pidx * pstride, sT0, sT1, 0.0,
                                     tmp + pidx * sblk * sblk, sblk))) return fail2(r);
      }
      if (with_b1 > 0)
        if ((r = pthip_gemm(dt, with_b1, sblk, sblk, sblk, -1.0, tmp, sblk * sblk, sblk, 1, InvCur + sblk * sblk, 2 * sblk * sblk, sblk, 1, 0.0, nullptr, 0, 0,
                            0, off))) return fail2(r);
    }
    PTHIP_KLAUNCH((tri_inv_assemble_kernel<T>), dim3(64, (unsigned)npairs), dim3(BLOCK), 0, st, Inv2, (const T*)InvCur, (const T*)off, (int)sblk, (int)nb_s, lower);
    if ((r = pthip::post_launch("tri_inv_assemble"))) return fail2(r);
    InvCur = Inv2;
    SB = 2 * sblk;
  }
  // block [k0, k0 + nb): solve it, then update rows [u0, u1)
  auto step = [&](long long k0, long long nb, long long u0, long long u1) -> int {
    T* Xk = out + k0 * nrhs;
    int rc = pthip::gemm_inplace(dt, nb, nrhs, nb, 1.0, InvCur + (k0 / SB) * SB * SB, SB, 1, Xk, nrhs, 1, 0.0, Xt, nrhs);
    if (rc) return rc;
    if (hipError_t e = pthip::memcpy_async(Xk, Xt, (size_t)nb * nrhs * sizeof(T), hipMemcpyDeviceToDevice, st); e != hipSuccess)
      return pthip::check(e, "trsm block copy");
    if (u1 > u0) return pthip::gemm_inplace(dt, u1 - u0, nrhs, nb, -1.0, Tm + u0 * sT0 + k0 * sT1, sT0, sT1, Xt, nrhs, 1, 1.0, out + u0 * nrhs, nrhs);
    return 0;
  };
  if (lower) {
    for (long long k0 = 0; k0 < n; k0 += SB) {
      const long long nb = (n - k0) < SB ? (n - k0) : SB;
      if ((r = step(k0, nb, k0 + nb, n))) return fail2(r);
    }
  } else {
    for (long long k0 = (n - 1) / SB * SB; k0 >= 0; k0 -= SB) {
      const long long nb = (n - k0) < SB ? (n - k0) : SB;
      if ((r = step(k0, nb, 0, k0))) return fail2(r);
    }
  }
  PTHIP_KLAUNCH((nan_fill_if_kernel<T>), dim3(256), dim3(BLOCK), 0, st, out, n * nrhs, (const int*)flag);
  r = pthip::post_launch("trsm nan fill");
  for (void* q : lvl_scratch) if (q) pthip_free(q);
  pthip_free(scratch);
  return r;
}

template <class T>
int trsm_typed(int lower, int unit, long long batch, long long n, long long nrhs, const void* Tm,
               long long sTb, long long sT0, long long sT1, const void* B, long long sBb,
               void* out) {
  hipStream_t st = pthip::ctx().stream;
  if (batch == 0 || n == 0 || nrhs == 0) return 0;
  if (nrhs == 1 && n <= 256) {
    const size_t ld = (size_t)(n | 1);
    const size_t need = (size_t)n * ld * sizeof(T);
    if (need <= 160 * 1024 - 256) {
#define LAUNCH_TRSV(RPL)                                                                         \
  do {                                                                                           \
    auto k = trsv_lds_kernel<T, RPL>;                                                            \
    if (need > 64 * 1024)                                                                        \
      PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need)); \
    PTHIP_KLAUNCH(k, dim3((unsigned)batch), dim3(BLOCK), need, st, (T*)out, (const T*)Tm,   \
                       sTb, sT0, sT1, (const T*)B, sBb, (int)n, lower, unit);                    \
  } while (0)
      if (n <= 64) LAUNCH_TRSV(1);
      else if (n <= 128) LAUNCH_TRSV(2);
      else LAUNCH_TRSV(4);
#undef LAUNCH_TRSV
      return pthip::post_launch("trsv_lds");
    }
  }
  {
    const size_t need = (size_t)n * (size_t)(n | 1) * sizeof(T);
    // (a single triangle of more than one 64-row block with at least 64 right-hand sides goes to the blocked
    //  MFMA solve below even when it fits the LDS: one thread per column is 8192 dependent steps at n = 128;
    //  batches keep the one-launch LDS kernel)
    static const bool lds_always = getenv("PTHIP_TRSM") && !strcmp(getenv("PTHIP_TRSM"), "lds");
    static const bool few_off = getenv("PTHIP_TRSM_FEW") && !strcmp(getenv("PTHIP_TRSM_FEW"), "0");
    const size_t need_few = need + (size_t)(BLOCK / 64) * 2 * n * sizeof(T);
    if (!few_off && nrhs >= 2 && nrhs <= 16 && n <= 256 && need_few <= 160 * 1024 - 256 && B != out) {
#define LAUNCH_FEW(RPL)                                                                                                            \
  do {                                                                                                                             \
    auto k = trsm_few_lds_kernel<T, RPL>;                                                                                          \
    if (need_few > 64 * 1024)                                                                                                      \
      PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need_few));                 \
    PTHIP_KLAUNCH(k, dim3((unsigned)batch), dim3(BLOCK), need_few, st, (T*)out, (const T*)Tm, sTb, sT0, sT1, (const T*)B, sBb,     \
                  (int)n, (int)nrhs, lower, unit);                                                                                 \
  } while (0)
      if (n <= 64) LAUNCH_FEW(1);
      else if (n <= 128) LAUNCH_FEW(2);
      else LAUNCH_FEW(4);
#undef LAUNCH_FEW
      return pthip::post_launch("trsm_few_lds");
    }
    if (need <= 160 * 1024 - 256 && (lds_always || n <= TV || nrhs < 64 || batch > 2)) {
      auto k = trsm_lds_kernel<T>;
      if (need > 64 * 1024)
        PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      PTHIP_KLAUNCH(k, dim3((unsigned)((nrhs + BLOCK - 1) / BLOCK), (unsigned)batch), dim3(BLOCK), need,
                         st, (T*)out, (const T*)Tm, sTb, sT0, sT1, (const T*)B, sBb, (int)n, (int)nrhs, lower, unit, (int*)nullptr);
      return pthip::post_launch("trsm_lds");
    }
  }
  static const bool generic = getenv("PTHIP_TRSM") && !strcmp(getenv("PTHIP_TRSM"), "generic");
  if (!generic) {
    // beyond the LDS: the persistent row-block solve for a few right-hand sides, blocked GEMM updates for many
    for (long long b = 0; b < batch; b++) {
      const T* Tb = (const T*)Tm + b * sTb;
      const T* Bb = (const T*)B + b * sBb;
      T* Ob = (T*)out + b * n * nrhs;
      int r = (nrhs <= 16 && !pthip::ctx().safe_mode) ? trsv_dag<T>(lower, unit, n, nrhs, Tb, sT0, sT1, Bb, Ob)
                         : trsm_blocked<T>(lower, unit, n, nrhs, Tb, sT0, sT1, Bb, Ob);
      if (r == (1 << 30)) r = trsm_blocked<T>(lower, unit, n, nrhs, Tb, sT0, sT1, Bb, Ob);  // (over-subscribed: see trsv_dag)
      if (r) return r;
    }
    return 0;
  }
  PTHIP_KLAUNCH((trsm_kernel<T>), dim3((unsigned)((nrhs + BLOCK - 1) / BLOCK), (unsigned)batch),
                     dim3(BLOCK), 0, st, (T*)out, (const T*)Tm, sTb, sT0, sT1, (const T*)B, sBb,
                     (int)n, (int)nrhs, lower, unit);
  return pthip::post_launch("trsm");
}

}  // namespace

extern "C" {

int pthip_potrf(int dtype, int lower, int64_t batch, int64_t n, const void* A, void* L) {
  PTHIP_REQUIRE_INIT();
  if (n > 32767) return pthip::set_error("pthip_potrf: n too large");
  if (dtype == PTHIP_F64) return potrf_typed<double>(lower, batch, n, A, L);
  if (dtype == PTHIP_F32) return potrf_typed<float>(lower, batch, n, A, L);
  return pthip::set_error("pthip_potrf: dtype %d not supported", dtype);
}

int pthip_potrf_trsv(int dtype, int64_t batch, int64_t n, const void* A, const void* b, void* L,
                     void* x) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64) return potrf_typed<double>(1, batch, n, A, L, b, x);
  if (dtype == PTHIP_F32) return potrf_typed<float>(1, batch, n, A, L, b, x);
  return pthip::set_error("pthip_potrf_trsv: dtype %d not supported", dtype);
}

int pthip_trsm(int dtype, int lower, int trans, int unit_diag, int64_t batch, int64_t n,
               int64_t nrhs, const void* T, int64_t sTb, int64_t sT0, int64_t sT1, const void* B,
               int64_t sBb, void* out) {
  PTHIP_REQUIRE_INIT();
  if (n > 32767) return pthip::set_error("pthip_trsm: n too large");
  if (trans) {  // op(T) = T^T: swap strides, flip the triangle
    int64_t t = sT0;
    sT0 = sT1;
    sT1 = t;
    lower = !lower;
  }
  if (dtype == PTHIP_F64)
    return trsm_typed<double>(lower, unit_diag, batch, n, nrhs, T, sTb, sT0, sT1, B, sBb, out);
  if (dtype == PTHIP_F32)
    return trsm_typed<float>(lower, unit_diag, batch, n, nrhs, T, sTb, sT0, sT1, B, sBb, out);
  return pthip::set_error("pthip_trsm: dtype %d not supported", dtype);
}

}  // extern "C"
