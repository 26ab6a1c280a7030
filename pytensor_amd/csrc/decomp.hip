// decomp.hip — Householder QR, one-sided Jacobi SVD, tridiagonal LU: the dense decompositions of
// SURVEY §8f row 3 that round 1 left on the host.
//
// Reference: QR.perform (pytensor/tensor/linalg/decomposition/qr.py:153-221: LAPACK geqrf, then
// orgqr for Q), SVD.perform (linalg/decomposition/svd.py: np.linalg.svd = gesdd),
// LUFactorTridiagonal / SolveLUFactorTridiagonal.perform (linalg/solvers/tridiagonal.py:70-90,
// 170-180: LAPACK gttrf / gttrs).
//
// "Correct first" tier: one workgroup per matrix (batches on grid.x), the matrix in global memory
// (L2-resident at the sizes these ops see), every inner loop a row-contiguous sweep so that a wave
// reads whole 512-byte rows.  Conventions are LAPACK's where the result depends on them:
//  * geqrf: H_k = I - tau v v^T, v_k = 1 implicit, beta = -sign(alpha) ||x|| on the diagonal
//    (dlarfg), so R's diagonal carries LAPACK's signs and Q, R match the reference entry by entry;
//  * gttrf / gttrs: dgttrf's pivoting rule (|d_i| >= |dl_i| keeps the row), 1-based ipiv, the
//    same operation order as dgtts2, no fp contraction — bit-for-bit the reference's numbers;
//  * SVD: singular values descending; U / V are defined up to a sign per pair (and up to a basis
//    of the null space / complement), which LAPACK does not fix either.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int VMAX = 4096;  // reflector entries staged in LDS (longer ones are read from L2)

template <class T> __device__ __forceinline__ T dabs(T x) { return x < T(0) ? -x : x; }

template <class T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// the same sum out of DPP row operations and v_readlane (no LDS permutes: __shfl_xor is a ds_bpermute round trip
// per step, six dependent ones per sum) — used where a wave sum sits on the critical path of every Jacobi rotation
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_bcast(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float lane_bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <class T> __device__ __forceinline__ T wave_sum_dpp(T v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row's sum
  return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}

// the sum over each 16-lane row of the wave, in every lane of that row
template <class T> __device__ __forceinline__ T row_sum_dpp(T v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// sum over the block; every thread gets the result (two barriers)
template <class T> __device__ T block_sum(T v, T* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = T(0);
#pragma unroll
  for (int w = 0; w < BLOCK / 64; w++) r += s_red[w];
  return r;
}

// M[r0:m, c0:c1] -= tau * v (v^T M[r0:m, c0:c1]),  v = (1, V[r0+1, vc], ..., V[m-1, vc])
// threads: 64 columns x 4 row slices per pass; s_v: VMAX, s_w: 4 x 64
template <class T>
__device__ void apply_reflector(T* __restrict__ M, long long ld, int r0, int m, int c0, int c1,
                                const T* __restrict__ V, long long ldv, int vc, T tau, T* s_v, T* s_w, int vmax = VMAX) {
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int len = m - r0;
  const bool staged = len <= vmax;
  __syncthreads();
  if (staged)
    for (int i = tid; i < len; i += BLOCK) s_v[i] = i == 0 ? T(1) : V[(long long)(r0 + i) * ldv + vc];
  __syncthreads();
  for (int cc = c0; cc < c1; cc += 64) {
    const int col = cc + tx;
    T acc = T(0);
    if (col < c1) {
#pragma unroll 8
      for (int i = ty; i < len; i += 4) {  // (unrolled: the row loads of eight steps are in flight together)
        const T vi = staged ? s_v[i] : (i == 0 ? T(1) : V[(long long)(r0 + i) * ldv + vc]);
        acc += vi * M[(long long)(r0 + i) * ld + col];
      }
    }
    s_w[ty * 64 + tx] = acc;
    __syncthreads();
    const T w = tau * (s_w[tx] + s_w[64 + tx] + s_w[128 + tx] + s_w[192 + tx]);
    if (col < c1) {
#pragma unroll 8
      for (int i = ty; i < len; i += 4) {
        const T vi = staged ? s_v[i] : (i == 0 ? T(1) : V[(long long)(r0 + i) * ldv + vc]);
        M[(long long)(r0 + i) * ld + col] -= vi * w;
      }
    }
    __syncthreads();
  }
}

// in place: reflectors below the diagonal, R on and above it, tau[min(m,n)]
// use_lds (round 4): the whole matrix in LDS for the factorisation (m n elements of dynamic shared memory; 128 x 128
// fp64 = 128 KB) — every column step makes two passes over the trailing block, from L2 that was 2.9 ms at n = 128.
template <class T, int VM>
__global__ __launch_bounds__(BLOCK) void geqrf_kernel(T* __restrict__ Aall, T* __restrict__ tauall, int m, int n, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char qr_smem[];
  __shared__ T s_v[VM];
  __shared__ T s_w[256];
  __shared__ T s_red[BLOCK / 64];
  T* Ag = Aall + (long long)blockIdx.x * m * n;
  T* A = use_lds ? (T*)qr_smem : Ag;
  const int K = m < n ? m : n;
  T* tau = tauall + (long long)blockIdx.x * K;
  const int tid = threadIdx.x;
  if (use_lds) {
    for (long long e = tid; e < (long long)m * n; e += BLOCK) A[e] = Ag[e];
    __syncthreads();
  }
  for (int k = 0; k < K; k++) {
    T part = T(0);
    for (int i = k + 1 + tid; i < m; i += BLOCK) {
      const T x = A[(long long)i * n + k];
      part += x * x;
    }
    const T xn2 = block_sum(part, s_red);
    const T alpha = A[(long long)k * n + k];
    if (xn2 == T(0)) {  // dlarfg: H = I
      if (tid == 0) tau[k] = T(0);
      continue;
    }
    const T nrm = hypot(alpha, sqrt(xn2));
    const T beta = alpha >= T(0) ? -nrm : nrm;
    const T tk = (beta - alpha) / beta;
    const T scale = T(1) / (alpha - beta);
    __syncthreads();  // (every thread has read alpha)
    for (int i = k + 1 + tid; i < m; i += BLOCK) A[(long long)i * n + k] *= scale;
    if (tid == 0) { A[(long long)k * n + k] = beta; tau[k] = tk; }
    if (k + 1 < n) apply_reflector(A, n, k, m, k + 1, n, A, n, k, tk, s_v, s_w, VM);
    __syncthreads();
  }
  if (use_lds) {
    __syncthreads();
    for (long long e = tid; e < (long long)m * n; e += BLOCK) Ag[e] = A[e];
  }
}

// Q (m x nc) = H_0 H_1 ... H_{k-1} applied to the first nc columns of the identity (dorg2r)
template <class T, int VM>
__global__ __launch_bounds__(BLOCK) void orgqr_kernel(const T* __restrict__ QRall, long long ldqr, long long qr_stride,
                                                     const T* __restrict__ tauall, T* __restrict__ Qall, int m, int nc, int k, int use_lds) {
  __shared__ T s_v[VM];
  __shared__ T s_w[256];
  const T* QR = QRall + (long long)blockIdx.x * qr_stride;
  const T* tau = tauall + (long long)blockIdx.x * k;
  extern __shared__ __attribute__((aligned(16))) unsigned char qr_smem[];
  T* Qg = Qall + (long long)blockIdx.x * m * nc;
  T* Q = use_lds ? (T*)qr_smem : Qg;
  for (long long e = threadIdx.x; e < (long long)m * nc; e += BLOCK) Q[e] = (e / nc == e % nc) ? T(1) : T(0);
  __syncthreads();
  for (int j = k - 1; j >= 0; j--) {
    const T tj = tau[j];
    if (tj != T(0) && j < nc) apply_reflector(Q, nc, j, m, j, nc, QR, ldqr, j, tj, s_v, s_w, VM);
  }
  if (use_lds) {
    __syncthreads();
    for (long long e = threadIdx.x; e < (long long)m * nc; e += BLOCK) Qg[e] = Q[e];
  }
}

// ---- one-sided Jacobi SVD on the rows of X (r x c, r <= c) ------------------------------------
// X = P diag(s) Wt: rotations from the left orthogonalise the rows (Hestenes), Pt accumulates them.
// Parallel order: round-robin tournament, one wave per pair, a barrier per round.
constexpr int SVD_CACHE = 4;      // row values per lane kept in registers (rows up to 256 long)
constexpr int SVD_UP16 = 8;       // row values per lane of a 16-lane row (four pairs per wave; rows up to 128 long)
constexpr int SVD_BLOCK = 1024;  // 16 waves: a round of the tournament has r/2 independent pairs, each a chain of L2 round trips

template <class T>
__global__ __launch_bounds__(SVD_BLOCK) void svd_rows_kernel(T* __restrict__ Xall, T* __restrict__ Ptall, T* __restrict__ Sall,
                                                        T* __restrict__ Wtall, T* __restrict__ Poutall, int r, int c,
                                                        int want_vectors, int* __restrict__ status) {
  __shared__ int s_rot;
  __shared__ T s_norm[2048];  // r <= 2048 (checked by the caller)
  // squared row norms carried along analytically through the rotations (a' = a - t g, b' = b + t g) and recomputed at
  // the start of every sweep: one wave sum per pair instead of three (short rows; round 4).
  // (measured and dropped: X itself in LDS when it fits — 14.3 ms either way at 128 x 128: the rows of a pair already
  //  arrive in one L2 round trip, the time is in the sweeps)
  __shared__ T s_n2[2048];
  T* X = Xall + (long long)blockIdx.x * r * c;
  T* Pt = Ptall + (long long)blockIdx.x * r * r;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const T eps = sizeof(T) == 8 ? T(2.220446049250313e-16) : T(1.1920929e-07);
  if (want_vectors) {
    for (long long e = tid; e < (long long)r * r; e += SVD_BLOCK) Pt[e] = (e / r == e % r) ? T(1) : T(0);
  }
  __syncthreads();
  const int R = (r + 1) & ~1;  // players (one bye when r is odd)
  bool converged = r < 2;
  for (int sweep = 0; sweep < 60 && !converged; sweep++) {
    if (tid == 0) s_rot = 0;
    if (c <= 64 * SVD_CACHE) {
      for (int i = wid; i < r; i += SVD_BLOCK / 64) {
        T a = T(0);
        for (int j = lane; j < c; j += 64) { const T u = X[(long long)i * c + j]; a += u * u; }
        a = wave_sum_dpp(a);
        if (lane == 0) s_n2[i] = a;
      }
    }
    __syncthreads();
    for (int step = 0; step < R - 1; step++) {
      if (c <= 16 * SVD_UP16 && r <= 16 * SVD_UP16) {
        // rows up to 128 long: FOUR pairs per wave and step, one per 16-lane row — a pair's dot product is four DPP
        // steps inside its row, and the angles and rotations of all four run in the same instructions (the same
        // restructuring took Eigh(128) from 4.9 to 2.1 ms).  Branch-free up to the stores.
        for (int i0 = 4 * wid; i0 < R / 2; i0 += 4 * (SVD_BLOCK / 64)) {
          const int l16 = lane & 15, idx = i0 + (lane >> 4);
          int p = 0, q = 0;
          bool real = idx < R / 2;
          if (real) {
            if (idx == 0) { p = R - 1; q = step; }
            else { p = (step + idx) % (R - 1); q = (step - idx + (R - 1)) % (R - 1); }
            real = p < r && q < r;
            if (p > q) { const int t = p; p = q; q = t; }
            if (!real) { p = 0; q = 0; }
          }
          T* xp = X + (long long)p * c;
          T* xq = X + (long long)q * c;
          T* pp = Pt + (long long)p * r;
          T* pq = Pt + (long long)q * r;
          T xu[SVD_UP16], xv[SVD_UP16], pu[SVD_UP16], pv[SVD_UP16];
          T g = T(0);
#pragma unroll
          for (int t = 0; t < SVD_UP16; t++) {
            const int j = l16 + 16 * t;
            xu[t] = (real && j < c) ? xp[j] : T(0);
            xv[t] = (real && j < c) ? xq[j] : T(0);
            pu[t] = (real && want_vectors && j < r) ? pp[j] : T(0);
            pv[t] = (real && want_vectors && j < r) ? pq[j] : T(0);
            g += xu[t] * xv[t];
          }
          const T a = s_n2[p], b = s_n2[q];
          g = row_sum_dpp(g);
          const bool rotate = real && g != T(0) && g * g > eps * eps * a * b;
          const float gf = rotate ? (float)g : 1.f;
          const float zf = (float)(b - a) / (2.f * gf);
          float tf = (zf >= 0.f ? 1.f : -1.f) / ((zf < 0.f ? -zf : zf) + __builtin_sqrtf(1.f + zf * zf));
          T tt = (T)tf;
          if (rotate && !(tf != 0.f && tf == tf)) {
            const T zeta = (b - a) / (T(2) * g);
            tt = (zeta >= T(0) ? T(1) : T(-1)) / (dabs(zeta) + sqrt(T(1) + zeta * zeta));
          }
          if (!rotate) tt = T(0);
          T cs;
          if constexpr (sizeof(T) == 8) {
            const double xx = 1.0 + tt * tt;
            double y = __builtin_amdgcn_rsq(xx);
            y = y * (1.5 - 0.5 * xx * y * y);
            y = y * (1.5 - 0.5 * xx * y * y);
            cs = y;
          } else {
            cs = T(1) / __builtin_sqrtf(T(1) + tt * tt);
          }
          const T sn = cs * tt;
          if (rotate) {  // (uniform inside a 16-lane row: the DPP sums below stay inside rows)
            T za = T(0), zb = T(0);
#pragma unroll
            for (int t = 0; t < SVD_UP16; t++) {
              const int j = l16 + 16 * t;
              const T nu = cs * xu[t] - sn * xv[t], nv = sn * xu[t] + cs * xv[t];
              za += nu * nu;
              zb += nv * nv;
              if (j < c) { xp[j] = nu; xq[j] = nv; }
              if (want_vectors && j < r) { pp[j] = cs * pu[t] - sn * pv[t]; pq[j] = sn * pu[t] + cs * pv[t]; }
            }
            T na = a - tt * g, nb = b + tt * g;
            if (!(na > T(0.01) * a)) na = row_sum_dpp(za);  // (cancellation: summed again from the rotated row)
            if (!(nb > T(0.01) * b)) nb = row_sum_dpp(zb);
            if (l16 == 0) { s_n2[p] = na; s_n2[q] = nb; s_rot = 1; }
          }
        }
        __syncthreads();
        continue;
      }
      for (int idx = wid; idx < R / 2; idx += SVD_BLOCK / 64) {
        int p, q;
        if (idx == 0) { p = R - 1; q = step; }
        else { p = (step + idx) % (R - 1); q = (step - idx + (R - 1)) % (R - 1); }
        if (p >= r || q >= r) continue;
        if (p > q) { const int t = p; p = q; q = t; }
        T* xp = X + (long long)p * c;
        T* xq = X + (long long)q * c;
        T* pp = Pt + (long long)p * r;
        T* pq = Pt + (long long)q * r;
        if (c <= 64 * SVD_CACHE) {
          // short rows: one trip to memory per pair — both rows of X and of Pt are requested up front
          // and stay in registers through the dot products and the rotation
          T xu[SVD_CACHE], xv[SVD_CACHE], pu[SVD_CACHE], pv[SVD_CACHE];
#pragma unroll
          for (int t = 0; t < SVD_CACHE; t++) {
            const int j = lane + 64 * t;
            xu[t] = j < c ? xp[j] : T(0);
            xv[t] = j < c ? xq[j] : T(0);
            pu[t] = (want_vectors && j < r) ? pp[j] : T(0);
            pv[t] = (want_vectors && j < r) ? pq[j] : T(0);
          }
          T g = T(0);
#pragma unroll
          for (int t = 0; t < SVD_CACHE; t++) g += xu[t] * xv[t];
          const T a = s_n2[p], b = s_n2[q];
          g = wave_sum_dpp(g);
          if (g == T(0) || g * g <= eps * eps * a * b) continue;
          // the angle in single precision (it steers convergence only), the rotation in working precision
          const float zf = (float)(b - a) / (2.f * (float)g);
          float tf = (zf >= 0.f ? 1.f : -1.f) / ((zf < 0.f ? -zf : zf) + __builtin_sqrtf(1.f + zf * zf));
          T tt = (T)tf;
          if (!(tf != 0.f && tf == tf)) {
            // out of single precision's range (a graded matrix: |g| or the ratio under / overflows): the same formula in T
            const T zeta = (b - a) / (T(2) * g);
            tt = (zeta >= T(0) ? T(1) : T(-1)) / (dabs(zeta) + sqrt(T(1) + zeta * zeta));
          }
          T cs;
          if constexpr (sizeof(T) == 8) {
            const double xx = 1.0 + tt * tt;
            double y = __builtin_amdgcn_rsq(xx);
            y = y * (1.5 - 0.5 * xx * y * y);
            y = y * (1.5 - 0.5 * xx * y * y);
            cs = y;
          } else {
            cs = T(1) / __builtin_sqrtf(T(1) + tt * tt);
          }
          const T sn = cs * tt;
          T na = a - tt * g, nb = b + tt * g;
#pragma unroll
          for (int t = 0; t < SVD_CACHE; t++) {
            const int j = lane + 64 * t;
            const T nu = cs * xu[t] - sn * xv[t], nv = sn * xu[t] + cs * xv[t];
            xu[t] = nu;
            xv[t] = nv;
            if (j < c) { xp[j] = nu; xq[j] = nv; }
            if (want_vectors && j < r) { pp[j] = cs * pu[t] - sn * pv[t]; pq[j] = sn * pu[t] + cs * pv[t]; }
          }
          // the analytic update cancels when a row shrinks (a rank-deficient or graded matrix: the small singular values
          // are exactly what a one-sided method is accurate for): such a norm is summed again from the rotated row
          if (!(na > T(0.01) * a)) {
            T z = T(0);
#pragma unroll
            for (int t = 0; t < SVD_CACHE; t++) z += xu[t] * xu[t];
            na = wave_sum_dpp(z);
          }
          if (!(nb > T(0.01) * b)) {
            T z = T(0);
#pragma unroll
            for (int t = 0; t < SVD_CACHE; t++) z += xv[t] * xv[t];
            nb = wave_sum_dpp(z);
          }
          if (lane == 0) { s_n2[p] = na; s_n2[q] = nb; s_rot = 1; }
          continue;
        }
        T a = T(0), b = T(0), g = T(0);
        for (int j = lane; j < c; j += 64) {
          const T u = xp[j], v = xq[j];
          a += u * u; b += v * v; g += u * v;
        }
        a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
        if (g == T(0) || dabs(g) <= eps * sqrt(a) * sqrt(b)) continue;
        const T zeta = (b - a) / (T(2) * g);
        const T t = (zeta >= T(0) ? T(1) : T(-1)) / (dabs(zeta) + sqrt(T(1) + zeta * zeta));
        const T cs = T(1) / sqrt(T(1) + t * t), sn = cs * t;
        for (int j = lane; j < c; j += 64) {
          const T u = xp[j], v = xq[j];
          xp[j] = cs * u - sn * v;
          xq[j] = sn * u + cs * v;
        }
        if (want_vectors) {
          for (int j = lane; j < r; j += 64) {
            const T u = pp[j], v = pq[j];
            pp[j] = cs * u - sn * v;
            pq[j] = sn * u + cs * v;
          }
        }
        if (lane == 0) s_rot = 1;
      }
      __syncthreads();
    }
    converged = s_rot == 0;
    __syncthreads();
  }
  if (!converged && tid == 0 && status) atomicOr(status, 4);
  // singular values = row norms; rank them descending (ties by position)
  for (int i = wid; i < r; i += SVD_BLOCK / 64) {
    T a = T(0);
    for (int j = lane; j < c; j += 64) { const T u = X[(long long)i * c + j]; a += u * u; }
    a = wave_sum(a);
    if (lane == 0) s_norm[i] = sqrt(a);
  }
  __syncthreads();
  T* S = Sall + (long long)blockIdx.x * r;
  T* Wt = Wtall + (long long)blockIdx.x * r * c;
  T* Pout = Poutall + (long long)blockIdx.x * r * r;
  for (int i = wid; i < r; i += SVD_BLOCK / 64) {
    const T si = s_norm[i];
    int rank = 0;
    for (int j = lane; j < r; j += 64) rank += (s_norm[j] > si || (s_norm[j] == si && j < i)) ? 1 : 0;
    rank = (int)wave_sum((T)rank);
    if (lane == 0) S[rank] = si;
    if (want_vectors) {
      const T inv = si > T(0) ? T(1) / si : T(0);
      for (int j = lane; j < c; j += 64) Wt[(long long)rank * c + j] = X[(long long)i * c + j] * inv;
      for (int j = lane; j < r; j += 64) Pout[(long long)rank * r + j] = Pt[(long long)i * r + j];
    }
  }
}

// rows of Wt that came out zero (singular value 0) are replaced by the same rows of Qt, an
// orthonormal completion (transposed Householder Q of Wt^T): rows stay orthonormal
template <class T>
__global__ void fill_null_rows_kernel(T* __restrict__ Wt, const T* __restrict__ S, int r_valid, const T* __restrict__ Q,
                                      long long batch, int r, int c, int ldq) {
  const long long total = batch * r * c;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / ((long long)r * c);
    const int i = (int)((e / c) % r), j = (int)(e % c);
    if (i >= r_valid || !(S[b * r_valid + i] > T(0))) Wt[e] = Q[b * (long long)c * ldq + (long long)j * ldq + i];  // Q is c x ldq, column i
  }
}

// dst (rows x cols, contiguous) = one triangle of src's leading rows (row stride ld), zeros in the
// other; unit: the diagonal is 1 (the L of a packed LU)
template <class T>
__global__ void triu_kernel(T* __restrict__ dst, const T* __restrict__ src, long long batch, int rows, int cols, long long ld,
                            long long src_stride, int lower, int unit) {
  const long long total = batch * rows * cols;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / ((long long)rows * cols);
    const int i = (int)((e / cols) % rows), j = (int)(e % cols);
    const bool keep = lower ? j <= i : j >= i;
    T v = keep ? src[b * src_stride + (long long)i * ld + j] : T(0);
    if (unit && i == j) v = T(1);
    dst[e] = v;
  }
}

// ---- tridiagonal LU (dgttrf) and its solves (dgtts2); one thread per system / right-hand side ----
template <class T>
__global__ void gttrf_kernel(long long batch, int n, T* __restrict__ dl, T* __restrict__ d, T* __restrict__ du,
                             T* __restrict__ du2, int* __restrict__ ipiv) {
#pragma clang fp contract(off)
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= batch) return;
  dl += s * (n - 1); du += s * (n - 1); d += s * n; ipiv += s * n;
  if (n > 2) du2 += s * (n - 2);
  for (int i = 0; i < n; i++) ipiv[i] = i + 1;
  for (int i = 0; i < n - 2; i++) du2[i] = T(0);
  for (int i = 0; i < n - 1; i++) {
    const bool last = i == n - 2;
    if (dabs(d[i]) >= dabs(dl[i])) {
      if (d[i] != T(0)) {
        const T fact = dl[i] / d[i];
        dl[i] = fact;
        d[i + 1] = d[i + 1] - fact * du[i];
      }
    } else {
      const T fact = d[i] / dl[i];
      d[i] = dl[i];
      dl[i] = fact;
      const T temp = du[i];
      du[i] = d[i + 1];
      d[i + 1] = temp - fact * d[i + 1];
      if (!last) {
        du2[i] = du[i + 1];
        du[i + 1] = -fact * du[i + 1];
      }
      ipiv[i] = i + 2;
    }
  }
}

template <class T>
__global__ void gttrs_kernel(long long batch, int n, int nrhs, int trans, const T* __restrict__ dl, const T* __restrict__ d,
                             const T* __restrict__ du, const T* __restrict__ du2, const int* __restrict__ ipiv,
                             T* __restrict__ B) {
#pragma clang fp contract(off)
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * nrhs) return;
  const long long s = e / nrhs;
  const int col = (int)(e % nrhs);
  dl += s * (n - 1); du += s * (n - 1); d += s * n; ipiv += s * n;
  if (n > 2) du2 += s * (n - 2);
  T* b = B + s * (long long)n * nrhs + col;
  const long long ld = nrhs;
  if (!trans) {
    for (int i = 0; i < n - 1; i++) {
      const int ip = ipiv[i] - 1;
      const T temp = b[(i + 1 - ip + i) * ld] - dl[i] * b[ip * ld];
      b[i * ld] = b[ip * ld];
      b[(i + 1) * ld] = temp;
    }
    b[(n - 1) * ld] = b[(n - 1) * ld] / d[n - 1];
    if (n > 1) b[(n - 2) * ld] = (b[(n - 2) * ld] - du[n - 2] * b[(n - 1) * ld]) / d[n - 2];
    for (int i = n - 3; i >= 0; i--) b[i * ld] = (b[i * ld] - du[i] * b[(i + 1) * ld] - du2[i] * b[(i + 2) * ld]) / d[i];
  } else {
    b[0] = b[0] / d[0];
    if (n > 1) b[ld] = (b[ld] - du[0] * b[0]) / d[1];
    for (int i = 2; i < n; i++) b[i * ld] = (b[i * ld] - du[i - 1] * b[(i - 1) * ld] - du2[i - 2] * b[(i - 2) * ld]) / d[i];
    for (int i = n - 2; i >= 0; i--) {
      const int ip = ipiv[i] - 1;
      const T temp = b[i * ld] - dl[i] * b[(i + 1) * ld];
      b[i * ld] = b[ip * ld];
      b[ip * ld] = temp;
    }
  }
}

constexpr size_t QR_LDS_MAX = 160 * 1024 - 12 * 1024;  // what is left beside the kernels' static arrays (s_v[512] in this form)
static const bool qr_no_lds = getenv("PTHIP_QR_NO_LDS") != nullptr;

template <class T> int geqrf_typed(long long batch, int m, int n, void* A, void* tau) {
  const size_t need = (size_t)m * n * sizeof(T);
  if (need <= QR_LDS_MAX && m <= 512 && !qr_no_lds) {
    auto k = geqrf_kernel<T, 512>;
    static bool attr = false;
    if (!attr) {
      PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QR_LDS_MAX));
      attr = true;
    }
    PTHIP_KLAUNCH(k, dim3((unsigned)batch), dim3(BLOCK), need, pthip::ctx().stream, (T*)A, (T*)tau, m, n, 1);
    return pthip::post_launch("geqrf(lds)");
  }
  PTHIP_KLAUNCH((geqrf_kernel<T, VMAX>), dim3((unsigned)batch), dim3(BLOCK), 0, pthip::ctx().stream, (T*)A, (T*)tau, m, n, 0);
  return pthip::post_launch("geqrf");
}

template <class T> int orgqr_typed(long long batch, int m, int nc, int k, const void* QR, long long ldqr, long long stride,
                                   const void* tau, void* Q) {
  const size_t need = (size_t)m * nc * sizeof(T);
  if (need <= QR_LDS_MAX && m <= 512 && !qr_no_lds) {
    auto kk = orgqr_kernel<T, 512>;
    static bool attr = false;
    if (!attr) {
      PTHIP_CHECK(hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)QR_LDS_MAX));
      attr = true;
    }
    PTHIP_KLAUNCH(kk, dim3((unsigned)batch), dim3(BLOCK), need, pthip::ctx().stream, (const T*)QR, ldqr, stride, (const T*)tau, (T*)Q, m, nc, k, 1);
    return pthip::post_launch("orgqr(lds)");
  }
  PTHIP_KLAUNCH((orgqr_kernel<T, VMAX>), dim3((unsigned)batch), dim3(BLOCK), 0, pthip::ctx().stream, (const T*)QR, ldqr, stride,
                (const T*)tau, (T*)Q, m, nc, k, 0);
  return pthip::post_launch("orgqr");
}

template <class T> int svd_typed(long long batch, int r, int c, int vectors, void* X, void* Pt, void* S, void* Wt, void* Pout) {
  PTHIP_KLAUNCH(svd_rows_kernel<T>, dim3((unsigned)batch), dim3(SVD_BLOCK), 0, pthip::ctx().stream, (T*)X, (T*)Pt, (T*)S, (T*)Wt,
                (T*)Pout, r, c, vectors, pthip::ctx().status_dev);
  return pthip::post_launch("svd_rows");
}

}  // namespace

extern "C" int pthip_geqrf(int dtype, int64_t batch, int64_t m, int64_t n, void* A, void* tau) {
  PTHIP_REQUIRE_INIT();
  if (batch <= 0 || m <= 0 || n <= 0) return 0;
  if (dtype == PTHIP_F64) return geqrf_typed<double>(batch, (int)m, (int)n, A, tau);
  if (dtype == PTHIP_F32) return geqrf_typed<float>(batch, (int)m, (int)n, A, tau);
  return pthip::set_error("pthip_geqrf: dtype %d not supported (float32/float64 only)", dtype);
}

extern "C" int pthip_orgqr(int dtype, int64_t batch, int64_t m, int64_t ncols, int64_t k, const void* QR, int64_t ldqr,
                           int64_t qr_stride, const void* tau, void* Q) {
  PTHIP_REQUIRE_INIT();
  if (batch <= 0 || m <= 0 || ncols <= 0) return 0;
  if (dtype == PTHIP_F64) return orgqr_typed<double>(batch, (int)m, (int)ncols, (int)k, QR, ldqr, qr_stride, tau, Q);
  if (dtype == PTHIP_F32) return orgqr_typed<float>(batch, (int)m, (int)ncols, (int)k, QR, ldqr, qr_stride, tau, Q);
  return pthip::set_error("pthip_orgqr: dtype %d not supported (float32/float64 only)", dtype);
}

extern "C" int pthip_svd_rows(int dtype, int64_t batch, int64_t r, int64_t c, int vectors, void* X, void* Pt_work, void* S,
                              void* Wt, void* Pt) {
  PTHIP_REQUIRE_INIT();
  if (batch <= 0 || r <= 0) return 0;
  if (r > c) return pthip::set_error("pthip_svd_rows: %lld rows > %lld columns (pass the transpose)", (long long)r, (long long)c);
  if (r > 2048) return pthip::set_error("pthip_svd_rows: min(m, n) = %lld above 2048", (long long)r);
  if (dtype == PTHIP_F64) return svd_typed<double>(batch, (int)r, (int)c, vectors, X, Pt_work, S, Wt, Pt);
  if (dtype == PTHIP_F32) return svd_typed<float>(batch, (int)r, (int)c, vectors, X, Pt_work, S, Wt, Pt);
  return pthip::set_error("pthip_svd_rows: dtype %d not supported (float32/float64 only)", dtype);
}

extern "C" int pthip_triu(int dtype, int64_t batch, int64_t rows, int64_t cols, const void* src, int64_t ld,
                          int64_t src_stride, void* dst, int lower, int unit_diag) {
  PTHIP_REQUIRE_INIT();
  const long long total = (long long)batch * rows * cols;
  if (total <= 0) return 0;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipStream_t st = pthip::ctx().stream;
  if (dtype == PTHIP_F64) PTHIP_KLAUNCH(triu_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)dst, (const double*)src, (long long)batch, (int)rows, (int)cols, (long long)ld, (long long)src_stride, lower, unit_diag);
  else if (dtype == PTHIP_F32) PTHIP_KLAUNCH(triu_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)dst, (const float*)src, (long long)batch, (int)rows, (int)cols, (long long)ld, (long long)src_stride, lower, unit_diag);
  else return pthip::set_error("pthip_triu: dtype %d not supported", dtype);
  return pthip::post_launch("triu");
}

extern "C" int pthip_fill_null_rows(int dtype, int64_t batch, int64_t r, int64_t c, void* Wt, const void* S, int64_t r_valid,
                                    const void* Q, int64_t ldq) {
  PTHIP_REQUIRE_INIT();
  const long long total = (long long)batch * r * c;
  if (total <= 0) return 0;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipStream_t st = pthip::ctx().stream;
  if (dtype == PTHIP_F64) PTHIP_KLAUNCH(fill_null_rows_kernel<double>, dim3(grid), dim3(256), 0, st, (double*)Wt, (const double*)S, (int)r_valid, (const double*)Q, (long long)batch, (int)r, (int)c, (int)ldq);
  else if (dtype == PTHIP_F32) PTHIP_KLAUNCH(fill_null_rows_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)Wt, (const float*)S, (int)r_valid, (const float*)Q, (long long)batch, (int)r, (int)c, (int)ldq);
  else return pthip::set_error("pthip_fill_null_rows: dtype %d not supported", dtype);
  return pthip::post_launch("fill_null_rows");
}

extern "C" int pthip_gttrf(int dtype, int64_t batch, int64_t n, void* dl, void* d, void* du, void* du2, void* ipiv) {
  PTHIP_REQUIRE_INIT();
  if (batch <= 0 || n <= 0) return 0;
  const int grid = (int)((batch + 63) / 64);
  hipStream_t st = pthip::ctx().stream;
  if (dtype == PTHIP_F64) PTHIP_KLAUNCH(gttrf_kernel<double>, dim3(grid), dim3(64), 0, st, (long long)batch, (int)n, (double*)dl, (double*)d, (double*)du, (double*)du2, (int*)ipiv);
  else if (dtype == PTHIP_F32) PTHIP_KLAUNCH(gttrf_kernel<float>, dim3(grid), dim3(64), 0, st, (long long)batch, (int)n, (float*)dl, (float*)d, (float*)du, (float*)du2, (int*)ipiv);
  else return pthip::set_error("pthip_gttrf: dtype %d not supported (float32/float64 only)", dtype);
  return pthip::post_launch("gttrf");
}

extern "C" int pthip_gttrs(int dtype, int64_t batch, int64_t n, int64_t nrhs, int trans, const void* dl, const void* d,
                           const void* du, const void* du2, const void* ipiv, void* B) {
  PTHIP_REQUIRE_INIT();
  if (batch <= 0 || n <= 0 || nrhs <= 0) return 0;
  const long long total = (long long)batch * nrhs;
  const int grid = (int)((total + 63) / 64);
  hipStream_t st = pthip::ctx().stream;
  if (dtype == PTHIP_F64) PTHIP_KLAUNCH(gttrs_kernel<double>, dim3(grid), dim3(64), 0, st, (long long)batch, (int)n, (int)nrhs, trans, (const double*)dl, (const double*)d, (const double*)du, (const double*)du2, (const int*)ipiv, (double*)B);
  else if (dtype == PTHIP_F32) PTHIP_KLAUNCH(gttrs_kernel<float>, dim3(grid), dim3(64), 0, st, (long long)batch, (int)n, (int)nrhs, trans, (const float*)dl, (const float*)d, (const float*)du, (const float*)du2, (const int*)ipiv, (float*)B);
  else return pthip::set_error("pthip_gttrs: dtype %d not supported (float32/float64 only)", dtype);
  return pthip::post_launch("gttrs");
}
