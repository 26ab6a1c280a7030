// eigh.hip — symmetric eigendecomposition by parallel cyclic Jacobi rotations.
//
// Reference: Eigh.perform (pytensor/tensor/linalg/decomposition/eigen.py:177-195:
// scipy.linalg.eigh(a, lower=..., driver=evr|evd) — LAPACK syevr/syevd): eigenvalues ascending,
// eigenvectors as columns, only the `lower` (or upper) triangle of the input is read.  The sign
// of an eigenvector is not defined by the reference either; parity is checked on sign-free
// quantities (w, |v|, V f(w) V^T, gradients).
//
// correct first.  LAPACK's tridiagonalisation + MRRR is a chain of
// short dependent steps; on a GPU the two-sided Jacobi method is the natural small-matrix
// algorithm: a round-robin ordering gives n/2 disjoint (p, q) pairs per round whose rotations
// are applied together, every phase is a conflict-free LDS sweep (leading dimension n|1), and
// the result has high relative accuracy.  One workgroup per matrix (batches on grid.x); A and
// the accumulated V live in LDS when they fit (n <= 96 fp64 / 136 fp32), otherwise V (then A)
// moves to an L2-resident global scratch.  A round is three phases: rotation angles (one thread
// per pair), column update of A and V, row update of A.  Sweeps repeat until one passes without
// a rotation; no convergence within MAX_SWEEPS raises bit 1 of the device error word
// (scipy raises LinAlgError when LAPACK reports non-convergence).
#include "common.h"

#include <cstring>
#include <utility>
#include <vector>

namespace {

constexpr int MAX_SWEEPS = 60;
constexpr int MAX_N = 512;

template <class T> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; };

// d^2 + h^2 inside these bounds is a normal number whose square root loses nothing
template <class T> struct SafeRange;
template <> struct SafeRange<double> { static constexpr double lo = 1e-290, hi = 1e290; };
template <> struct SafeRange<float> { static constexpr float lo = 1e-30f, hi = 1e30f; };
static __device__ __forceinline__ double jacobi_rsqrt(double x) { return rsqrt(x); }
static __device__ __forceinline__ float jacobi_rsqrt(float x) { return rsqrtf(x); }

// BLOCK: 256 for small matrices (several workgroups per CU when batched), 1024 above n = 64 — a round
// has n/2 independent pairs and every pair is a chain of LDS / L2 round trips, so waves are what
// shortens it (n = 128: 15.2 -> 12.0 ms with 16 waves instead of 4)
template <class T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void eigh_jacobi_kernel(const T* __restrict__ Ain, T* __restrict__ Wout,
                                                           T* __restrict__ Vout, int n, int lower, int a_lds,
                                                           int v_lds, T* scratchA, T* scratchV,
                                                           int* __restrict__ status, int sorted, int max_sweeps,
                                                           int cross) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ T s_c[MAX_N / 2], s_s[MAX_N / 2];
  __shared__ short s_p[MAX_N / 2], s_q[MAX_N / 2], s_rank[MAX_N];
  __shared__ T s_red[BLOCK / 64];
  __shared__ int s_rot;
  const long long mat = blockIdx.x;
  const int ld = n | 1;
  T* A = a_lds ? (T*)smem_raw : scratchA + mat * (long long)n * ld;
  T* V = v_lds ? (T*)smem_raw + (a_lds ? (size_t)n * ld : 0) : scratchV + mat * (long long)n * ld;
  const T* Ag = Ain + mat * (long long)n * n;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // symmetric load from the chosen triangle; V = I; Frobenius norm for the absolute threshold
  T fro = T(0);
  for (int i = wid; i < n; i += BLOCK / 64)
    for (int j = lane; j < n; j += 64) {
      const int r = lower ? (i > j ? i : j) : (i < j ? i : j);
      const int c = lower ? (i > j ? j : i) : (i < j ? j : i);
      const T a = Ag[(long long)r * n + c];
      A[i * ld + j] = a;
      V[i * ld + j] = i == j ? T(1) : T(0);
      fro += a * a;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) fro += __shfl_xor(fro, o);
  if (lane == 0) s_red[wid] = fro;
  if (tid == 0) s_rot = 0;
  __syncthreads();
  fro = T(0);
  for (int w = 0; w < BLOCK / 64; w++) fro += s_red[w];
  // below this, a rotation changes nothing at working precision relative to ||A||
  const T tiny_abs = Eps<T>::v * T(1.0 / 1024) * sqrt(fro);

  const int m = (n + 1) & ~1, half = m >> 1;
  bool converged = n < 2;
  for (int sweep = 0; sweep < max_sweeps && !converged; sweep++) {
    // cross != 0 (n even; the block method's subproblems after a sweep's first round): only the pairs (i, n/2 + j) —
    // n/2 rounds of n/2 disjoint pairs, round r pairs i with n/2 + (i + r) mod n/2 — the two halves are left as they are
    const int nrounds = cross ? half : m - 1;
    for (int r = 0; r < nrounds; r++) {
      // ---- phase 1: one thread per pair computes its rotation ----
      for (int i = tid; i < half; i += BLOCK) {
        int p, q;
        if (cross) { p = i; q = i + r; if (q >= half) q -= half; q += half; }
        else if (i == 0) { p = r; q = m - 1; }
        else { p = r + i; if (p >= m - 1) p -= m - 1; q = r - i; if (q < 0) q += m - 1; }
        if (p > q) { const int t = p; p = q; q = t; }
        T c = T(1), s = T(0);
        if (q < n) {
          const T app = A[p * ld + p], aqq = A[q * ld + q], apq = A[p * ld + q];
          const T mag = apq < T(0) ? -apq : apq;
          const T dd = app * aqq;
          if (mag > tiny_abs && mag > Eps<T>::v * sqrt(dd < T(0) ? -dd : dd)) {
            // t = tan(phi), the smaller root of t^2 + 2 theta t - 1 = 0 with theta = (aqq - app) / (2 apq):
            // t = sgn(theta) / (|theta| + sqrt(theta^2 + 1)) = sgn(theta) |h| / (|d| + sqrt(d^2 + h^2)), d = aqq - app,
            // h = 2 apq — one square root and one division in the dependent chain instead of two and two (this chain,
            // on one wave, is a third of a round); the textbook form only where d^2 + h^2 leaves the normal range
            const T d = aqq - app, h = apq + apq;
            const T r2 = d * d + h * h;
            T t;
            if (r2 > SafeRange<T>::lo && r2 < SafeRange<T>::hi) {
              t = (h < T(0) ? -h : h) / ((d < T(0) ? -d : d) + sqrt(r2));
              if ((d < T(0) && h > T(0)) || (d > T(0) && h < T(0))) t = -t;
            } else {
              const T theta = d / h;
              const T at = theta < T(0) ? -theta : theta;
              t = T(1) / (at + sqrt(theta * theta + T(1)));
              if (theta < T(0)) t = -t;
            }
            c = jacobi_rsqrt(t * t + T(1));
            s = t * c;
            s_rot = 1;
          }
        }
        s_c[i] = c; s_s[i] = s; s_p[i] = (short)p; s_q[i] = (short)q;
      }
      __syncthreads();
      // ---- phase 2: A <- J^T A J and V <- V J in ONE pass.  The round's rotations act on disjoint index pairs, so the
      //      2 x 2 block of A at (pair P, pair Q) depends on nothing but itself and the two rotations: a thread reads
      //      its four elements, rotates the columns (Q), then the rows (P) — the arithmetic and its order are those of a
      //      column pass followed by a row pass — and writes them back; no barrier between a column phase and a row
      //      phase, one LDS round trip instead of two.  A ghost index (odd n: q == n) has c = 1, s = 0 and no storage.
      constexpr int NW = BLOCK / 64;
      {
        const float inv_half = 1.0f / (float)half;
        for (int e = tid; e < half * half; e += BLOCK) {
          const int P = (int)(((float)e + 0.5f) * inv_half), Q = e - P * half;  // (exact: e < 2^16, half <= 256)
          const T sP = s_s[P], sQ = s_s[Q];
          if (sP == T(0) && sQ == T(0)) continue;
          const T cP = s_c[P], cQ = s_c[Q];
          const int p1 = s_p[P], p2 = s_q[P], q1 = s_p[Q], q2 = s_q[Q];
          const bool okP = p2 < n, okQ = q2 < n;
          const T x11 = A[p1 * ld + q1], x12 = okQ ? A[p1 * ld + q2] : T(0);
          const T x21 = okP ? A[p2 * ld + q1] : T(0), x22 = (okP && okQ) ? A[p2 * ld + q2] : T(0);
          const T a11 = cQ * x11 - sQ * x12, a12 = sQ * x11 + cQ * x12;
          const T a21 = cQ * x21 - sQ * x22, a22 = sQ * x21 + cQ * x22;
          T b11 = cP * a11 - sP * a21, b21 = sP * a11 + cP * a21;
          T b12 = cP * a12 - sP * a22, b22 = sP * a12 + cP * a22;
          if (P == Q) { b12 = T(0); b21 = T(0); }  // the annihilated pair is stored as an exact zero
          A[p1 * ld + q1] = b11;
          if (okQ) A[p1 * ld + q2] = b12;
          if (okP) A[p2 * ld + q1] = b21;
          if (okP && okQ) A[p2 * ld + q2] = b22;
        }
        // (V is kept transposed: eigenvector p is row p — contiguous in k, which is what the global-scratch case needs
        //  to stay coalesced)
        for (int Q = wid; Q < half; Q += NW) {
          const T sQ = s_s[Q], cQ = s_c[Q];
          if (sQ == T(0)) continue;
          const int q1 = s_p[Q], q2 = s_q[Q];
          for (int k = lane; k < n; k += 64) {
            const T vx = V[q1 * ld + k], vy = V[q2 * ld + k];
            V[q1 * ld + k] = cQ * vx - sQ * vy;
            V[q2 * ld + k] = sQ * vx + cQ * vy;
          }
        }
      }
      __syncthreads();
    }
    const int rot = s_rot;
    __syncthreads();
    if (tid == 0) s_rot = 0;
    converged = rot == 0;
    __syncthreads();
  }
  if (!converged && tid == 0 && status != nullptr && max_sweeps >= MAX_SWEEPS) atomicOr(status, 8);  // bit 3: LinAlgError("Eigenvalues did not converge")

  // ascending order: rank of each eigenvalue (ties by index), then the permuted write
  T* Wg = Wout + mat * (long long)n;
  T* Vg = Vout + mat * (long long)n * n;
  for (int i = tid; i < n; i += BLOCK) {
    const T wi = A[i * ld + i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
      const T wj = A[j * ld + j];
      rank += (wj < wi || (wj == wi && j < i)) ? 1 : 0;
    }
    if (wi != wi) rank = i;  // NaN input: every slot still gets written
    // (sorted == 0: eigenpair i stays in slot i — the accumulated rotation matrix, close to the identity when the
    //  input is nearly diagonal: what the block method below needs from its subproblems to converge)
    if (!sorted) rank = i;
    Wg[rank] = wi;
    s_rank[i] = (short)rank;
  }
  __syncthreads();
  for (int k = wid; k < n; k += BLOCK / 64)
    for (int i = lane; i < n; i += 64) Vg[(long long)k * n + s_rank[i]] = V[i * ld + k];
}

// ------------------------------------------------------------------------------------------
// One workgroup, ONE matrix in LDS: one-sided Jacobi on the shifted matrix (round 4).
//
// The two-sided kernel above needs A and V; for fp64 both fit the LDS only up to n = 96, above that V lives in
// L2 and a round costs 11.8 us (n = 128: 12 ms, ten times one host core).  Here B = sym(A) + sigma I with
// sigma = 1.25 min(||A||_inf, ||A||_F) >= 1.25 ||A||_2, so B is positive definite with condition <= 9, and the
// iteration runs on G = B V alone (rows of G^T in LDS, 132 KB at n = 128): a wave takes a pair of rows, three wave
// sums (|g_p|^2, |g_q|^2, g_p . g_q — DPP within 16-lane rows, v_readlane across), one rotation, rows back.  At
// convergence the columns of G are orthogonal, g_j = (lambda_j + sigma) v_j: the eigenvector is g_j / |g_j| (B is well
// conditioned: the normalisation loses nothing), the eigenvalue |g_j| - sigma — no V to accumulate, no second
// matrix, and nothing to drift.  Accuracy is absolute, eps sigma.
// ------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T eigh_sym_at(const T* A, long long n, int lower, long long i, long long j) {
  const long long r = lower ? (i > j ? i : j) : (i < j ? i : j);
  const long long c = lower ? (i > j ? j : i) : (i < j ? j : i);
  return A[r * n + c];
}

template <int CTRL>
__device__ __forceinline__ double eigh_dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float eigh_dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// the sum over the 64 lanes, in every lane (the same value, bit for bit: fixed tree)
__device__ __forceinline__ double eigh_wave_sum(double v) {
  v += eigh_dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += eigh_dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += eigh_dpp_f64<0x141>(v);  // row_half_mirror
  v += eigh_dpp_f64<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row's sum
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double t = 0.0;
#pragma unroll
  for (int r = 0; r < 4; r++) t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
  return t;
}
__device__ __forceinline__ float eigh_wave_sum(float v) {
  v += eigh_dpp_f32<0xB1>(v);
  v += eigh_dpp_f32<0x4E>(v);
  v += eigh_dpp_f32<0x141>(v);
  v += eigh_dpp_f32<0x140>(v);
  const int b = __float_as_int(v);
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 4; r++) t += __int_as_float(__builtin_amdgcn_readlane(b, 16 * r));
  return t;
}

// the sum over each 16-lane row of the wave, in every lane of that row: four DPP steps, nothing else
__device__ __forceinline__ double eigh_row_sum(double v) {
  v += eigh_dpp_f64<0xB1>(v);
  v += eigh_dpp_f64<0x4E>(v);
  v += eigh_dpp_f64<0x141>(v);
  v += eigh_dpp_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ float eigh_row_sum(float v) {
  v += eigh_dpp_f32<0xB1>(v);
  v += eigh_dpp_f32<0x4E>(v);
  v += eigh_dpp_f32<0x141>(v);
  v += eigh_dpp_f32<0x140>(v);
  return v;
}
// the sum over each 32-lane half of the wave, in every lane of that half
__device__ __forceinline__ double eigh_half_sum(double v) {
  v += eigh_dpp_f64<0xB1>(v);
  v += eigh_dpp_f64<0x4E>(v);
  v += eigh_dpp_f64<0x141>(v);
  v += eigh_dpp_f64<0x140>(v);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double h0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0)) +
                    __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
  const double h1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)) +
                    __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
  return (threadIdx.x & 32) ? h1 : h0;
}
__device__ __forceinline__ float eigh_half_sum(float v) {
  v += eigh_dpp_f32<0xB1>(v);
  v += eigh_dpp_f32<0x4E>(v);
  v += eigh_dpp_f32<0x141>(v);
  v += eigh_dpp_f32<0x140>(v);
  const int b = __float_as_int(v);
  const float h0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float h1 = __int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return (threadIdx.x & 32) ? h1 : h0;
}

constexpr int EIGH1_BLOCK = 1024;
constexpr int EIGH1_MAXN = 192;  // the kernel's per-row LDS arrays
constexpr int EIGH1_UP = 9;      // row values per lane of a 16-lane row: n <= 144 (fp64 fits the LDS up to n = 138)

template <class T>
__global__ __launch_bounds__(EIGH1_BLOCK) void eigh_onesided_kernel(const T* __restrict__ Ain, T* __restrict__ Wout,
                                                                    T* __restrict__ Vout, int n, int lower,
                                                                    int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NW = EIGH1_BLOCK / 64;
  __shared__ double s_red[2][NW];
  __shared__ T s_lam[EIGH1_MAXN], s_nrm[EIGH1_MAXN];
  __shared__ T s_n2[EIGH1_MAXN];  // |g_j|^2, carried along analytically (a' = a - t c, b' = b + t c) and recomputed every sweep
  __shared__ short s_rank[EIGH1_MAXN];
  __shared__ int s_rot;
  const long long mat = blockIdx.x;
  const int ld = n | 1;
  T* G = (T*)smem_raw;  // G^T: row j = column j of G
  const T* Ag = Ain + mat * (long long)n * n;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // ||A||_F^2 and ||A||_inf of the symmetric matrix the chosen triangle defines
  double fro = 0.0, rmax = 0.0;
  for (int i = wid; i < n; i += NW) {
    double rs = 0.0;
    for (int j = lane; j < n; j += 64) {
      const double a = (double)eigh_sym_at(Ag, (long long)n, lower, (long long)i, (long long)j);
      fro += a * a;
      rs += a < 0.0 ? -a : a;
    }
    rs = eigh_wave_sum(rs);
    rmax = rs > rmax ? rs : rmax;
  }
  fro = eigh_wave_sum(fro);
  if (lane == 0) { s_red[0][wid] = fro; s_red[1][wid] = rmax; }
  if (tid == 0) s_rot = 0;
  __syncthreads();
  fro = 0.0;
  rmax = 0.0;
  for (int w = 0; w < NW; w++) { fro += s_red[0][w]; rmax = s_red[1][w] > rmax ? s_red[1][w] : rmax; }
  double bound = sqrt(fro);
  if (rmax < bound) bound = rmax;
  double sg = 1.25 * bound;
  if (!(sg > 0.0) || sg > 1e300) sg = 1.0;  // (zero matrix; NaN / inf input: the NaNs take care of the result)
  const T sigma = (T)sg;
  for (int i = wid; i < n; i += NW)
    for (int j = lane; j < n; j += 64) G[i * ld + j] = eigh_sym_at(Ag, (long long)n, lower, (long long)i, (long long)j) + (i == j ? sigma : T(0));
  __syncthreads();
  // (measured and dropped: a Cholesky factorisation of B first, iterating on its factor — Veselic-Hari — 5.4 ms against
  //  4.9 at n = 128: B is already well conditioned, the factor's columns are no closer to orthogonal)
  const int m = (n + 1) & ~1, half = m >> 1;
  bool converged = n < 2;
  const T tol = Eps<T>::v;
  for (int sweep = 0; sweep < MAX_SWEEPS && !converged; sweep++) {
    // the squared norms afresh (the running values only steer angles and the skip test, but they should not wander)
    for (int j = wid; j < n; j += NW) {
      T a = T(0);
      for (int k = lane; k < n; k += 64) { const T g = G[j * ld + k]; a += g * g; }
      a = eigh_wave_sum(a);
      if (lane == 0) s_n2[j] = a;
    }
    __syncthreads();
    for (int r = 0; r < m - 1; r++) {
      // FOUR pairs per wave and step, one in each 16-lane row: the sum of a pair is four DPP steps inside its row (no
      // v_readlane at all), and the angle and the rotation of all four run in the same instructions — at n = 128 the 64
      // pairs of a round are ONE step of the 16 waves (a pair per wave: 4.9 ms; two: 3.4 ms).  Branch-free: a pair below
      // the threshold — or the ghost pair of an odd n — rotates by the identity.
      for (int i0p = 4 * wid; i0p < half; i0p += 4 * NW) {
        const int hsel = lane >> 4, l32 = lane & 15;
        const int i = i0p + hsel;
        int p, q;
        if (i == 0) { p = r; q = m - 1; }
        else { p = r + i; if (p >= m - 1) p -= m - 1; q = r - i; if (q < 0) q += m - 1; }
        if (p > q) { const int t = p; p = q; q = t; }
        const bool real = i < half && q < n;
        if (!real) { p = 0; q = 0; }
        T x[EIGH1_UP], y[EIGH1_UP];
        T c = T(0);
#pragma unroll
        for (int u = 0; u < EIGH1_UP; u++) {
          const int k = l32 + 16 * u;
          x[u] = (real && k < n) ? G[p * ld + k] : T(0);
          y[u] = (real && k < n) ? G[q * ld + k] : T(0);
          c += x[u] * y[u];
        }
        const T a = s_n2[p], b = s_n2[q];
        c = eigh_row_sum(c);  // the sum over this lane's 16-lane row
        const bool rotate = real && (c * c > tol * tol * a * b);
        // the rotation ANGLE in single precision (it only steers convergence: v_rcp_f32 / v_sqrt_f32 instead of three
        // fp64 divisions and a square root), the rotation itself — cs = (1 + t^2)^-1/2, sn = t cs — in working
        // precision (v_rsq_f64 + two Newton steps), so that cs^2 + sn^2 = 1 to the last bit that matters
        const float cf = rotate ? (float)c : 1.f;
        const float zf = (float)(b - a) / (2.f * cf);
        const float azf = zf < 0.f ? -zf : zf;
        float tf = 1.f / (azf + __builtin_sqrtf(zf * zf + 1.f));
        if (zf < 0.f) tf = -tf;
        T t = (T)tf;
        if (rotate && !(tf != 0.f && tf == tf)) {  // (out of single precision's range: the same formula in T)
          const T zeta = (b - a) / (T(2) * c);
          t = (zeta >= T(0) ? T(1) : T(-1)) / ((zeta < T(0) ? -zeta : zeta) + sqrt(zeta * zeta + T(1)));
        }
        if (!rotate) t = T(0);
        T cs;
        if constexpr (sizeof(T) == 8) {
          const double xx = t * t + 1.0;
          double yy = __builtin_amdgcn_rsq(xx);
          yy = yy * (1.5 - 0.5 * xx * yy * yy);
          yy = yy * (1.5 - 0.5 * xx * yy * yy);
          cs = yy;
        } else {
          cs = T(1) / __builtin_sqrtf(t * t + T(1));
        }
        const T sn = t * cs;
        if (rotate) {
#pragma unroll
          for (int u = 0; u < EIGH1_UP; u++) {
            const int k = l32 + 16 * u;
            if (k < n) {
              G[p * ld + k] = cs * x[u] - sn * y[u];
              G[q * ld + k] = sn * x[u] + cs * y[u];
            }
          }
          if (l32 == 0) { s_rot = 1; s_n2[p] = a - t * c; s_n2[q] = b + t * c; }
        }
      }
      __syncthreads();
    }
    const int rot = s_rot;
    __syncthreads();
    if (tid == 0) s_rot = 0;
    converged = rot == 0;
    __syncthreads();
  }
  if (!converged && tid == 0 && status != nullptr) atomicOr(status, 8);
  // lambda_j = |g_j| - sigma, v_j = g_j / |g_j|; ascending order by rank (ties by index), permuted write
  for (int j = wid; j < n; j += NW) {
    T a = T(0);
    for (int k = lane; k < n; k += 64) { const T g = G[j * ld + k]; a += g * g; }
    a = eigh_wave_sum(a);
    if (lane == 0) { const T nr = sqrt(a); s_nrm[j] = nr; s_lam[j] = nr - sigma; }
  }
  __syncthreads();
  T* Wg = Wout + mat * (long long)n;
  T* Vg = Vout + mat * (long long)n * n;
  for (int i = tid; i < n; i += EIGH1_BLOCK) {
    const T wi = s_lam[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
      const T wj = s_lam[j];
      rank += (wj < wi || (wj == wi && j < i)) ? 1 : 0;
    }
    if (wi != wi) rank = i;  // NaN input: every slot still gets written
    Wg[rank] = wi;
    s_rank[i] = (short)rank;
  }
  __syncthreads();
  for (int k = wid; k < n; k += NW)
    for (int i = lane; i < n; i += 64) {
      const T nr = s_nrm[i];
      Vg[(long long)k * n + s_rank[i]] = nr > T(0) ? G[i * ld + k] / nr : G[i * ld + k];
    }
}

// ------------------------------------------------------------------------------------------
// Beyond one CU (n > EIGH_BLOCK_MIN): block one-sided Jacobi over the whole chip, out of kernels that exist.
//
// B = sym(A) + sigma I with sigma = 1.25 x a rigorous bound on ||A||_2 (eigh_shift_kernel), so that every eigenvalue
// of B lies in [sigma/5, 9 sigma/5]:
// positive (a one-sided method diagonalises B^T B = B^2 and could not tell +lambda from -lambda) and
// well conditioned.  G^T (rows = columns of G = B V) and V^T start as B and I.  A round pairs the 32-column
// blocks (adjacent blocks 2k, 2k+1 — between rounds the blocks move like the players of a round-robin
// tournament, one row gather): per pair the 64 x 64 Gram matrix S = G_p^T G_p on the MFMA GEMM, its full
// eigendecomposition U by the LDS Jacobi kernel above (batched: one workgroup per pair; its accumulated rotation
// matrix UNSORTED — the block iteration only converges, quadratically, when U stays close to the identity), and
// G_p <- G_p U, V_p <- V_p U with the move to the next round's positions in one launch (eigh_pair_update_kernel;
// round 4: two more batched GEMMs and two gathers).  A sweep is nblocks-1 rounds — the first rotates all pairs of a
// subproblem, the later ones only the cross pairs of the two blocks that met (see the loop) — and the largest
// |S_ij| / sqrt(S_ii S_jj) seen during a sweep is read back once per sweep and ends the iteration.  At the end
// column j of G is (lambda_j + sigma) v_j: lambda_j = v_j . g_j - sigma, sorted ascending with their vectors.
// Rows / columns added to pad n to a whole number of block pairs carry 4 sigma on the diagonal: exactly
// decoupled, they sort to the end and are dropped.  Accuracy is absolute, eps ||A||, like LAPACK's syevd / syevr.
// Reference: Eigh.perform, pytensor/tensor/linalg/decomposition/eigen.py:177-195 (scipy.linalg.eigh, any n).
// ------------------------------------------------------------------------------------------
constexpr int EIGH_BLOCK_MIN = 160;  // up to here the one-workgroup kernel wins (n = 128: 12 ms)
constexpr int EB = 32;               // columns per block; a pair is a 64 x 64 subproblem in LDS

// max_i sum_j |a_ij| as the bits of a non-negative double (ordered like an unsigned integer)
template <class T>
__global__ __launch_bounds__(256) void eigh_rowsum_kernel(const T* __restrict__ A, long long n, int lower,
                                                          unsigned long long* __restrict__ maxbits) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  double s = 0.0;
  for (long long j = lane; j < n; j += 64) {
    const double a = (double)eigh_sym_at(A, n, lower, i, j);
    s += a < 0.0 ? -a : a;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0 && s == s) atomicMax(maxbits, (unsigned long long)__double_as_longlong(s));
}

// max_i sum_j |c_ij| of a dense N x N matrix (bits of a non-negative double)
template <class T>
__global__ __launch_bounds__(256) void eigh_absrow_kernel(const T* __restrict__ Cm, long long N, unsigned long long* __restrict__ maxbits) {
  const int lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  double s = 0.0;
  for (long long j = lane; j < N; j += 64) {
    const double a = (double)Cm[i * N + j];
    s += a < 0.0 ? -a : a;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0 && s == s) atomicMax(maxbits, (unsigned long long)__double_as_longlong(s));
}

// sigma = 1.25 min(||A||_inf, ||A^4||_inf^(1/4)) >= 1.25 ||A||_2 (both bounds are rigorous for a symmetric A; the
// second is within a small factor of the spectral norm where Gershgorin's is off by sqrt(n) for a random matrix:
// a smaller shift keeps eps * sigma — the accuracy of everything below — near eps ||A||_2 and B's eigenvalues spread
// out, which is what the iteration's early sweeps feed on).  B = A + sigma I, pads 4 sigma; G^T = B as well.
template <class T>
__global__ __launch_bounds__(256) void eigh_shift_kernel(T* __restrict__ Gt, T* __restrict__ B, long long n, long long N,
                                                         const unsigned long long* __restrict__ bits, T* __restrict__ sigma_out) {
  const double b1 = __longlong_as_double((long long)bits[0]);
  const double b4 = sqrt(sqrt(__longlong_as_double((long long)bits[2])));
  double bound = b1;
  if (b4 == b4 && b4 > 0.0 && b4 * 1.01 < bound) bound = b4 * 1.01;
  double sg = 1.25 * bound;
  if (!(sg > 0.0) || sg != sg || sg > 1e300) sg = 1.0;
  const T sigma = (T)sg;
  if (blockIdx.x == 0 && threadIdx.x == 0) sigma_out[0] = sigma;
  const long long total = N * N;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i = e / N, j = e - i * N;
    T g = Gt[e];
    if (i == j) g = i < n ? g + sigma : T(4) * sigma;
    Gt[e] = g;
    B[e] = g;
  }
}

template <class T>
__global__ __launch_bounds__(256) void eigh_prep_kernel(const T* __restrict__ A, long long n, long long N, int lower,
                                                        T* __restrict__ Gt, T* __restrict__ Vt) {
  const long long total = N * N;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long i = e / N, j = e - i * N;
    Gt[e] = (i < n && j < n) ? eigh_sym_at(A, n, lower, i, j) : T(0);  // (the shift comes later: eigh_shift_kernel)
    Vt[e] = i == j ? T(1) : T(0);
  }
}

// largest |S_ij| / sqrt(S_ii S_jj), i != j, over a batch of m x m Gram matrices (float bits, atomicMax)
template <class T>
__global__ __launch_bounds__(256) void eigh_offdiag_kernel(const T* __restrict__ S, long long batch, int m,
                                                           unsigned* __restrict__ maxbits) {
  const long long total = batch * m * m;
  float best = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / (m * m);
    const int r = (int)(e - b * m * m), i = r / m, j = r - i * m;
    if (i == j) continue;
    const T* Sb = S + b * m * m;
    const double d = (double)Sb[i * m + i] * (double)Sb[j * m + j];
    const double a = (double)Sb[r];
    if (d > 0.0) {
      const float v = (float)((a < 0.0 ? -a : a) / sqrt(d));
      best = v > best ? v : best;
    } else if (a != 0.0) {
      best = 1.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const float ov = __shfl_xor(best, o); best = ov > best ? ov : best; }
  if ((threadIdx.x & 63) == 0 && best > 0.f) atomicMax(maxbits, __float_as_uint(best));
}

// lambda_j = v_j . g_j - sigma (one wave per column of G, i.e. per row of Gt / Vt)
template <class T>
__global__ __launch_bounds__(256) void eigh_lambda_kernel(const T* __restrict__ Gt, const T* __restrict__ Vt, long long N,
                                                          const T* __restrict__ sigma, T* __restrict__ lam) {
  const int lane = threadIdx.x & 63;
  const long long j = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  T s = T(0);
  for (long long i = lane; i < N; i += 64) s += Vt[j * N + i] * Gt[j * N + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) lam[j] = s - sigma[0];
}

// ascending ranks (ties by index; NaN keeps its slot), eigenvalues of the first n ranks written out
template <class T>
__global__ __launch_bounds__(256) void eigh_rank_kernel(const T* __restrict__ lam, long long N, long long n,
                                                        int* __restrict__ rank, T* __restrict__ W) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const T wj = lam[j];
  int r = 0;
  for (long long k = 0; k < N; k++) {
    const T wk = lam[k];
    r += (wk < wj || (wk == wj && k < j)) ? 1 : 0;
  }
  if (wj != wj) r = (int)j;
  rank[j] = r;
  if (r < n) W[r] = wj;
}

// V[i][rank_j] = Vt[j][i] for the n smallest (the real) eigenpairs
template <class T>
__global__ __launch_bounds__(256) void eigh_scatter_kernel(const T* __restrict__ Vt, const int* __restrict__ rank,
                                                           long long N, long long n, T* __restrict__ V) {
  const long long j = blockIdx.y;
  const int r = rank[j];
  if (r >= n) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) V[i * n + r] = Vt[j * N + i];
}

// One round's update of the block method, both matrices in one launch: for every pair p, rows [p m, (p+1) m) of Gt and
// of Vt become U_p^T times themselves (G_p <- G_p U_p, V_p <- V_p U_p), and each new row is stored where the next
// round's pairing wants it (dst_row: the tournament step as a scatter) — what were two batched GEMMs (13 us each for
// 64 x N x 64 products) and two row gathers.  grid (N / 64 column chunks, pairs, 2); a workgroup holds U_p and a
// 64 x 64 chunk of the rows in LDS, a thread accumulates a 4 x 4 piece of the result over the 64 terms in index order.
template <class T>
__global__ __launch_bounds__(256) void eigh_pair_update_kernel(const T* __restrict__ U, const T* __restrict__ Gt, const T* __restrict__ Vt,
                                                               T* __restrict__ Gout, T* __restrict__ Vout, long long N,
                                                               const long long* __restrict__ dst_row) {
  constexpr int M = 2 * EB;  // 64
  __shared__ __attribute__((aligned(16))) T Us[M][M];  // U[i][i']
  __shared__ __attribute__((aligned(16))) T Xs[M][M];  // X[i][c]
  const int tid = threadIdx.x;
  const long long c0 = (long long)blockIdx.x * M, p = blockIdx.y;
  const T* X = (blockIdx.z ? Vt : Gt) + p * M * N + c0;
  T* Y = blockIdx.z ? Vout : Gout;
  const T* Up = U + p * M * M;
  for (int e = tid; e < M * M; e += 256) {
    const int i = e / M, j = e - i * M;
    Us[i][j] = Up[e];
    Xs[i][j] = X[(long long)i * N + j];
  }
  __syncthreads();
  const int r0 = (tid >> 4) * 4, q0 = (tid & 15) * 4;
  T acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = T(0);
#pragma unroll 8
  for (int i = 0; i < M; i++) {
    T u[4], x[4];
#pragma unroll
    for (int a = 0; a < 4; a++) { u[a] = Us[i][r0 + a]; x[a] = Xs[i][q0 + a]; }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] += u[a] * x[b];
  }
#pragma unroll
  for (int a = 0; a < 4; a++) {
    T* yr = Y + dst_row[p * M + r0 + a] * N + c0 + q0;
#pragma unroll
    for (int b = 0; b < 4; b++) yr[b] = acc[a][b];
  }
}

__global__ void eigh_status_or_kernel(int* status, int bits) { atomicOr(status, bits); }

template <class T>
int eigh_typed(long long batch, long long n, int lower, const void* A, void* W, void* V, int sorted = 1, int max_sweeps = MAX_SWEEPS,
               int cross = 0);

template <class T>
int eigh_block_jacobi(long long n, int lower, const T* A, T* W, T* V) {
  hipStream_t st = pthip::ctx().stream;
  const int dt = sizeof(T) == 8 ? PTHIP_F64 : PTHIP_F32;
  const long long m = 2 * EB;
  const long long npairs = (n + m - 1) / m, N = npairs * m, nbk = 2 * npairs;
  const size_t mat = (size_t)N * N * sizeof(T);
  const size_t sbytes = (size_t)npairs * m * m * sizeof(T);
  const size_t ibytes = ((size_t)N * 8 + 255) / 256 * 256;
  const size_t lbytes = ((size_t)N * sizeof(T) + 255) / 256 * 256;
  void* ws = nullptr;
  int r = pthip_alloc(5 * mat + 2 * sbytes + (size_t)npairs * m * sizeof(T) + 256 + 3 * ibytes + lbytes + 256, &ws);
  if (r) return r;
  auto fail = [&](int rc) { pthip_free(ws); return rc; };
  char* p = (char*)ws;
  T* Gt = (T*)p; p += mat;
  T* G2 = (T*)p; p += mat;
  T* Vt = (T*)p; p += mat;
  T* V2 = (T*)p; p += mat;
  T* B = (T*)p; p += mat;
  T* S = (T*)p; p += sbytes;
  T* U = (T*)p; p += sbytes;
  T* Wsub = (T*)p; p += ((size_t)npairs * m * sizeof(T) + 255) / 256 * 256;
  long long* idx = (long long*)p; p += ibytes;
  long long* dst = (long long*)p; p += ibytes;  // the same step as a scatter: row r of a round's result goes to row dst[r]
  int* rank = (int*)p; p += ibytes;
  T* lam = (T*)p; p += lbytes;
  unsigned long long* maxbits = (unsigned long long*)p;  // [0] ||A||_inf bits, [1] low word: off-diagonal bits, [2] ||A^4||_inf bits
  unsigned* offbits = (unsigned*)(maxbits + 1);
  T* sigma = (T*)(maxbits + 3);
  // the tournament step on block positions (pairs are the adjacent positions 2k, 2k+1): position 0 stays, the top
  // row shifts right, the bottom row shifts left; as a row gather: new row r of block position q comes from ...
  {
    std::vector<long long> src(nbk), h(N);
    const long long hp = npairs;
    for (long long k = 0; k < hp; k++) {
      // top_k = position 2k, bottom_k = position 2k+1
      long long top_src, bot_src;
      if (k == 0) top_src = 0;
      else if (k == 1) top_src = 1;                 // bottom_0
      else top_src = 2 * (k - 1);                    // top_{k-1}
      if (k == hp - 1) bot_src = hp == 1 ? 1 : 2 * (hp - 1);  // top_{hp-1}
      else bot_src = 2 * (k + 1) + 1;                // bottom_{k+1}
      src[2 * k] = top_src;
      src[2 * k + 1] = bot_src;
    }
    for (long long q = 0; q < nbk; q++)
      for (long long t = 0; t < EB; t++) h[q * EB + t] = src[q] * EB + t;
    if (hipError_t e = hipMemcpyAsync(idx, h.data(), (size_t)N * 8, hipMemcpyHostToDevice, st); e != hipSuccess) return fail(pthip::check(e, "eigh idx upload"));
    std::vector<long long> inv(N);
    for (long long r_ = 0; r_ < N; r_++) inv[nbk > 2 ? h[r_] : r_] = r_;  // (one pair: nothing moves)
    if (hipError_t e = hipMemcpyAsync(dst, inv.data(), (size_t)N * 8, hipMemcpyHostToDevice, st); e != hipSuccess) return fail(pthip::check(e, "eigh dst upload"));
    if (hipError_t e = hipStreamSynchronize(st); e != hipSuccess) return fail(pthip::check(e, "eigh idx sync"));  // (h goes out of scope)
  }
  if (hipError_t e = hipMemsetAsync(maxbits, 0, 32, st); e != hipSuccess) return fail(pthip::check(e, "eigh memset"));
  PTHIP_KLAUNCH((eigh_rowsum_kernel<T>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, A, n, lower, maxbits);
  const unsigned pg = (unsigned)(N * N / 256 > 4096 ? 4096 : (N * N + 255) / 256);
  PTHIP_KLAUNCH((eigh_prep_kernel<T>), dim3(pg), dim3(256), 0, st, A, n, N, lower, Gt, Vt);
  // ||A||_2^4 = ||A^4||_2 <= ||A^4||_inf: two squarings on the MFMA GEMM
  if ((r = pthip_gemm(dt, 1, N, N, N, 1.0, Gt, 0, N, 1, Gt, 0, N, 1, 0.0, nullptr, 0, 0, 0, G2))) return fail(r);
  if ((r = pthip_gemm(dt, 1, N, N, N, 1.0, G2, 0, N, 1, G2, 0, N, 1, 0.0, nullptr, 0, 0, 0, V2))) return fail(r);
  PTHIP_KLAUNCH((eigh_absrow_kernel<T>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, (const T*)V2, N, maxbits + 2);
  PTHIP_KLAUNCH((eigh_shift_kernel<T>), dim3(pg), dim3(256), 0, st, Gt, B, n, N, (const unsigned long long*)maxbits, sigma);
  if ((r = pthip::post_launch("eigh_prep"))) return fail(r);
  const double eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920929e-07;
  const double tol = eps * (N > 256 ? (double)N / 4 : 64.0);
  double prev = 1e300;
  bool converged = false;
  static const int inner_env = getenv("PTHIP_EIGH_INNER") ? atoi(getenv("PTHIP_EIGH_INNER")) : 0;
  // one inner sweep per subproblem: n = 1024 in 131 ms (13 outer sweeps) against 207 / 277 / 384 ms with 2 / 3 / until
  // converged (12 outer sweeps each) — profiles/r4n_eigh.txt
  const int inner_cap = inner_env > 0 ? inner_env : 1;
  const int max_sweeps = 24;  // (a 260-fold eigenvalue — rank 40 at n = 300 — takes 16: convergence inside a cluster is linear)
  for (int sweep = 0; sweep < max_sweeps && !converged; sweep++) {
    const long long rounds = nbk > 2 ? nbk - 1 : 1;
    // G = B V afresh at the start of a sweep: G and V take the same rotations, but their rounding errors are their
    // own — over hundreds of rounds G drifts away from B V (measured: eigenvalues off by 1e2 eps sigma without this)
    if (sweep > 0)
      if ((r = pthip_gemm(dt, 1, N, N, N, 1.0, Vt, 0, N, 1, B, 0, N, 1, 0.0, nullptr, 0, 0, 0, Gt))) return fail(r);
    for (long long rd = 0; rd < rounds; rd++) {
      // Gram matrices of the pairs: S_p = G_p^T G_p with G_p^T = rows [p m, (p+1) m) of Gt
      if ((r = pthip_gemm(dt, npairs, m, m, N, 1.0, Gt, m * N, N, 1, Gt, m * N, 1, N, 0.0, nullptr, 0, 0, 0, S))) return fail(r);
      PTHIP_KLAUNCH((eigh_offdiag_kernel<T>), dim3((unsigned)(npairs * m * m / 256)), dim3(256), 0, st, (const T*)S, npairs, (int)m, offbits);
      // (inner sweeps are capped: a subproblem need not be diagonalised to the last bit while the pairs around it
      //  are still far from orthogonal; PTHIP_EIGH_INNER overrides)
      // A sweep of the block method should visit every pair of columns once.  The pairs INSIDE a block meet in every
      // round of the sweep (the block travels as one): they are rotated in the sweep's first round only (a full
      // round-robin sweep of the 2 EB x 2 EB subproblem, 2 EB - 1 rounds); every later round rotates just the
      // EB x EB cross pairs of the two blocks that have met (EB rounds) — half the dependent rounds of a subproblem,
      // the standard cyclic-by-blocks order.  Only while the iteration is far from done (the last sweep's largest
      // relative off-diagonal above 1e-3 — 7 of 10 sweeps at n = 256, where most of the time goes): the end game keeps
      // the full sweep in every round, which is what pulls a multiple eigenvalue's vectors apart (rank 40 at n = 300:
      // 16 sweeps with it, not converged after 16 without; profiles/r6e_eigh_cross.txt).  PTHIP_EIGH_CROSS=0: the
      // full sweep in every round (round 4).
      static const bool cross_ok = !(getenv("PTHIP_EIGH_CROSS") && atoi(getenv("PTHIP_EIGH_CROSS")) == 0);
      const int cross = (cross_ok && rd > 0 && prev > 1e-3) ? 1 : 0;
      if ((r = eigh_typed<T>(npairs, m, 1, S, Wsub, U, /*sorted=*/0, inner_cap, cross))) return fail(r);
      // G_p <- G_p U  <=>  rows: Gt_p <- U^T Gt_p ; the same for V
      static const bool fused_update = !(getenv("PTHIP_EIGH_FUSED_UPDATE") && atoi(getenv("PTHIP_EIGH_FUSED_UPDATE")) == 0);
      if (fused_update) {
        PTHIP_KLAUNCH((eigh_pair_update_kernel<T>), dim3((unsigned)(N / m), (unsigned)npairs, 2u), dim3(256), 0, st, (const T*)U, (const T*)Gt,
                      (const T*)Vt, G2, V2, N, (const long long*)dst);
        std::swap(Gt, G2);
        std::swap(Vt, V2);
        continue;
      }
      if ((r = pthip_gemm(dt, npairs, m, N, m, 1.0, U, m * m, 1, m, Gt, m * N, N, 1, 0.0, nullptr, 0, 0, 0, G2))) return fail(r);
      if ((r = pthip_gemm(dt, npairs, m, N, m, 1.0, U, m * m, 1, m, Vt, m * N, N, 1, 0.0, nullptr, 0, 0, 0, V2))) return fail(r);
      if (nbk > 2) {
        if ((r = pthip_take_rows((int)sizeof(T), N, N, G2, N, N, (const int64_t*)idx, Gt))) return fail(r);
        if ((r = pthip_take_rows((int)sizeof(T), N, N, V2, N, N, (const int64_t*)idx, Vt))) return fail(r);
      } else {
        std::swap(Gt, G2);
        std::swap(Vt, V2);
      }
    }
    unsigned hb = 0;
    if (hipError_t e = hipMemcpyAsync(&hb, offbits, 4, hipMemcpyDeviceToHost, st); e != hipSuccess) return fail(pthip::check(e, "eigh off-diagonal read"));
    if (hipError_t e = hipStreamSynchronize(st); e != hipSuccess) return fail(pthip::check(e, "eigh sweep sync"));
    if (hipError_t e = hipMemsetAsync(offbits, 0, 4, st); e != hipSuccess) return fail(pthip::check(e, "eigh memset"));
    float offf;
    memcpy(&offf, &hb, 4);
    const double off = offf;
    static const bool trace = getenv("PTHIP_EIGH_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[pthip eigh] n=%lld sweep %d: largest relative off-diagonal of the pair Gram matrices %.3e (tol %.1e)\n", n, sweep, off, tol);
    converged = off <= tol || (sweep >= 2 && off <= 100 * tol && off > 0.5 * prev);
    prev = off;
  }
  if (!converged) {
    // (scipy raises LinAlgError when LAPACK reports non-convergence: bit 3 of the device error word)
    PTHIP_KLAUNCH(eigh_status_or_kernel, dim3(1), dim3(1), 0, st, (int*)pthip_status_ptr(), 8);
  }
  if ((r = pthip_gemm(dt, 1, N, N, N, 1.0, Vt, 0, N, 1, B, 0, N, 1, 0.0, nullptr, 0, 0, 0, Gt))) return fail(r);  // lambda from a fresh B V
  PTHIP_KLAUNCH((eigh_lambda_kernel<T>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, (const T*)Gt, (const T*)Vt, N, (const T*)sigma, lam);
  PTHIP_KLAUNCH((eigh_rank_kernel<T>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, (const T*)lam, N, n, rank, W);
  PTHIP_KLAUNCH((eigh_scatter_kernel<T>), dim3((unsigned)((n + 255) / 256), (unsigned)N), dim3(256), 0, st, (const T*)Vt, (const int*)rank, N, n, V);
  r = pthip::post_launch("eigh_block_finish");
  pthip_free(ws);
  return r;
}

template <class T>
int eigh_typed(long long batch, long long n, int lower, const void* A, void* W, void* V, int sorted, int max_sweeps, int cross) {
  if (batch == 0 || n == 0) return 0;
  static const bool no_block = getenv("PTHIP_EIGH_ONE_WG") != nullptr;
  if (n > EIGH_BLOCK_MIN && !(no_block && n <= MAX_N)) {
    for (long long b = 0; b < batch; b++) {
      int r = eigh_block_jacobi<T>(n, lower, (const T*)A + b * n * n, (T*)W + b * n, (T*)V + b * n * n);
      if (r) return r;
    }
    return 0;
  }
  hipStream_t st = pthip::ctx().stream;
  const size_t one = (size_t)n * (size_t)(n | 1) * sizeof(T);
  // the one-matrix kernel whenever a caller wants sorted eigenpairs (not the rotation matrix of a block-method
  // subproblem) and G fits the LDS; PTHIP_EIGH_TWO_SIDED=1 keeps the round-3 kernel (A/B reference)
  static const bool two_sided = getenv("PTHIP_EIGH_TWO_SIDED") != nullptr;
  // (fp64 only: its accuracy is absolute, eps sigma with sigma up to ~5 ||A||_2 — inside north_star's 1e-12 in fp64,
  //  outside its 1e-5 for the small eigenvalues in fp32, where A and V both fit the LDS up to n = 136 anyway)
  if (sizeof(T) == 8 && sorted && !two_sided && n >= 8 && n <= EIGH1_MAXN && n <= 16 * EIGH1_UP && one <= 160 * 1024 - 8 * 1024) {
    auto k1 = eigh_onesided_kernel<T>;
    static bool attr1 = false;
    if (!attr1) {
      PTHIP_CHECK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
      attr1 = true;
    }
    PTHIP_KLAUNCH(k1, dim3((unsigned)batch), dim3(EIGH1_BLOCK), one, st, (const T*)A, (T*)W, (T*)V, (int)n, lower, (int*)pthip_status_ptr());
    return pthip::post_launch("eigh(one-sided)");
  }
  const size_t budget = 160 * 1024 - 12 * 1024;  // static scratch of the kernel + slack
  const bool a_lds = one <= budget;
  const bool v_lds = a_lds && 2 * one <= budget;
  const size_t dyn = (a_lds ? one : 0) + (v_lds ? one : 0);
  // (the subproblems of the block method: a handful of matrices, one per CU, every round a chain of LDS round trips —
  //  16 waves shorten it; PTHIP_EIGH_INNER_WIDE=0 keeps 4)
  static const bool inner_wide = !(getenv("PTHIP_EIGH_INNER_WIDE") && atoi(getenv("PTHIP_EIGH_INNER_WIDE")) == 0);
  const bool wide = n > 64 || (inner_wide && !sorted && batch <= 256);
  auto k = wide ? eigh_jacobi_kernel<T, 1024> : eigh_jacobi_kernel<T, 256>;
  if (dyn > 48 * 1024)
    PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  void *sa = nullptr, *sv = nullptr;
  if (!a_lds) { int r = pthip_alloc((size_t)batch * one, &sa); if (r) return r; }
  if (!v_lds) { int r = pthip_alloc((size_t)batch * one, &sv); if (r) { if (sa) pthip_free(sa); return r; } }
  PTHIP_KLAUNCH(k, dim3((unsigned)batch), dim3(wide ? 1024 : 256), dyn, st, (const T*)A, (T*)W, (T*)V, (int)n, lower,
                     a_lds ? 1 : 0, v_lds ? 1 : 0, (T*)sa, (T*)sv, (int*)pthip_status_ptr(), sorted, max_sweeps, (cross && n % 2 == 0) ? 1 : 0);
  int r = pthip::post_launch("eigh");
  if (sa) pthip_free(sa);  // stream-ordered reuse keeps this safe
  if (sv) pthip_free(sv);
  return r;
}

}  // namespace

// out = the symmetric matrix whose `lower` (or upper) triangle is A's: what LAPACK's sy* routines
// read.  The generalised problem A v = w B v (Eigh with two inputs, eigen.py:177-186) is reduced on
// the full matrices (Cholesky of B, two triangular solves), so the unread triangle must not leak in.
namespace {
template <class T>
__global__ __launch_bounds__(256) void symmetrize_kernel(T* __restrict__ out, const T* __restrict__ A, long long n,
                                                         long long total, int lower) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long b = e / (n * n), r = e - b * n * n;
    const long long i = r / n, j = r - i * n;
    const bool take = lower ? (i >= j) : (i <= j);
    out[e] = take ? A[e] : A[b * n * n + j * n + i];
  }
}
}  // namespace

extern "C" int pthip_symmetrize(int dtype, int64_t batch, int64_t n, int lower, const void* A, void* out) {
  PTHIP_REQUIRE_INIT();
  const long long total = (long long)batch * n * n;
  if (total == 0) return 0;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = pthip::ctx().stream;
  if (dtype == PTHIP_F64)
    PTHIP_KLAUNCH((symmetrize_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, (double*)out, (const double*)A, (long long)n, total, lower);
  else if (dtype == PTHIP_F32)
    PTHIP_KLAUNCH((symmetrize_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (float*)out, (const float*)A, (long long)n, total, lower);
  else
    return pthip::set_error("pthip_symmetrize: dtype %d not supported (float32/float64 only)", dtype);
  return pthip::post_launch("symmetrize");
}

extern "C" int pthip_eigh(int dtype, int64_t batch, int64_t n, int lower, const void* A, void* W, void* V) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64) return eigh_typed<double>(batch, n, lower, A, W, V);
  if (dtype == PTHIP_F32) return eigh_typed<float>(batch, n, lower, A, W, V);
  return pthip::set_error("pthip_eigh: dtype %d not supported (float32/float64 only)", dtype);
}
