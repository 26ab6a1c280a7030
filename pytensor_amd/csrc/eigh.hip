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

namespace {

constexpr int MAX_SWEEPS = 60;
constexpr int MAX_N = 512;

template <class T> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; };

// BLOCK: 256 for small matrices (several workgroups per CU when batched), 1024 above n = 64 — a round
// has n/2 independent pairs and every pair is a chain of LDS / L2 round trips, so waves are what
// shortens it (n = 128: 15.2 -> 12.0 ms with 16 waves instead of 4)
template <class T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void eigh_jacobi_kernel(const T* __restrict__ Ain, T* __restrict__ Wout,
                                                           T* __restrict__ Vout, int n, int lower, int a_lds,
                                                           int v_lds, T* scratchA, T* scratchV,
                                                           int* __restrict__ status) {
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
  for (int sweep = 0; sweep < MAX_SWEEPS && !converged; sweep++) {
    for (int r = 0; r < m - 1; r++) {
      // ---- phase 1: one thread per pair computes its rotation ----
      for (int i = tid; i < half; i += BLOCK) {
        int p, q;
        if (i == 0) { p = r; q = m - 1; }
        else { p = r + i; if (p >= m - 1) p -= m - 1; q = r - i; if (q < 0) q += m - 1; }
        if (p > q) { const int t = p; p = q; q = t; }
        T c = T(1), s = T(0);
        if (q < n) {
          const T app = A[p * ld + p], aqq = A[q * ld + q], apq = A[p * ld + q];
          const T mag = apq < T(0) ? -apq : apq;
          const T dd = app * aqq;
          if (mag > tiny_abs && mag > Eps<T>::v * sqrt(dd < T(0) ? -dd : dd)) {
            const T theta = (aqq - app) / (T(2) * apq);
            const T at = theta < T(0) ? -theta : theta;
            T t = T(1) / (at + sqrt(theta * theta + T(1)));
            if (theta < T(0)) t = -t;
            c = T(1) / sqrt(t * t + T(1));
            s = t * c;
            s_rot = 1;
          }
        }
        s_c[i] = c; s_s[i] = s; s_p[i] = (short)p; s_q[i] = (short)q;
      }
      __syncthreads();
      // ---- phase 2: columns p, q of A and of V (lanes down the rows; each wave works on two
      //      pairs at a time so that their LDS round trips overlap) ----
      constexpr int NW = BLOCK / 64;
      for (int i0 = wid; i0 < half; i0 += 2 * NW) {
        const int i1 = i0 + NW < half ? i0 + NW : i0;  // (odd tail: the second slot idles)
        const bool two = i1 != i0;
        const T s0 = s_s[i0], c0 = s_c[i0], s1 = two ? s_s[i1] : T(0), c1 = two ? s_c[i1] : T(1);
        if (s0 == T(0) && s1 == T(0)) continue;
        const int p0 = s_p[i0], q0 = s_q[i0], p1 = s_p[i1], q1 = s_q[i1];
        for (int k = lane; k < n; k += 64) {
          const T x0 = A[k * ld + p0], y0 = A[k * ld + q0], x1 = A[k * ld + p1], y1 = A[k * ld + q1];
          // (V is kept transposed: eigenvector p is row p — contiguous in k, which is what the
          //  global-scratch case needs to stay coalesced)
          const T vx0 = V[p0 * ld + k], vy0 = V[q0 * ld + k], vx1 = V[p1 * ld + k], vy1 = V[q1 * ld + k];
          if (s0 != T(0)) {
            A[k * ld + p0] = c0 * x0 - s0 * y0;
            A[k * ld + q0] = s0 * x0 + c0 * y0;
            V[p0 * ld + k] = c0 * vx0 - s0 * vy0;
            V[q0 * ld + k] = s0 * vx0 + c0 * vy0;
          }
          if (s1 != T(0)) {
            A[k * ld + p1] = c1 * x1 - s1 * y1;
            A[k * ld + q1] = s1 * x1 + c1 * y1;
            V[p1 * ld + k] = c1 * vx1 - s1 * vy1;
            V[q1 * ld + k] = s1 * vx1 + c1 * vy1;
          }
        }
      }
      __syncthreads();
      // ---- phase 3: rows p, q of A; the annihilated pair is stored as an exact zero ----
      for (int i0 = wid; i0 < half; i0 += 2 * NW) {
        const int i1 = i0 + NW < half ? i0 + NW : i0;
        const bool two = i1 != i0;
        const T s0 = s_s[i0], c0 = s_c[i0], s1 = two ? s_s[i1] : T(0), c1 = two ? s_c[i1] : T(1);
        if (s0 == T(0) && s1 == T(0)) continue;
        const int p0 = s_p[i0], q0 = s_q[i0], p1 = s_p[i1], q1 = s_q[i1];
        for (int k = lane; k < n; k += 64) {
          const T x0 = A[p0 * ld + k], y0 = A[q0 * ld + k], x1 = A[p1 * ld + k], y1 = A[q1 * ld + k];
          if (s0 != T(0)) {
            A[p0 * ld + k] = k == q0 ? T(0) : c0 * x0 - s0 * y0;
            A[q0 * ld + k] = k == p0 ? T(0) : s0 * x0 + c0 * y0;
          }
          if (s1 != T(0)) {
            A[p1 * ld + k] = k == q1 ? T(0) : c1 * x1 - s1 * y1;
            A[q1 * ld + k] = k == p1 ? T(0) : s1 * x1 + c1 * y1;
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
  if (!converged && tid == 0 && status != nullptr) atomicOr(status, 8);  // bit 3: LinAlgError("Eigenvalues did not converge")

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
    Wg[rank] = wi;
    s_rank[i] = (short)rank;
  }
  __syncthreads();
  for (int k = wid; k < n; k += BLOCK / 64)
    for (int i = lane; i < n; i += 64) Vg[(long long)k * n + s_rank[i]] = V[i * ld + k];
}

template <class T>
int eigh_typed(long long batch, long long n, int lower, const void* A, void* W, void* V) {
  if (batch == 0 || n == 0) return 0;
  if (n > MAX_N) return pthip::set_error("pthip_eigh: n = %lld > %d is not supported yet", n, MAX_N);
  hipStream_t st = pthip::ctx().stream;
  const size_t one = (size_t)n * (size_t)(n | 1) * sizeof(T);
  const size_t budget = 160 * 1024 - 12 * 1024;  // static scratch of the kernel + slack
  const bool a_lds = one <= budget;
  const bool v_lds = a_lds && 2 * one <= budget;
  const size_t dyn = (a_lds ? one : 0) + (v_lds ? one : 0);
  const bool wide = n > 64;
  auto k = wide ? eigh_jacobi_kernel<T, 1024> : eigh_jacobi_kernel<T, 256>;
  if (dyn > 48 * 1024)
    PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  void *sa = nullptr, *sv = nullptr;
  if (!a_lds) { int r = pthip_alloc((size_t)batch * one, &sa); if (r) return r; }
  if (!v_lds) { int r = pthip_alloc((size_t)batch * one, &sv); if (r) { if (sa) pthip_free(sa); return r; } }
  PTHIP_KLAUNCH(k, dim3((unsigned)batch), dim3(wide ? 1024 : 256), dyn, st, (const T*)A, (T*)W, (T*)V, (int)n, lower,
                     a_lds ? 1 : 0, v_lds ? 1 : 0, (T*)sa, (T*)sv, (int*)pthip_status_ptr());
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
