// lu.hip — LU factorisation with partial pivoting (getrf) for the general dense solves.
//
// Reference: Solve.perform (pytensor/tensor/linalg/solvers/general.py: scipy.linalg.solve,
// assume_a="gen" -> LAPACK gesv = getrf + getrs), Det / SLogDet (pytensor/tensor/linalg/
// summary.py: np.linalg.det / slogdet = getrf + the diagonal), MatrixInverse (np.linalg.inv).
// An exactly singular matrix: Solve NaN-fills (general.py:74-75), det is 0, slogdet (0, -inf);
// only np.linalg.inv raises LinAlgError — with flag_singular the kernel raises bit 1 of the
// device error word and the executor raises that exception at its next synchronisation.
//
// correct first.  One workgroup per matrix (batches on grid.x),
// the matrix resident in LDS when it fits (n <= ~140 fp64), unblocked right-looking
// elimination: per column one block-wide arg-max (first maximum, like idamax), a row swap,
// a scale by the reciprocal pivot (dgetf2 does the same above sfmin) and a rank-1 update.
// Outputs: the packed factors, the row permutation as a gather vector (row i of P*A is row
// perm[i] of A), its sign (0 when singular) and log|det|.
#include "common.h"

namespace {

constexpr int BLOCK = 256;

template <class T> __device__ __forceinline__ T dev_abs(T x) { return x < T(0) ? -x : x; }

template <class T>
__global__ __launch_bounds__(BLOCK) void getrf_kernel(T* __restrict__ LUout,
                                                     const T* __restrict__ Ain,
                                                     long long* __restrict__ perm_out,
                                                     T* __restrict__ sign_out,
                                                     T* __restrict__ logabs_out, int n, int use_lds,
                                                     T* __restrict__ scratch, int* __restrict__ status) {  // status == nullptr: singular is not an error
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ T s_val[BLOCK / 64];
  __shared__ int s_idx[BLOCK / 64];
  __shared__ int s_perm[512];
  const long long mat = blockIdx.x;
  const T* A = Ain + mat * (long long)n * n;
  T* Lo = LUout + mat * (long long)n * n;
  const int ld = use_lds ? (n | 1) : n;
  T* W = use_lds ? (T*)smem_raw : scratch + mat * (long long)n * n;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int e = tid; e < n * n; e += BLOCK) {
    const int i = e / n, j = e - i * n;
    W[i * ld + j] = A[e];
  }
  for (int i = tid; i < n; i += BLOCK) s_perm[i] = i;
  __syncthreads();
  T sign = T(1);
  bool singular = false;
  for (int k = 0; k < n; k++) {
    // ---- pivot: first row with the largest |w_ik|, i >= k ----
    T best = T(-1);
    int bi = k;
    for (int i = k + tid; i < n; i += BLOCK) {
      const T v = dev_abs(W[i * ld + k]);
      if (v > best) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const T ov = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    // (every thread folds the per-wave candidates itself: three barriers per column, not five;
    //  the slots are rewritten only after the two barriers below)
    if (lane == 0) { s_val[wid] = best; s_idx[wid] = bi; }
    __syncthreads();
    int p = s_idx[0];
    {
      T bv = s_val[0];
      for (int w = 1; w < BLOCK / 64; w++)
        if (s_val[w] > bv || (s_val[w] == bv && s_idx[w] < p)) { bv = s_val[w]; p = s_idx[w]; }
    }
    if (p != k) {
      for (int j = tid; j < n; j += BLOCK) {
        const T t = W[k * ld + j];
        W[k * ld + j] = W[p * ld + j];
        W[p * ld + j] = t;
      }
      if (tid == 0) { const int t = s_perm[k]; s_perm[k] = s_perm[p]; s_perm[p] = t; }
      sign = -sign;
    }
    __syncthreads();
    const T piv = W[k * ld + k];
    if (piv == T(0)) singular = true;  // dgetf2: info = k+1, no scaling, elimination continues
    const T rp = piv == T(0) ? T(1) : T(1) / piv;
    // ---- scale column k and rank-1 update of the trailing block, one phase ----
    // (16 x 16 thread tiles: no index division; the threads of tile column 0 store the
    //  multipliers, which only their own tile row reads)
    for (int i = k + 1 + (tid >> 4); i < n; i += BLOCK / 16) {
      const T lik = W[i * ld + k] * rp;
      for (int j = k + 1 + (tid & 15); j < n; j += 16) W[i * ld + j] -= lik * W[k * ld + j];
      if ((tid & 15) == 0) W[i * ld + k] = lik;
    }
    __syncthreads();
  }
  for (int e = tid; e < n * n; e += BLOCK) {
    const int i = e / n, j = e - i * n;
    Lo[e] = W[i * ld + j];
  }
  for (int i = tid; i < n; i += BLOCK) perm_out[mat * n + i] = s_perm[i];
  if (tid == 0) {
    // slogdet the way umath_linalg does it: sign and log|det| accumulated along the diagonal
    T la = T(0);
    T sg = sign;
    for (int i = 0; i < n; i++) {
      const T d = W[i * ld + i];
      if (d < T(0)) sg = -sg;
      la += log(dev_abs(d));
    }
    if (singular) { sg = T(0); la = -__builtin_huge_val(); }
    sign_out[mat] = sg;
    logabs_out[mat] = la;
    if (singular && status != nullptr) atomicOr(status, 2);
  }
}

template <class T>
int getrf_typed(long long batch, long long n, const void* A, void* LU, void* perm, void* sign,
                void* logabs, int flag_singular) {
  if (batch == 0) return 0;
  if (n > 512) return pthip::set_error("pthip_getrf: n = %lld > 512 is not supported yet", n);
  hipStream_t st = pthip::ctx().stream;
  const size_t need = (size_t)n * (size_t)(n | 1) * sizeof(T);
  const bool lds = need <= 160 * 1024 - 8192;
  auto k = getrf_kernel<T>;
  void* scratch = nullptr;
  if (lds) {
    if (need > 48 * 1024)
      PTHIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
  } else {
    int r = pthip_alloc((size_t)batch * n * n * sizeof(T), &scratch);
    if (r) return r;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)batch), dim3(BLOCK), lds ? need : 0, st, (T*)LU, (const T*)A,
                     (long long*)perm, (T*)sign, (T*)logabs, (int)n, lds ? 1 : 0, (T*)scratch,
                     flag_singular ? (int*)pthip_status_ptr() : (int*)nullptr);
  int r = pthip::post_launch("getrf");
  if (scratch) pthip_free(scratch);  // stream-ordered reuse keeps this safe
  return r;
}

}  // namespace

extern "C" int pthip_getrf(int dtype, int64_t batch, int64_t n, const void* A, void* LU, void* perm,
                           void* sign, void* logabsdet, int flag_singular) {
  PTHIP_REQUIRE_INIT();
  if (n == 0) return 0;
  if (dtype == PTHIP_F64) return getrf_typed<double>(batch, n, A, LU, perm, sign, logabsdet, flag_singular);
  if (dtype == PTHIP_F32) return getrf_typed<float>(batch, n, A, LU, perm, sign, logabsdet, flag_singular);
  return pthip::set_error("pthip_getrf: dtype %d not supported (float32/float64 only)", dtype);
}
