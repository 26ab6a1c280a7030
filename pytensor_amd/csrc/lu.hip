// lu.hip — LU factorisation with partial pivoting (getrf) for the general dense solves.
//
// Reference: Solve.perform (pytensor/tensor/linalg/solvers/general.py: scipy.linalg.solve,
// assume_a="gen" -> LAPACK gesv = getrf + getrs), Det / SLogDet (pytensor/tensor/linalg/
// summary.py: np.linalg.det / slogdet = getrf + the diagonal), MatrixInverse (np.linalg.inv).
// An exactly singular matrix: Solve NaN-fills (general.py:74-75), det is 0, slogdet (0, -inf);
// only np.linalg.inv raises LinAlgError — with flag_singular the kernel raises bit 1 of the
// device error word and the executor raises that exception at its next synchronisation.
//
// SURVEY §8f row 3 (widening).  One workgroup per matrix (batches on grid.x).  Up to n = 128
// the matrix lives in registers and pivoting is implicit (getrf_reg_kernel below: two barriers
// per column); larger matrices are swept in LDS (n <= ~140 fp64) or in an L2-resident global
// scratch (getrf_kernel: unblocked right-looking elimination, per column one block-wide
// arg-max — first maximum, like idamax —, a row swap, a scale by the reciprocal pivot as dgetf2
// does above sfmin, and a rank-1 update).
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

// ---- n <= 128: the matrix lives in registers, pivoting is implicit --------------------------
// 256 threads as a 16 x 16 grid, thread (ty, tx) owns the NB x NB elements (ty + 16 r, tx + 16 c).
// Rows are never moved: a pivoted row just stops taking part (a bit of `active`); its position in
// P*A is tracked in LDS (s_pos / s_cur, updated exactly like LAPACK's explicit swaps would move
// it, so that ties in the pivot search resolve to the row LAPACK would pick: the first in the
// *current* order) and applied once, when the factors are written out.  Per column: every wave
// finds the pivot itself from the column copy in LDS (DPP arg-max, no cross-wave step), the
// owners of the pivot row publish it, one barrier, a rank-1 update out of registers (the
// multipliers go to a staging matrix in global memory, so the update needs no per-element
// tests), the owners of the next column publish that, one barrier.  Two barriers and ~3 LDS
// reads per thread-row and column instead of a read-modify-write sweep of the trailing block.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_own(int own) {
  return __builtin_amdgcn_update_dpp(own, own, CTRL, ROW_MASK, 0xf, false);
}

// wave-wide arg-max of (v, key): larger v wins, ties go to the smaller key; result in lane 63
template <class T, int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmax_step(T& v, int& key) {
  T ov;
  if constexpr (sizeof(T) == 8) {
    const int lo = dpp_or_own<CTRL, ROW_MASK>(__double2loint(v));
    const int hi = dpp_or_own<CTRL, ROW_MASK>(__double2hiint(v));
    ov = __hiloint2double(hi, lo);
  } else {
    ov = __int_as_float(dpp_or_own<CTRL, ROW_MASK>(__float_as_int(v)));
  }
  const int ok = dpp_or_own<CTRL, ROW_MASK>(key);
  if (ov > v || (ov == v && ok < key)) { v = ov; key = ok; }
}

template <class T, int NB>
__global__ __launch_bounds__(BLOCK) void getrf_reg_kernel(T* __restrict__ LUout, const T* __restrict__ Ain,
                                                         long long* __restrict__ perm_out,
                                                         T* __restrict__ sign_out, T* __restrict__ logabs_out,
                                                         int n, T* __restrict__ Lstage, int* __restrict__ status) {
  __shared__ T s_col[2][128], s_row[128], s_piv[128];
  __shared__ int s_pos[128], s_cur[128];
  __shared__ T s_sum[BLOCK / 64];
  __shared__ int s_neg[BLOCK / 64];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, lane = tid & 63, wid = tid >> 6;
  const long long mat = blockIdx.x;
  const T* A = Ain + mat * (long long)n * n;
  T* Lg = Lstage + mat * (long long)n * n;
  T a[NB][NB];
  unsigned active = 0;
#pragma unroll
  for (int r = 0; r < NB; r++) {
    const int i = ty + 16 * r;
    if (i < n) active |= 1u << r;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      const int j = tx + 16 * c;
      a[r][c] = (i < n && j < n) ? A[(long long)i * n + j] : T(0);
    }
  }
  if (tid < 128) { s_pos[tid] = tid; s_cur[tid] = tid; }
  if (tx == 0) {
#pragma unroll
    for (int r = 0; r < NB; r++) s_col[0][ty + 16 * r] = a[r][0];
  }
  int flips = 0;
  bool singular = false;
  __syncthreads();
  for (int k = 0; k < n; k++) {
    const T* col = s_col[k & 1];
    // ---- pivot: among the rows not pivoted yet, largest |a_ik|, first in the current order ----
    T v = T(-1);
    int key = 0x7fffffff;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int rho = lane + 64 * h;
      if (rho < n) {
        const int p = s_pos[rho];
        if (p >= k) {
          const T x = dev_abs(col[rho]);
          const int ky = p * 256 + rho;
          if (x > v || (x == v && ky < key)) { v = x; key = ky; }
        }
      }
    }
    argmax_step<T, 0x111, 0xf>(v, key);  // row_shr:1
    argmax_step<T, 0x112, 0xf>(v, key);  // row_shr:2
    argmax_step<T, 0x114, 0xf>(v, key);  // row_shr:4
    argmax_step<T, 0x118, 0xf>(v, key);  // row_shr:8
    argmax_step<T, 0x142, 0xa>(v, key);  // row_bcast:15 into rows 1, 3
    argmax_step<T, 0x143, 0xc>(v, key);  // row_bcast:31 into rows 2, 3
    key = __builtin_amdgcn_readlane(key, 63);
    const int rho = key & 255, q = key >> 8;
    const T piv = col[rho];
    if (piv == T(0)) singular = true;  // dgetf2: info = k+1, no scaling, elimination continues
    const T rp = piv == T(0) ? T(1) : T(1) / piv;
    // ---- the owners of the pivot row publish it; that row is done ----
    if (ty == (rho & 15)) {
#pragma unroll
      for (int r = 0; r < NB; r++)
        if (r == (rho >> 4)) {
#pragma unroll
          for (int c = 0; c < NB; c++) s_row[tx + 16 * c] = a[r][c];
          active &= ~(1u << r);
        }
    }
    __syncthreads();
    if (tid == 0) {  // the bookkeeping of LAPACK's row interchange k <-> q
      const int sigma = s_cur[k];
      s_cur[q] = sigma; s_pos[sigma] = q;
      s_cur[k] = rho; s_pos[rho] = k;
      s_piv[k] = piv;
      flips += q != k;
    }
    // ---- rank-1 update out of registers ----
    // The multipliers of this column go to the staging matrix Lg (physical row order), so the
    // register entries at or left of column k of a still-active row are dead from here on and
    // the update needs no per-element tests: a frozen row gets l = 0 (a - 0 * row = a), whole
    // column blocks left of k are skipped by a wave-uniform branch.
    const int cb = k >> 4;
    T l[NB];
#pragma unroll
    for (int r = 0; r < NB; r++) {
      const bool on = (active >> r) & 1u;
      l[r] = on ? col[ty + 16 * r] * rp : T(0);
      if (on && tx == (k & 15)) Lg[(long long)(ty + 16 * r) * n + k] = l[r];
    }
#pragma unroll
    for (int c = 0; c < NB; c++) {
      if (c >= cb) {
        const T rv = s_row[tx + 16 * c];
#pragma unroll
        for (int r = 0; r < NB; r++) a[r][c] -= l[r] * rv;
      }
    }
    // ---- the owners of column k+1 publish it ----
    if (k + 1 < n && tx == ((k + 1) & 15)) {
#pragma unroll
      for (int c = 0; c < NB; c++)
        if (c == ((k + 1) >> 4)) {
#pragma unroll
          for (int r = 0; r < NB; r++) s_col[(k + 1) & 1][ty + 16 * r] = a[r][c];
        }
    }
    __syncthreads();
  }
  // ---- write out: physical row i goes to row s_pos[i] of P*A = L*U ----
  T* Lo = LUout + mat * (long long)n * n;
#pragma unroll
  for (int r = 0; r < NB; r++) {
    const int i = ty + 16 * r;
    if (i < n) {
      const int p = s_pos[i];  // = the column at which row i was pivoted
      const long long orow = (long long)p * n;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        const int j = tx + 16 * c;
        // left of the diagonal the multipliers (written by this very thread), U from registers
        if (j < n) Lo[orow + j] = j < p ? Lg[(long long)i * n + j] : a[r][c];
      }
    }
  }
  for (int i = tid; i < n; i += BLOCK) perm_out[mat * n + i] = s_cur[i];
  // slogdet: sign of the permutation times the signs of the pivots, sum of log|pivot|
  T la = T(0);
  int neg = 0;
  for (int i = tid; i < n; i += BLOCK) {
    const T d = s_piv[i];
    neg += d < T(0);
    la += log(dev_abs(d));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { la += __shfl_xor(la, o); neg += __shfl_xor(neg, o); }
  if (lane == 0) { s_sum[wid] = la; s_neg[wid] = neg; }
  __syncthreads();
  if (tid == 0) {
    la = T(0); neg = 0;
    for (int w = 0; w < BLOCK / 64; w++) { la += s_sum[w]; neg += s_neg[w]; }
    T sg = ((flips + neg) & 1) ? T(-1) : T(1);
    if (singular) { sg = T(0); la = -__builtin_huge_val(); }
    sign_out[mat] = sg;
    logabs_out[mat] = la;
    if (singular && status != nullptr) atomicOr(status, 2);
  }
}

template <class T, int NB>
int launch_getrf_reg(long long batch, long long n, const void* A, void* LU, void* perm, void* sign, void* logabs,
                     int flag_singular) {
  void* stage = nullptr;  // multipliers in physical row order until the permutation is known
  int r = pthip_alloc((size_t)batch * n * n * sizeof(T), &stage);
  if (r) return r;
  PTHIP_KLAUNCH((getrf_reg_kernel<T, NB>), dim3((unsigned)batch), dim3(BLOCK), 0, pthip::ctx().stream, (T*)LU,
                     (const T*)A, (long long*)perm, (T*)sign, (T*)logabs, (int)n, (T*)stage,
                     flag_singular ? (int*)pthip_status_ptr() : (int*)nullptr);
  r = pthip::post_launch("getrf_reg");
  pthip_free(stage);  // stream-ordered reuse keeps this safe
  return r;
}

template <class T>
int getrf_typed(long long batch, long long n, const void* A, void* LU, void* perm, void* sign,
                void* logabs, int flag_singular) {
  if (batch == 0) return 0;
  static const bool no_reg = getenv("PTHIP_LU_NO_REG") != nullptr;
  if (n <= 128 && !no_reg) {
    if (n <= 16) return launch_getrf_reg<T, 1>(batch, n, A, LU, perm, sign, logabs, flag_singular);
    if (n <= 32) return launch_getrf_reg<T, 2>(batch, n, A, LU, perm, sign, logabs, flag_singular);
    if (n <= 64) return launch_getrf_reg<T, 4>(batch, n, A, LU, perm, sign, logabs, flag_singular);
    return launch_getrf_reg<T, 8>(batch, n, A, LU, perm, sign, logabs, flag_singular);
  }
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
  PTHIP_KLAUNCH(k, dim3((unsigned)batch), dim3(BLOCK), lds ? need : 0, st, (T*)LU, (const T*)A,
                     (long long*)perm, (T*)sign, (T*)logabs, (int)n, lds ? 1 : 0, (T*)scratch,
                     flag_singular ? (int*)pthip_status_ptr() : (int*)nullptr);
  int r = pthip::post_launch("getrf");
  if (scratch) pthip_free(scratch);  // stream-ordered reuse keeps this safe
  return r;
}

// P * I for the inverse: row i of the result is the unit vector e_perm[i]
template <class T>
__global__ __launch_bounds__(BLOCK) void permuted_identity_kernel(T* __restrict__ out, const long long* __restrict__ perm,
                                                                 long long n) {
  const long long total = n * n;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long i = e / n, j = e - i * n;
    out[e] = j == perm[i] ? T(1) : T(0);
  }
}

}  // namespace

extern "C" int pthip_permuted_identity(int dtype, int64_t n, const void* perm, void* out) {
  PTHIP_REQUIRE_INIT();
  if (n <= 0) return 0;
  long long g = (n * n + BLOCK - 1) / BLOCK;
  if (g > 4096) g = 4096;
  hipStream_t st = pthip::ctx().stream;
  if (dtype == PTHIP_F64) PTHIP_KLAUNCH(permuted_identity_kernel<double>, dim3((unsigned)g), dim3(BLOCK), 0, st, (double*)out, (const long long*)perm, (long long)n);
  else if (dtype == PTHIP_F32) PTHIP_KLAUNCH(permuted_identity_kernel<float>, dim3((unsigned)g), dim3(BLOCK), 0, st, (float*)out, (const long long*)perm, (long long)n);
  else return pthip::set_error("pthip_permuted_identity: dtype %d not supported (float32/float64 only)", dtype);
  return pthip::post_launch("permuted_identity");
}

// ---- LAPACK's pivot vector from the row order, and back ------------------------------------------
// getrf above returns the row ORDER (LU rows = A[perm]); scipy's getrf — what LUFactor.perform
// returns (linalg/decomposition/lu.py:239-300) — reports the interchanges: row i was swapped with
// row piv[i] (0-based).  One thread replays the swaps (n sequential steps on <= a few hundred
// rows); the workgroup NaN-fills a factor with an exactly zero pivot (perform: `info != 0`).
namespace {
template <class T>
__global__ __launch_bounds__(256) void lu_factor_finish_kernel(T* __restrict__ LU, const long long* __restrict__ perm,
                                                               int* __restrict__ piv, int n) {
  extern __shared__ int sm_[];  // cur[n]: original row at each position, inv[n]: position of each row
  __shared__ int singular;
  const long long b = blockIdx.x;
  T* lu = LU + b * (long long)n * n;
  perm += b * n;
  piv += b * n;
  int* cur = sm_;
  int* inv = sm_ + n;
  if (threadIdx.x == 0) singular = 0;
  for (int i = threadIdx.x; i < n; i += 256) { cur[i] = i; inv[i] = i; }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256)
    if (lu[(long long)i * n + i] == T(0)) singular = 1;
  if (threadIdx.x == 0) {
    for (int i = 0; i < n; i++) {
      const int r = (int)perm[i], j = inv[r], ri = cur[i];
      piv[i] = j;
      cur[i] = r; cur[j] = ri;
      inv[r] = i; inv[ri] = j;
    }
  }
  __syncthreads();
  if (singular) {
    const T nan = T(NAN);
    for (long long i = threadIdx.x; i < (long long)n * n; i += 256) lu[i] = nan;
  }
}

// PivotToPermutations.perform (lu.py:206-231): p = arange(n); for i: swap(p[i], p[piv[i]]);
// inverse -> p, else argsort(p) (p is a permutation: its inverse).
template <class I>
__global__ __launch_bounds__(256) void pivots_to_perm_kernel(const I* __restrict__ piv, long long* __restrict__ out,
                                                             int n, int inverse, int* status) {
  extern __shared__ int sm_[];
  const long long b = blockIdx.x;
  piv += b * n;
  out += b * n;
  int* p = sm_;
  for (int i = threadIdx.x; i < n; i += 256) p[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < n; i++) {
      long long j = (long long)piv[i];
      if (j < 0 || j >= n) { atomicOr(status, 1); j = i; }  // IndexError, like the NumPy loop
      const int t = p[i]; p[i] = p[j]; p[j] = t;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    if (inverse) out[i] = p[i];
    else out[p[i]] = i;
  }
}
}  // namespace

extern "C" int pthip_lu_factor_finish(int dtype, int64_t batch, int64_t n, void* LU, const void* perm, void* piv) {
  PTHIP_REQUIRE_INIT();
  if (n == 0 || batch == 0) return 0;
  if (n > 8000) return pthip::set_error("pthip_lu_factor_finish: n = %lld exceeds the LDS bookkeeping", (long long)n);
  hipStream_t st = pthip::ctx().stream;
  const size_t sh = (size_t)n * 2 * sizeof(int);
  if (dtype == PTHIP_F64)
    PTHIP_KLAUNCH((lu_factor_finish_kernel<double>), dim3((unsigned)batch), dim3(256), sh, st, (double*)LU, (const long long*)perm, (int*)piv, (int)n);
  else if (dtype == PTHIP_F32)
    PTHIP_KLAUNCH((lu_factor_finish_kernel<float>), dim3((unsigned)batch), dim3(256), sh, st, (float*)LU, (const long long*)perm, (int*)piv, (int)n);
  else
    return pthip::set_error("pthip_lu_factor_finish: dtype %d not supported (float32/float64 only)", dtype);
  return pthip::post_launch("lu_factor_finish");
}

extern "C" int pthip_pivots_to_perm(int itemsize, int inverse, int64_t batch, int64_t n, const void* piv, void* out) {
  PTHIP_REQUIRE_INIT();
  if (n == 0 || batch == 0) return 0;
  if (n > 16000) return pthip::set_error("pthip_pivots_to_perm: n = %lld exceeds the LDS bookkeeping", (long long)n);
  hipStream_t st = pthip::ctx().stream;
  int* status = (int*)pthip_status_ptr();
  const size_t sh = (size_t)n * sizeof(int);
  if (itemsize == 4)
    PTHIP_KLAUNCH((pivots_to_perm_kernel<int>), dim3((unsigned)batch), dim3(256), sh, st, (const int*)piv, (long long*)out, (int)n, inverse, status);
  else if (itemsize == 8)
    PTHIP_KLAUNCH((pivots_to_perm_kernel<long long>), dim3((unsigned)batch), dim3(256), sh, st, (const long long*)piv, (long long*)out, (int)n, inverse, status);
  else
    return pthip::set_error("pthip_pivots_to_perm: pivots must be int32 or int64");
  return pthip::post_launch("pivots_to_perm");
}

extern "C" int pthip_getrf(int dtype, int64_t batch, int64_t n, const void* A, void* LU, void* perm,
                           void* sign, void* logabsdet, int flag_singular) {
  PTHIP_REQUIRE_INIT();
  if (n == 0) return 0;
  if (dtype == PTHIP_F64) return getrf_typed<double>(batch, n, A, LU, perm, sign, logabsdet, flag_singular);
  if (dtype == PTHIP_F32) return getrf_typed<float>(batch, n, A, LU, perm, sign, logabsdet, flag_singular);
  return pthip::set_error("pthip_getrf: dtype %d not supported (float32/float64 only)", dtype);
}
