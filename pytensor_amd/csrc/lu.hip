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

#include <cstring>
#include <type_traits>
#include <utility>

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


// ------------------------------------------------------------------------------------------
// n > 140: blocked right-looking LU with partial pivoting over the working matrix W (= the output,
// row-major), LB-column panels.  Per panel: (1) lu_panel_kernel factors the (n - k0) x LB panel —
// one ROW per thread, the row's LB entries in registers, ceil(rows / 256) workgroups resident at once.
// Per column every workgroup finds its best row (first maximum, like idamax) and publishes
// {|value|, row, the row's LB entries} as self-validating 16-byte pairs (see linalg.hip, trsv_dag_kernel:
// no fence per hop); all workgroups read all candidates, agree on the pivot, and already hold the pivot
// row for the rank-1 update: ONE memory round trip per column.  The owner of row k also publishes its
// row, which the owner of the pivot row takes in exchange.  (2) lu_laswp_kernel applies the panel's
// interchanges to the columns left and right of it, (3) lu_u12_kernel solves L11 U12 = A12 (L11 in LDS,
// one thread per column), (4) the trailing matrix takes A22 -= L21 U12 on the MFMA GEMM (gemm.hip, in
// place).  lu_finish_kernel turns the interchanges into the gather vector, its sign, and log|det|.
// A zero pivot: no scaling, elimination continues (dgetf2), sign 0 / log|det| -inf.
// ------------------------------------------------------------------------------------------
constexpr int LB = 32;  // panel width
constexpr unsigned long long LU_MAGIC = 0x7ff4c0de5ea1ed02ull;
constexpr int LU_SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ unsigned long long lu_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned long long lu_bits(float v) { return (unsigned long long)__float_as_uint(v); }
__device__ __forceinline__ void lu_from_bits(unsigned long long b, double& v) { v = __longlong_as_double((long long)b); }
__device__ __forceinline__ void lu_from_bits(unsigned long long b, float& v) { v = __uint_as_float((unsigned)b); }

// {bits, bits ^ MAGIC} in one 16-byte write-through store; a reader accepts the pair only when the two
// halves match (a torn or not-yet-written pair over the zeroed array fails unless it equals the final one)
__device__ __forceinline__ void lu_publish(unsigned long long* slot, unsigned long long bits) {
  typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
  u2 pr = {bits, bits ^ LU_MAGIC};
  // (s_nop: a store of more than 8 bytes reads its upper data dwords after issue; the compiler pads its own
  //  stores against a following VALU write of those registers, but cannot see into this one)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"((u2*)slot), "v"(pr) : "memory");
}
__device__ __forceinline__ bool lu_poll(const unsigned long long* slot, unsigned long long& bits) {
  bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (bits ^ b) == LU_MAGIC;
}

template <class T>
__device__ __forceinline__ void wave_argmax(T& v, int& key) {  // result valid in lane 63
  argmax_step<T, 0x111, 0xf>(v, key);
  argmax_step<T, 0x112, 0xf>(v, key);
  argmax_step<T, 0x114, 0xf>(v, key);
  argmax_step<T, 0x118, 0xf>(v, key);
  argmax_step<T, 0x142, 0xa>(v, key);
  argmax_step<T, 0x143, 0xc>(v, key);
}

template <class F, int... Js>
__device__ __forceinline__ bool lu_for_each_column(F&& f, std::integer_sequence<int, Js...>) {
  return (f(std::integral_constant<int, Js>{}) && ...);
}

template <class T>
__global__ __launch_bounds__(BLOCK) void lu_panel_kernel(T* __restrict__ W, long long ld, int n, int k0, int nW,
                                                        unsigned long long* __restrict__ box,
                                                        unsigned long long* __restrict__ boxk, int* __restrict__ ipiv,
                                                        int* __restrict__ info, int* __restrict__ abortflag,
                                                        int* __restrict__ status) {
  __shared__ T s_val[BLOCK / 64];
  __shared__ int s_row[BLOCK / 64];
  __shared__ T s_u[LB], s_rk[LB];
  __shared__ int s_p, s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, w = blockIdx.x;
  const int pw = (n - k0) < LB ? (n - k0) : LB;
  const long long grow = (long long)k0 + (long long)w * BLOCK + tid;  // this thread's row
  const bool have = grow < n;
  T a[LB];
#pragma unroll
  for (int c = 0; c < LB; c++) a[c] = (have && c < pw) ? W[grow * ld + k0 + c] : T(0);
  // (the column index must be a compile-time constant: a[] lives in registers; the loop over the LB
  //  columns is too large for the unroller, so it is spelled as a fold over an index sequence)
  auto column = [&](auto jc) -> bool {
    constexpr int j = decltype(jc)::value;
    if (j < pw) {  // (uniform: the last panel may be narrower)
    const int k = k0 + j;
    // ---- this workgroup's candidate: largest |a_ik| among its rows i >= k, first such row
    T v = (have && grow >= k) ? dev_abs(a[j]) : T(-1);
    int key = (have && grow >= k) ? (int)grow : 0x7fffffff;
    wave_argmax<T>(v, key);
    if (lane == 63) { s_val[wid] = v; s_row[wid] = key; }
    __syncthreads();
    T bv = s_val[0];
    int br = s_row[0];
#pragma unroll
    for (int q = 1; q < BLOCK / 64; q++)
      if (s_val[q] > bv || (s_val[q] == bv && s_row[q] < br)) { bv = s_val[q]; br = s_row[q]; }
    const bool mine = have && grow == br;
    if (nW > 1) {
      if (mine) {
        unsigned long long* rec = box + ((long long)(j * nW + w) * (LB + 2)) * 2;
        lu_publish(rec, lu_bits(bv));
        lu_publish(rec + 2, (unsigned long long)br);
#pragma unroll
        for (int c = 0; c < LB; c++) lu_publish(rec + (2 + c) * 2, lu_bits(a[c]));
      }
      if (have && grow == k) {
#pragma unroll
        for (int c = 0; c < LB; c++) lu_publish(boxk + ((long long)j * LB + c) * 2, lu_bits(a[c]));
      }
      if (wid == 0) {
        // all candidates -> the pivot (largest value, then smallest row: the first maximum in row order)
        T cv = T(-1);
        int cr = 0x7fffffff;
        int spins = 0;
        bool ok = true;
        for (int base = 0; base < nW && ok; base += 64) {
          const int cand = base + lane;
          const unsigned long long* rec = box + ((long long)(j * nW + (cand < nW ? cand : 0)) * (LB + 2)) * 2;
          unsigned long long vb = 0, rb = 0;
          for (;;) {
            const bool got = cand >= nW || (lu_poll(rec, vb) && lu_poll(rec + 2, rb));
            if (__builtin_amdgcn_ballot_w64(!got) == 0ull) break;
            if (++spins > LU_SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(abortflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (ok && cand < nW) {
            T x;
            lu_from_bits(vb, x);
            const int r = (int)rb;
            if (x > cv || (x == cv && r < cr)) { cv = x; cr = r; }
          }
        }
        wave_argmax<T>(cv, cr);
        const int p = __builtin_amdgcn_readlane(cr, 63);
        const int wstar = ok ? (p - k0) / BLOCK : 0;
        if (ok && lane < LB) {
          const unsigned long long* rec = box + ((long long)(j * nW + wstar) * (LB + 2)) * 2 + (2 + lane) * 2;
          const unsigned long long* rk = boxk + ((long long)j * LB + lane) * 2;
          unsigned long long ub = 0, kb = 0;
          bool got = false;
          for (;;) {
            got = lu_poll(rec, ub) && lu_poll(rk, kb);
            if (got) break;
            if (++spins > LU_SPIN_LIMIT) break;
            __builtin_amdgcn_s_sleep(1);
          }
          T x, y;
          lu_from_bits(ub, x);
          lu_from_bits(kb, y);
          s_u[lane] = x;
          s_rk[lane] = y;
          if (!got) ok = false;
        }
        const bool allok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
        if (lane == 0) { s_p = p; s_ok = allok; }
      }
      __syncthreads();
      if (!s_ok) {
        if (tid == 0) { atomicOr(status, 16); __hip_atomic_store(abortflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        return false;
      }
    } else {
      if (mine) {
#pragma unroll
        for (int c = 0; c < LB; c++) s_u[c] = a[c];
        s_p = br;
      }
      if (have && grow == k) {
#pragma unroll
        for (int c = 0; c < LB; c++) s_rk[c] = a[c];
      }
      __syncthreads();
    }
    const int p = s_p;
    // ---- the interchange k <-> p inside the panel
    if (p != k && have) {
      if (grow == p) {
#pragma unroll
        for (int c = 0; c < LB; c++) a[c] = s_rk[c];
      } else if (grow == k) {
#pragma unroll
        for (int c = 0; c < LB; c++) a[c] = s_u[c];
      }
    }
    if (w == 0 && tid == 0) ipiv[k] = p;
    // ---- scale by the reciprocal pivot (dgetf2), rank-1 update of the panel's later columns
    const T piv = s_u[j];
    if (piv == T(0) && w == 0 && tid == 0) atomicOr(info, 1);
    const T rp = piv == T(0) ? T(1) : T(1) / piv;
    if (have && grow > k) {
      const T l = a[j] * rp;
      a[j] = l;
#pragma unroll
      for (int c = j + 1; c < LB; c++) a[c] -= l * s_u[c];
    }
    __syncthreads();
  
    }
    return true;
  };
  if (!lu_for_each_column(column, std::make_integer_sequence<int, LB>{})) return;
  if (have) {
#pragma unroll
    for (int c = 0; c < LB; c++)
      if (c < pw) W[grow * ld + k0 + c] = a[c];
  }
}

// the interchanges of panel [k0, k0 + pw) applied to the columns outside it: one thread per column
template <class T>
__global__ __launch_bounds__(BLOCK) void lu_laswp_kernel(T* __restrict__ W, long long ld, int n, int k0, int pw,
                                                        const int* __restrict__ ipiv) {
  int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= n - pw) return;
  if (c >= k0) c += pw;
  for (int j = 0; j < pw; j++) {
    const int k = k0 + j, p = ipiv[k];
    if (p != k) {
      const T t = W[(long long)k * ld + c];
      W[(long long)k * ld + c] = W[(long long)p * ld + c];
      W[(long long)p * ld + c] = t;
    }
  }
}

// U12 = L11^-1 A12 (L11 unit lower, pw x pw, in LDS): one thread per column right of the panel, its
// column of the solution in LDS as well (a register-resident column unrolls into 496 hoisted LDS loads
// and spills)
template <class T>
__global__ __launch_bounds__(BLOCK) void lu_u12_kernel(T* __restrict__ W, long long ld, int n, int k0, int pw) {
  __shared__ T Ls[LB][LB + 1];
  __shared__ T xs[LB][BLOCK];
  const int tid = threadIdx.x;
  for (int e = tid; e < LB * LB; e += BLOCK) {
    const int r = e / LB, q = e - r * LB;
    Ls[r][q] = (r < pw && q < r) ? W[(long long)(k0 + r) * ld + k0 + q] : T(0);
  }
  const long long c = (long long)k0 + pw + (long long)blockIdx.x * BLOCK + tid;
  const bool have = c < n;
  for (int r = 0; r < pw; r++) xs[r][tid] = have ? W[(long long)(k0 + r) * ld + c] : T(0);
  __syncthreads();
  for (int r = 1; r < pw; r++) {
    T sacc = xs[r][tid];
    for (int q = 0; q < r; q++) sacc -= Ls[r][q] * xs[q][tid];
    xs[r][tid] = sacc;
  }
  if (have)
    for (int r = 1; r < pw; r++) W[(long long)(k0 + r) * ld + c] = xs[r][tid];
}

// gather vector from the interchanges (p = arange(n); for k: swap(p[k], p[ipiv[k]])), sign, log|det|
template <class T>
__global__ __launch_bounds__(BLOCK) void lu_finish_kernel(const T* __restrict__ W, long long ld, int n,
                                                         const int* __restrict__ ipiv, const int* __restrict__ info,
                                                         long long* __restrict__ perm_out, T* __restrict__ sign_out,
                                                         T* __restrict__ logabs_out, int* __restrict__ status) {
  extern __shared__ int sm_p[];
  __shared__ T s_sum[BLOCK / 64];
  __shared__ int s_neg[BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < n; i += BLOCK) sm_p[i] = i;
  T la = T(0);
  int neg = 0;
  for (int i = tid; i < n; i += BLOCK) {
    const T d = W[(long long)i * ld + i];
    neg += d < T(0);
    la += log(dev_abs(d));
    neg += ipiv[i] != i;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { la += __shfl_xor(la, o); neg += __shfl_xor(neg, o); }
  if (lane == 0) { s_sum[wid] = la; s_neg[wid] = neg; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < n; k++) {
      const int p = ipiv[k];
      const int t = sm_p[k]; sm_p[k] = sm_p[p]; sm_p[p] = t;
    }
    la = T(0); neg = 0;
    for (int q = 0; q < BLOCK / 64; q++) { la += s_sum[q]; neg += s_neg[q]; }
    T sg = (neg & 1) ? T(-1) : T(1);
    if (*info) { sg = T(0); la = -__builtin_huge_val(); if (status != nullptr) atomicOr(status, 2); }
    sign_out[0] = sg;
    logabs_out[0] = la;
  }
  __syncthreads();
  for (int i = tid; i < n; i += BLOCK) perm_out[i] = sm_p[i];
}

template <class T>
int getrf_blocked(long long n, const T* A, T* LU, long long* perm, T* sign, T* logabs, int flag_singular) {
  hipStream_t st = pthip::ctx().stream;
  const int dt = sizeof(T) == 8 ? PTHIP_F64 : PTHIP_F32;
  const int nWmax = (int)((n + BLOCK - 1) / BLOCK);
  const size_t boxbytes = (size_t)LB * nWmax * (LB + 2) * 16, boxkbytes = (size_t)LB * LB * 16;
  const size_t ibytes = ((size_t)n * sizeof(int) + 255) / 256 * 256;
  void* scratch = nullptr;
  int r = pthip_alloc(boxbytes + boxkbytes + ibytes + 256, &scratch);
  if (r) return r;
  auto fail = [&](int rc) { pthip_free(scratch); return rc; };
  unsigned long long* box = (unsigned long long*)scratch;
  unsigned long long* boxk = (unsigned long long*)((char*)scratch + boxbytes);
  int* ipiv = (int*)((char*)scratch + boxbytes + boxkbytes);
  int* flags = (int*)((char*)scratch + boxbytes + boxkbytes + ibytes);  // [0] info, [1] abort
  if (hipError_t e = pthip::memset_async(flags, 0, 256, st); e != hipSuccess) return fail(pthip::check(e, "lu flags memset"));
  if (hipError_t e = pthip::memcpy_async(LU, A, (size_t)n * n * sizeof(T), hipMemcpyDeviceToDevice, st); e != hipSuccess)
    return fail(pthip::check(e, "lu copy"));
  for (long long k0 = 0; k0 < n; k0 += LB) {
    const int pw = (int)((n - k0) < LB ? (n - k0) : LB);
    const int nW = (int)((n - k0 + BLOCK - 1) / BLOCK);
    if (nW > 1)
      if (hipError_t e = pthip::memset_async(box, 0, boxbytes + boxkbytes, st); e != hipSuccess) return fail(pthip::check(e, "lu box memset"));
    PTHIP_KLAUNCH((lu_panel_kernel<T>), dim3((unsigned)nW), dim3(BLOCK), 0, st, LU, n, (int)n, (int)k0, nW, box, boxk, ipiv, flags, flags + 1,
                  pthip::ctx().status_dev);
    if ((r = pthip::post_launch("lu_panel"))) return fail(r);
    if (n - pw > 0) {
      PTHIP_KLAUNCH((lu_laswp_kernel<T>), dim3((unsigned)((n - pw + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, LU, n, (int)n, (int)k0, pw, (const int*)ipiv);
      if ((r = pthip::post_launch("lu_laswp"))) return fail(r);
    }
    const long long rest = n - k0 - pw;
    if (rest > 0) {
      PTHIP_KLAUNCH((lu_u12_kernel<T>), dim3((unsigned)((rest + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, LU, n, (int)n, (int)k0, pw);
      if ((r = pthip::post_launch("lu_u12"))) return fail(r);
      r = pthip::gemm_inplace(dt, rest, rest, pw, -1.0, LU + (k0 + pw) * n + k0, n, 1, LU + k0 * n + k0 + pw, n, 1, 1.0,
                              LU + (k0 + pw) * n + k0 + pw, n);
      if (r) return fail(r);
    }
  }
  auto kf = lu_finish_kernel<T>;
  const size_t sh = (size_t)n * sizeof(int);
  static bool attr = false;
  if (!attr && sh > 48 * 1024) {
    if (hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); e != hipSuccess)
      return fail(pthip::check(e, "lu_finish attribute"));
    attr = true;
  }
  PTHIP_KLAUNCH(kf, dim3(1), dim3(BLOCK), sh, st, (const T*)LU, n, (int)n, (const int*)ipiv, (const int*)flags, perm, sign, logabs,
                flag_singular ? pthip::ctx().status_dev : (int*)nullptr);
  r = pthip::post_launch("lu_finish");
  pthip_free(scratch);
  return r;
}


// ------------------------------------------------------------------------------------------
// Round 4: the panel as a LOOP (lu_panel2_kernel) and one launch for interchange + U12 (lu_swap_u12_kernel).
//
// lu_panel_kernel (round 3) spent 4.4 us per column.  Measured with in-kernel clocks (PTHIP_LU_PROF=1,
// profiles/r4d_getrf_phases.txt), the first looped form of this kernel still spent 1.4 us finding and staging a
// candidate, 1.1-2.2 us in the cross-workgroup exchange and 1.9 us in the rank-1 update of a 64-wide panel:
// the single lane that owns the candidate row wrote all PB entries to LDS (twice: candidate and row k for the
// physical interchange), each wave polled its candidates one after the other, and the update waited for an LDS
// load per element because every element sat under its own predicate.  This form:
//   * one ROW per thread in registers, in a ROTATED frame — the current column is always a[0], after a column the
//     array rotates left by one (compile-time register indices in a real loop; 15 KB of code instead of 270);
//   * rows never move: a thread tracks the CURRENT POSITION of its row (LAPACK's interchange k <-> p moves the row
//     at position k to position p: the thread holding position k just notes `cur = p`; the pivot row is done and
//     remembers k).  Ties still go to the first row in the current order, like idamax.  Rows go to their final
//     positions when the panel is written back.  No row-k exchange, no second record per column;
//   * a candidate record = {row, |value| implied by entry 0, the LIVE entries (positions 0 .. PB-1-j)} as epoch-tagged
//     self-validating 16-byte pairs, one lane each; wave q of every workgroup polls the records of workgroups
//     q, q+4, ... up to four at a time in ONE round trip, contents included;
//   * the update loads the pivot row from LDS in batches, then runs select-guarded FMAs.
// LDS buffers alternate with the column parity: two barriers per column (one with a single workgroup).  The pair
// tag carries the global column index and a per-call nonce: the boxes are zeroed once per call.  plist tells
// lu_swap_u12_kernel which original row ends up at each touched position: [0] number of displaced rows,
// [1 .. PB] the original row now at position k0+i, then (position, original row) pairs of the displaced ones.
// ------------------------------------------------------------------------------------------
constexpr unsigned long long LU_EPOCH_MUL = 0x9E3779B97F4A7C15ull;

// X1: every workgroup of the panel sits on ONE XCD (the launch is 8x as wide and only the workgroups with id % 8 == 0
// take part — ids go round-robin over the 8 XCDs), so the record can stop in that XCD's L2: a store without
// write-through, read by the agent-scope loads of the others out of the same L2.  tools/ubench/hop.hip: 315-335 ns
// one way against 470-580 ns for the write-through pair (across XCDs the plain store is never seen).
template <bool X1>
__device__ __forceinline__ void lu_publish_t(unsigned long long* slot, unsigned long long bits, unsigned long long tag) {
  typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
  u2 pr = {bits, bits ^ tag};
  if constexpr (X1) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 2" : : "v"((u2*)slot), "v"(pr) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"((u2*)slot), "v"(pr) : "memory");
}
__device__ __forceinline__ bool lu_poll_t(const unsigned long long* slot, unsigned long long& bits, unsigned long long tag) {
  bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (bits ^ b) == tag;
}
template <int L>
__device__ __forceinline__ unsigned long long lane_bcast_u64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, L);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), L);
  return ((unsigned long long)hi << 32) | lo;
}

// TB: threads (= rows) per workgroup.  256: four waves; 512 (PTHIP_LU_PANEL_THREADS=512, PB = 32 only): half as many
// workgroups take part in the exchange and seven waves poll at most two records each in one round trip.
// PROF (PTHIP_LU_PROF=1): thread 0 of workgroup 0 adds up the 100 MHz wall-clock ticks of the phases of every
// column — [0] A (candidate + barrier), [1] B (publish / poll / barrier), [2] C (update), [3] rotate,
// [4] shader cycles of the whole loop, [5] its wall ticks — into prof[0..5].
template <class T, int PB, bool PROF, bool X1, int TB>
__global__ __launch_bounds__(TB) void lu_panel2_kernel(T* __restrict__ W, long long ld, int n, int k0, int nW,
                                                         unsigned long long* __restrict__ box, int* __restrict__ ipiv,
                                                         int* __restrict__ plist, int* __restrict__ info,
                                                         int* __restrict__ abortflag, int* __restrict__ status,
                                                         unsigned long long nonce, long long* __restrict__ prof) {
  constexpr int NWAVE = TB / 64;
  constexpr int REC = PB + 2;  // pairs of a candidate record: row, (spare), up to PB entries
  constexpr int GRP = 4;       // candidates a wave polls at once
  static_assert(PB <= 64 && PB % 8 == 0, "panel width");
  __shared__ T s_val[2][NWAVE], g_val[2][NWAVE];
  __shared__ int s_row[2][NWAVE], g_row[2][NWAVE];
  __shared__ __attribute__((aligned(16))) T s_cand[2][NWAVE][PB];
  __shared__ __attribute__((aligned(16))) T g_cont[2][NWAVE][PB];
  __shared__ int s_ok[2];
  __shared__ int pl_cnt;
  // bookkeeping stays in LDS until the panel is done: a global store inside the loop is still in flight at the
  // next barrier (s_waitcnt vmcnt(0) in front of every s_barrier) and was costing ~0.4 us per column
  __shared__ int s_ipiv[PB], s_top[PB];
  if (X1 && (blockIdx.x & 7) != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, w = X1 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int pw = (n - k0) < PB ? (n - k0) : PB;
  const long long grow = (long long)k0 + (long long)w * TB + tid;  // the row this thread loaded
  const bool have = grow < n;
  int cur = have ? (int)grow : 0x7fffffff;  // its current position under the interchanges so far
  bool done = false;                        // it became a pivot row (cur = its final position then)
  T a[PB];
#pragma unroll
  for (int c = 0; c < PB; c++) a[c] = (have && c < pw) ? W[grow * ld + k0 + c] : T(0);
  if (tid == 0) { s_ok[0] = 1; s_ok[1] = 1; pl_cnt = 0; }
  if (tid < PB) { s_ipiv[tid] = 0; s_top[tid] = -1; }
  __syncthreads();
  long long pt[4] = {0, 0, 0, 0}, tq = 0, c_start = 0, w_start = 0;
  if constexpr (PROF) { c_start = clock64(); w_start = wall_clock64(); }
#pragma unroll 1
  for (int j = 0; j < PB; j++) {
    const int pb = j & 1;
    if constexpr (PROF) tq = wall_clock64();
    if (j < pw) {  // (uniform: the last panel may be narrower; the frame still rotates PB times)
      const int k = k0 + j;
      const int live = PB - j;  // positions 0 .. live-1 of the frame hold columns j .. PB-1
      const unsigned long long tag = LU_MAGIC ^ ((unsigned long long)(k + 1) * LU_EPOCH_MUL) ^ nonce;
      // ---- A: this workgroup's candidate: largest |a_ik| among its rows not yet pivoted, first in the current order
      const bool act = have && !done;
      T v = act ? dev_abs(a[0]) : T(-1);
      int key = act ? cur : 0x7fffffff;
      wave_argmax<T>(v, key);
      const int wkey = __builtin_amdgcn_readlane(key, 63);
      if (act && cur == wkey) {
#pragma unroll
        for (int c0 = 0; c0 < PB; c0 += 8)
          if (c0 < live) {
#pragma unroll
            for (int c = c0; c < c0 + 8; c++) s_cand[pb][wid][c] = a[c];
          }
      }
      if (lane == 63) { s_val[pb][wid] = v; s_row[pb][wid] = key; }
      __syncthreads();
      T bv = s_val[pb][0];
      int br = s_row[pb][0], bw = 0;
#pragma unroll
      for (int q = 1; q < NWAVE; q++)
        if (s_val[pb][q] > bv || (s_val[pb][q] == bv && s_row[pb][q] < br)) { bv = s_val[pb][q]; br = s_row[pb][q]; bw = q; }
      const T* u = s_cand[pb][bw];
      int p = br;
      if constexpr (PROF) { const long long t = wall_clock64(); pt[0] += t - tq; tq = t; }
      if (nW > 1) {
        // ---- B: publish {row, live entries} — one pair per lane
        const int npairs = live + 2;
        unsigned long long* rec = box + ((long long)(j * nW + w) * REC) * 2;
        // (the LAST wave publishes and does not poll: a wave's loads return behind its own write-through store,
        //  whose acknowledgement comes from memory; the other NPOLL waves share the candidates)
        constexpr int NPOLL = NWAVE - 1;
        if (wid == NWAVE - 1) {
#pragma unroll
          for (int pi = lane; pi < REC; pi += 64) {
            if (pi < npairs && pi != 1) {
              const unsigned long long bits = pi == 0 ? (unsigned long long)(unsigned)br : lu_bits(s_cand[pb][bw][pi >= 2 ? pi - 2 : 0]);
              lu_publish_t<X1>(rec + 2 * pi, bits, tag);
            }
          }
        }
        // every polling wave folds its share of the candidates, up to GRP at a time, contents included
        T cv = T(-1);
        int cr = 0x7fffffff, spins = 0;
        unsigned long long best0 = 0, best1 = 0;
        bool ok = true;
        const bool has0 = lane < npairs && lane != 1, has1 = lane + 64 < npairs;
        for (int base = wid; wid < NPOLL && base < nW && ok; base += NPOLL * GRP) {
          unsigned long long b0[GRP], b1[GRP];
#pragma unroll
          for (int g = 0; g < GRP; g++) { b0[g] = 0; b1[g] = 0; }
          for (;;) {
            bool got = true;
#pragma unroll
            for (int g = 0; g < GRP; g++) {
              const int cand = base + g * NPOLL;
              if (cand < nW) {
                const unsigned long long* crec = box + ((long long)(j * nW + cand) * REC) * 2;
                if (has0) got = lu_poll_t(crec + 2 * lane, b0[g], tag) && got;
                if (has1) got = lu_poll_t(crec + 2 * (lane + 64), b1[g], tag) && got;
              }
            }
            if (__builtin_amdgcn_ballot_w64(!got) == 0ull) break;
            if (++spins > LU_SPIN_LIMIT || ((spins & 255) == 0 && __hip_atomic_load(abortflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (ok) {
#pragma unroll
            for (int g = 0; g < GRP; g++) {
              if (base + g * NPOLL < nW) {  // (uniform)
                T x;
                lu_from_bits(lane_bcast_u64<2>(b0[g]), x);  // entry 0 of the candidate row: its column-k value
                x = dev_abs(x);
                const int r = (int)lane_bcast_u64<0>(b0[g]);
                if (r != 0x7fffffff && (x > cv || (x == cv && r < cr))) { cv = x; cr = r; best0 = b0[g]; best1 = b1[g]; }
              }
            }
          }
        }
        if (lane == 0) { g_val[pb][wid] = cv; g_row[pb][wid] = cr; }
        if (lane >= 2 && lane < npairs) { T y; lu_from_bits(best0, y); g_cont[pb][wid][lane - 2] = y; }
        if (has1) { T y; lu_from_bits(best1, y); g_cont[pb][wid][lane + 62] = y; }
        if (!ok && lane == 0) s_ok[pb] = 0;
        __syncthreads();
        if (!s_ok[pb]) {
          if (tid == 0) { atomicOr(status, 16); __hip_atomic_store(abortflag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          return;
        }
        T gv = g_val[pb][0];
        int gr = g_row[pb][0], gw = 0;
#pragma unroll
        for (int q = 1; q < NWAVE; q++)
          if (g_val[pb][q] > gv || (g_val[pb][q] == gv && g_row[pb][q] < gr)) { gv = g_val[pb][q]; gr = g_row[pb][q]; gw = q; }
        p = gr;
        u = g_cont[pb][gw];
      }
      if constexpr (PROF) { const long long t = wall_clock64(); pt[1] += t - tq; tq = t; }
      // (a column of NaNs has no maximum: p is out of range — no interchange, the NaNs spread as in dgetf2)
      const bool valid = p >= k && p < n;
      if (!valid) p = k;
      // ---- C: the interchange k <-> p as bookkeeping, scale, rank-1 update of the later columns (positions 1 .. live-1)
      if (act && cur == p) {
        done = true;
        cur = k;
        s_top[j] = (int)grow;  // the original row that ends up at position k
      } else if (act && cur == k) {
        cur = p;  // (p != k here: the row at position k was not the pivot)
      }
      if (w == 0 && tid == 0) s_ipiv[j] = p;
      T uu0 = valid ? u[0] : T(0);
      if (!valid) {  // no row was published for a NaN column in the single-workgroup path either: use row k's own value
        uu0 = T(0);
      }
      const T piv = uu0;
      if (piv == T(0) && w == 0 && tid == 0) atomicOr(info, 1);
      const T rp = piv == T(0) ? T(1) : T(1) / piv;
      if (have && !done) {  // (divergent: finished rows sit this out under the exec mask — no per-element select)
        const T l = a[0] * rp;
        a[0] = l;
#pragma unroll
        for (int c0 = 0; c0 < PB; c0 += 8) {
          if (c0 + 8 <= live) {  // (uniform) a whole chunk of later columns
            T uu[8];
#pragma unroll
            for (int c = 0; c < 8; c++) uu[c] = u[c0 + c];
#pragma unroll
            for (int c = 0; c < 8; c++)
              if (c0 + c >= 1) a[c0 + c] -= l * uu[c];
          } else if (c0 < live) {  // the chunk the boundary runs through
#pragma unroll
            for (int c = 0; c < 8; c++)
              if (c0 + c >= 1 && c0 + c < live) a[c0 + c] -= l * u[c0 + c];
          }
        }
      }
    }
    if constexpr (PROF) { const long long t = wall_clock64(); pt[2] += t - tq; tq = t; }
    // rotate the frame: the finished column goes to the back
    const T t0 = a[0];
#pragma unroll
    for (int c = 0; c + 1 < PB; c++) a[c] = a[c + 1];
    a[PB - 1] = t0;
    if constexpr (PROF) {
      asm volatile("" : "+v"(a[0]), "+v"(a[PB - 1]));
      const long long t = wall_clock64();
      pt[3] += t - tq;
    }
  }
  if constexpr (PROF) {
    if (w == 0 && tid == 0) {
      for (int q = 0; q < 4; q++) prof[q] += pt[q];
      prof[4] += clock64() - c_start;
      prof[5] += wall_clock64() - w_start;
    }
  }
  // every row goes to its position (pivot rows: the column they were chosen for; a displaced row: where the
  // interchanges left it; everything else stays).  All reads of the panel happened before column 0 was agreed on.
  if (have) {
#pragma unroll
    for (int c = 0; c < PB; c++)
      if (c < pw) W[(long long)cur * ld + k0 + c] = a[c];
  }
  if (tid < pw) {
    if (s_top[tid] >= 0) plist[1 + tid] = s_top[tid];
    if (w == 0) ipiv[k0 + tid] = s_ipiv[tid];
  }
  // displaced rows (they started in the top block — workgroup 0 — and were moved out of it without being chosen)
  if (w == 0) {
    if (have && !done && cur != (int)grow) {
      const int e = atomicAdd(&pl_cnt, 1);
      plist[1 + PB + 2 * e] = cur;
      plist[2 + PB + 2 * e] = (int)grow;
    }
    __syncthreads();
    if (tid == 0) plist[0] = pl_cnt;
  }
}

// Interchanges of one panel applied to every column outside it, and U12 = L11^-1 A12 for those to its right:
// one thread per column.  plist (from the panel kernel): [0] number of displaced rows, [1 .. PB] the original
// row that is now at position k0 + i, then (position, original row) pairs.  A column is independent of every
// other: its <= 2 PB touched values are all LOADED (unconditionally — a predicate per load serialises them)
// before any is stored.
template <class T, int PB, int CB>
__global__ __launch_bounds__(CB) void lu_swap_u12_kernel(T* __restrict__ W, long long ld, int n, int k0, int pw,
                                                        const int* __restrict__ plist, long long s0, long long l0,
                                                        long long s1, long long l1) {  // columns [s0, s0+l0) and [s1, s1+l1)
  extern __shared__ __attribute__((aligned(16))) unsigned char lu_smem[];
  T* xs = (T*)lu_smem;          // [PB][CB]: the panel's rows of this column block, then the solution
  T* Ls = xs + PB * CB;         // [PB][PB + 1]: L11 (strictly lower part)
  __shared__ int e_top[PB], e_dst[PB], e_src[PB];
  const int tid = threadIdx.x;
  const long long cc = (long long)blockIdx.x * CB + tid;
  const bool have = cc < l0 + l1;
  const long long c = have ? (cc < l0 ? s0 + cc : s1 + (cc - l0)) : 0;  // (idle threads read column 0: valid memory, results dropped)
  const bool right = have && c >= k0 + pw;
  const int ndisp = plist[0];
  for (int i = tid; i < PB; i += CB) {
    e_top[i] = i < pw ? plist[1 + i] : k0;
    e_dst[i] = i < ndisp ? plist[1 + PB + 2 * i] : k0;
    e_src[i] = i < ndisp ? plist[2 + PB + 2 * i] : k0;
  }
  {
    // L11: all PB * PB / CB loads of a thread in flight before the first LDS store (the looped form waited for each
    // load in turn: ~16 memory latencies in front of everything else in this kernel)
    constexpr int NL = (PB * PB + CB - 1) / CB;
    T lv[NL];
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int e = u * CB + tid, r = e / PB, q = e - r * PB;
      lv[u] = (e < PB * PB && r < pw && q < r) ? W[(long long)(k0 + r) * ld + k0 + q] : T(0);
    }
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int e = u * CB + tid, r = e / PB, q = e - r * PB;
      if (e < PB * PB) Ls[r * (PB + 1) + q] = lv[u];
    }
  }
  __syncthreads();
  T d[PB];
#pragma unroll
  for (int i = 0; i < PB; i++) d[i] = W[(long long)e_src[i] * ld + c];
#pragma unroll 16
  for (int i = 0; i < PB; i++) xs[i * CB + tid] = W[(long long)e_top[i] * ld + c];
  if (have) {
#pragma unroll
    for (int i = 0; i < PB; i++)
      if (i < ndisp) W[(long long)e_dst[i] * ld + c] = d[i];
  }
  // forward substitution, unit diagonal, the column in REGISTERS and fully unrolled: L11 comes from LDS as
  // broadcast loads (rows of a narrower last panel are zero below pw: the extra steps change nothing).
  // (the looped form with the column in LDS waited for eight LDS loads per four FMAs: ~40 of this kernel's 68 us)
  T x[PB];
#pragma unroll
  for (int i = 0; i < PB; i++) x[i] = xs[i * CB + tid];
  if (right) {
#pragma unroll
    for (int r = 1; r < PB; r++) {
      T s0 = x[r], s1 = T(0);
#pragma unroll
      for (int q = 0; q + 1 < r; q += 2) {
        s0 -= Ls[r * (PB + 1) + q] * x[q];
        s1 -= Ls[r * (PB + 1) + q + 1] * x[q + 1];
      }
      if (r & 1) s0 -= Ls[r * (PB + 1) + r - 1] * x[r - 1];
      x[r] = s0 + s1;
    }
  }
  if (have) {
#pragma unroll
    for (int i = 0; i < PB; i++)
      if (i < pw) W[(long long)(k0 + i) * ld + c] = x[i];
  }
}

// One-XCD panels rest on "workgroup id % 8 is the XCD" and "a store without write-through is seen by an agent-scope load
// of a workgroup on the same XCD".  Checked once per process before the first such panel: workgroup 0 publishes a
// record that way, workgroup 8 polls for it (200 us at most).  Not seen (another partition mode, another dispatch
// order): the panels stay spread over the XCDs with write-through records, the round-3/4 form.
__global__ void lu_one_xcd_probe_kernel(unsigned long long* box, int* seen) {
  if (threadIdx.x != 0) return;
  const unsigned long long tag = 0x5eed0f0e1ull;
  if (blockIdx.x == 0) {
    lu_publish_t<true>(box, 0x1234ull, tag);
  } else if (blockIdx.x == 8) {
    const long long t0 = (long long)wall_clock64();
    unsigned long long bits = 0;
    int ok = 0;
    for (;;) {
      if (lu_poll_t(box, bits, tag) && bits == 0x1234ull) { ok = 1; break; }
      if ((long long)wall_clock64() - t0 > 20000) break;
    }
    *seen = ok;
  }
}

// 1: the probe passed; 0: it did not (or could not be run: its result is read back with a stream synchronisation,
// which a capture in progress forbids — asked again on the next call)
int lu_one_xcd_ok(hipStream_t st, bool may_sync) {
  static int known = -1;
  if (known >= 0) return known;
  if (!may_sync) return 0;
  void* d = nullptr;
  if (hipMalloc(&d, 512) != hipSuccess) return 0;  // (not pthip_alloc: a plan's arena must see the same allocations in every pass)
  int seen = 0;
  bool ran = hipMemsetAsync(d, 0, 512, st) == hipSuccess;
  if (ran) {
    hipLaunchKernelGGL(lu_one_xcd_probe_kernel, dim3(16), dim3(64), 0, st, (unsigned long long*)d, (int*)((char*)d + 256));
    ran = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
          hipMemcpy(&seen, (char*)d + 256, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d);
  if (ran) known = seen ? 1 : 0;
  return ran ? known : 0;
}

template <class T, int PB>
int getrf_blocked2(long long n, const T* A, T* LU, long long* perm, T* sign, T* logabs, int flag_singular) {
  constexpr int CB = 64;
  hipStream_t st = pthip::ctx().stream;
  const int dt = sizeof(T) == 8 ? PTHIP_F64 : PTHIP_F32;
  const int nWmax = (int)((n + BLOCK - 1) / BLOCK);
  if (nWmax > pthip::kNumCU)
    return pthip::set_error("pthip_getrf: n = %lld needs %d co-resident panel workgroups (the device has %d CUs)", n, nWmax, pthip::kNumCU);
  const size_t boxbytes = (size_t)PB * nWmax * (PB + 2) * 16, boxkbytes = 0;
  const size_t ibytes = ((size_t)n * sizeof(int) + 255) / 256 * 256;
  const size_t pbytes = ((size_t)2 * (1 + 3 * PB) * sizeof(int) + 255) / 256 * 256;  // two panels' lists (look-ahead)
  void* scratch = nullptr;
  int r = pthip_alloc(boxbytes + boxkbytes + ibytes + pbytes + 256, &scratch);
  if (r) return r;
  auto fail = [&](int rc) { pthip_free(scratch); return rc; };
  unsigned long long* box = (unsigned long long*)scratch;
  int* ipiv = (int*)((char*)scratch + boxbytes + boxkbytes);
  int* plist = (int*)((char*)scratch + boxbytes + boxkbytes + ibytes);
  int* flags = (int*)((char*)scratch + boxbytes + boxkbytes + ibytes + pbytes);  // [0] info, [1] abort; [8..] phase ticks (PROF)
  long long* prof = (long long*)(flags + 8);
  static const bool prof_on = getenv("PTHIP_LU_PROF") != nullptr;
  static unsigned long long calls = 0;
  const unsigned long long nonce = (++calls) * 0xD1B54A32D192ED03ull;
  // (one fill for the whole call: the pair tags carry the column index, a stale pair of an earlier panel never validates)
  if (hipError_t e = pthip::memset_async(scratch, 0, boxbytes + boxkbytes + ibytes + pbytes + 256, st); e != hipSuccess)
    return fail(pthip::check(e, "lu scratch memset"));
  if (hipError_t e = pthip::memcpy_async(LU, A, (size_t)n * n * sizeof(T), hipMemcpyDeviceToDevice, st); e != hipSuccess)
    return fail(pthip::check(e, "lu copy"));
  auto ks = lu_swap_u12_kernel<T, PB, CB>;
  const size_t sh = ((size_t)PB * CB + (size_t)PB * (PB + 1)) * sizeof(T);
  static bool attr = false;
  if (!attr && sh > 48 * 1024) {
    if (hipError_t e = hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); e != hipSuccess)
      return fail(pthip::check(e, "lu_swap_u12 attribute"));
    attr = true;
  }
  // Look-ahead (round 4): the trailing update of panel k — interchanges + U12 of every column outside the NEXT panel,
  // and their GEMM — runs on a second stream while the next panel is being factored; only the next panel's own
  // columns are swapped / solved / updated on the panel stream first.  (panel 125 us, swap 26, GEMM 30 per 32 columns
  // at n = 4096: the 56 us hide behind the panel.)  Off while a plan is being captured or recorded (one stream there)
  // and with PTHIP_LU_LOOKAHEAD=0.  The panel's workgroups wait on each other: they may start late while the GEMM
  // holds the CUs, but GEMM workgroups always finish, so the panel's do get their CUs.
  // MEASURED (profiles/r4r_getrf_lookahead.txt): 24.2 ms against 25.0 at n = 4096, 4.67 against 4.60 at n = 1024 — the
  // panel's cross-workgroup hand-overs slow down by about what the overlap saves once the GEMM loads the memory system.
  // Correct (same pivots, tests green with it on), but not a win: opt-in with PTHIP_LU_LOOKAHEAD=1.
  static const bool la_env = getenv("PTHIP_LU_LOOKAHEAD") && atoi(getenv("PTHIP_LU_LOOKAHEAD")) == 1;
  const char* x1_env = getenv("PTHIP_LU_ONE_XCD");  // (read per call: the tests and the bench time both forms in one process)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  const bool one_xcd = !(x1_env && atoi(x1_env) == 0) && n > BLOCK &&
                       lu_one_xcd_ok(st, cap == hipStreamCaptureStatusNone && pthip::ctx().recorder == nullptr) == 1;
  const int cur = pthip::ctx().current;
  const int side = cur == 3 ? 2 : 3;
  const bool lookahead = la_env && !prof_on && pthip::ctx().recorder == nullptr && cap == hipStreamCaptureStatusNone && n > 4 * PB;
  auto on_stream = [&](int which) { return pthip_stream_select(which); };
  auto launch_swap = [&](const int* pl, long long k0, int pw, long long s0, long long l0, long long s1, long long l1) -> int {
    if (l0 + l1 <= 0) return 0;
    PTHIP_KLAUNCH(ks, dim3((unsigned)((l0 + l1 + CB - 1) / CB)), dim3(CB), sh, pthip::ctx().stream, LU, n, (int)n, (int)k0, pw, pl, s0, l0, s1, l1);
    return pthip::post_launch("lu_swap_u12");
  };
  auto launch_gemm = [&](long long k0, int pw, long long c0, long long nc) -> int {  // A22[:, c0 : c0+nc) -= L21 U12[:, c0 : c0+nc)
    const long long rest = n - k0 - pw;
    if (rest <= 0 || nc <= 0) return 0;
    return pthip::gemm_inplace(dt, rest, nc, pw, -1.0, LU + (k0 + pw) * n + k0, n, 1, LU + k0 * n + c0, n, 1, 1.0, LU + (k0 + pw) * n + c0, n);
  };
  const char* tb_env = getenv("PTHIP_LU_PANEL_THREADS");
  const bool tb512 = PB == 32 && tb_env && atoi(tb_env) == 512;
  const int TBsel = tb512 ? 512 : BLOCK;
  long long panel_index = 0;
  for (long long k0 = 0; k0 < n; k0 += PB, panel_index++) {
    const int pw = (int)((n - k0) < PB ? (n - k0) : PB);
    const int nW = (int)((n - k0 + TBsel - 1) / TBsel);
    int* pl = plist + (panel_index & 1) * (1 + 3 * PB);
    hipStream_t ps = pthip::ctx().stream;
    const bool x1 = one_xcd && nW > 1 && nW <= 32;  // (32 CUs on an XCD: every workgroup of the panel resident there)
#define LU_PANEL2_LAUNCH(PROFV, X1V, TBV)                                                                                            \
  PTHIP_KLAUNCH((lu_panel2_kernel<T, PB, PROFV, X1V, TBV>), dim3((unsigned)((X1V ? 8 : 1) * nW)), dim3(TBV), 0, ps, LU, n, (int)n, (int)k0, nW, \
                box, ipiv, pl, flags, flags + 1, pthip::ctx().status_dev, nonce, prof)
    bool launched = false;
    if constexpr (PB == 32) {
      if (tb512) {
        launched = true;
        if (prof_on && x1) LU_PANEL2_LAUNCH(true, true, 512);
        else if (prof_on) LU_PANEL2_LAUNCH(true, false, 512);
        else if (x1) LU_PANEL2_LAUNCH(false, true, 512);
        else LU_PANEL2_LAUNCH(false, false, 512);
      }
    }
    if (!launched) {
      if (prof_on && x1) LU_PANEL2_LAUNCH(true, true, BLOCK);
      else if (prof_on) LU_PANEL2_LAUNCH(true, false, BLOCK);
      else if (x1) LU_PANEL2_LAUNCH(false, true, BLOCK);
      else LU_PANEL2_LAUNCH(false, false, BLOCK);
    }
#undef LU_PANEL2_LAUNCH
    if ((r = pthip::post_launch("lu_panel2"))) return fail(r);
    const long long right0 = k0 + pw, nright = n - right0;
    if (!lookahead) {
      if ((r = launch_swap(pl, k0, pw, 0, k0, right0, nright))) return fail(r);
      if ((r = launch_gemm(k0, pw, right0, nright))) return fail(r);
      continue;
    }
    // the next panel's columns on this stream; everything else on the side stream, behind the panel
    const long long nnext = nright < PB ? nright : PB;
    if (panel_index > 0)
      if ((r = pthip_stream_wait(cur, side))) return fail(r);  // the previous trailing update touched these columns
    if ((r = launch_swap(pl, k0, pw, right0, nnext, 0, 0))) return fail(r);
    if ((r = launch_gemm(k0, pw, right0, nnext))) return fail(r);
    if ((r = pthip_stream_wait(side, cur))) return fail(r);  // (behind this panel; the two small launches above ride along)
    if ((r = on_stream(side))) return fail(r);
    r = launch_swap(pl, k0, pw, 0, k0, right0 + nnext, nright - nnext);
    if (!r) r = launch_gemm(k0, pw, right0 + nnext, nright - nnext);
    const int r2 = on_stream(cur);
    if (r || r2) return fail(r ? r : r2);
  }
  if (lookahead)
    if ((r = pthip_stream_wait(cur, side))) return fail(r);
  auto kf = lu_finish_kernel<T>;
  const size_t shf = (size_t)n * sizeof(int);
  static bool attrf = false;
  if (!attrf && shf > 48 * 1024) {
    if (hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); e != hipSuccess)
      return fail(pthip::check(e, "lu_finish attribute"));
    attrf = true;
  }
  PTHIP_KLAUNCH(kf, dim3(1), dim3(BLOCK), shf, st, (const T*)LU, n, (int)n, (const int*)ipiv, (const int*)flags, perm, sign, logabs,
                flag_singular ? pthip::ctx().status_dev : (int*)nullptr);
  r = pthip::post_launch("lu_finish");
  if (prof_on && !r) {
    long long h[6];
    if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, prof, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[pthip lu prof] n=%lld PB=%d f%d: per column us  A %.2f  B %.2f  C %.2f  rotate %.2f | loop %.2f us/col, shader clock %.0f MHz\n", n, PB,
              (int)sizeof(T) * 8, h[0] * 0.01 / n, h[1] * 0.01 / n, h[2] * 0.01 / n, h[3] * 0.01 / n, h[5] * 0.01 / n, h[5] ? (double)h[4] / (h[5] * 0.01) : 0.0);
  }
  pthip_free(scratch);
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
  hipStream_t st = pthip::ctx().stream;
  const size_t need = (size_t)n * (size_t)(n | 1) * sizeof(T);
  static const bool unblocked = getenv("PTHIP_LU_UNBLOCKED") != nullptr;
  // (safe mode: the one-workgroup sweep in global memory for ANY n — slow, but it waits for nobody)
  if (need <= 160 * 1024 - 8192 || (unblocked && n <= 512) || pthip::ctx().safe_mode) {
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
  // beyond one CU's LDS: the blocked factorisation, one matrix after the other
  // (PTHIP_LU_PANEL: "v1" = the round-3 unrolled 32-column panel, "32" / "64" = the looped panel of that width)
  static const char* panel_env = getenv("PTHIP_LU_PANEL");
  // default 32: measured 25.0 ms against 26.3 (64) and 30.2 (v1) at n = 4096, 4.6 / 4.9 / 6.2 ms at n = 1024
  // (profiles/r4f_getrf.txt) — the per-column work of a thread grows with the panel width, the launches per panel do not
  static const int panel = panel_env == nullptr ? 32 : (!strcmp(panel_env, "v1") ? 0 : atoi(panel_env));
  for (long long b = 0; b < batch; b++) {
    const T* Ab = (const T*)A + b * n * n;
    T* LUb = (T*)LU + b * n * n;
    long long* pb = (long long*)perm + b * n;
    int r;
    if (panel == 0) r = getrf_blocked<T>(n, Ab, LUb, pb, (T*)sign + b, (T*)logabs + b, flag_singular);
    else if (panel == 32) r = getrf_blocked2<T, 32>(n, Ab, LUb, pb, (T*)sign + b, (T*)logabs + b, flag_singular);
    else r = getrf_blocked2<T, 64>(n, Ab, LUb, pb, (T*)sign + b, (T*)logabs + b, flag_singular);
    if (r) return r;
  }
  return 0;
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
