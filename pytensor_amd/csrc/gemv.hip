// gemv.hip — Gemv / Ger: HBM-bound matrix-vector kernels for gfx950.
//
// Reference: Gemv.perform (pytensor/tensor/blas/gemv.py:64-108, SciPy ?gemv),
// C glue tensor/blas/c_code/codegen.py:542-805; Ger (tensor/blas/ger.py:8).
//   out[i] = beta*y[i] + alpha * sum_j A[i,j] x[j]
// Algorithmic bytes: M*N*itemsize (A read exactly once) — the roofline is HBM.
//
// Two layouts matter (element strides sA0 = row stride, sA1 = column stride):
//   "row" kernel (sA1 == 1): a wave owns a row at a time, lanes stride across the row
//       with 16-byte loads (fully coalesced 1 KiB per wave instruction), x cached in
//       LDS, ROWS rows in flight per wave, cross-lane transpose-reduce at the end.
//   "col" kernel (sA0 == 1, i.e. A is the transpose of a row-major matrix: the
//       X^T r of a gradient): lanes own output elements i, waves split the long
//       reduction over j; per-wave register accumulators → LDS → per-block partials
//       → second tiny launch (deterministic, no atomics).
// Anything else (both strides != 1) takes the generic strided kernel.
#include "common.h"
#include "reduce_device.h"

using namespace pthip_dev;

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

template <class T> struct Vec;
template <> struct Vec<double> { using type = double2; static constexpr int N = 2; };
template <> struct Vec<float> { using type = float4; static constexpr int N = 4; };

template <class T> __device__ __forceinline__ T vget(const typename Vec<T>::type& v, int k);
template <> __device__ __forceinline__ double vget<double>(const double2& v, int k) { return k == 0 ? v.x : v.y; }
template <> __device__ __forceinline__ float vget<float>(const float4& v, int k) {
  return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
}

// The matrix is read exactly once: non-temporal loads (`nt`: the line is not kept in L2/MALL)
// — the generated streaming kernels measured +7 % with them (codegen._stream_load).
template <class V> __device__ __forceinline__ V nt_load16(const V* p) {
  typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
  static_assert(sizeof(V) == 16, "16-byte pack");
  const u4_t r = __builtin_nontemporal_load((const u4_t*)p);
  V o;
  __builtin_memcpy(&o, &r, 16);
  return o;
}

// ---- row kernel ------------------------------------------------------------------------
// VEC: 16-byte vector loads (requires N % VN == 0, sA0 % VN == 0, aligned base).
// x is staged in LDS when it fits (XLDS), else read through L1/L2.
// ROWS rows in flight per wave (4 when there are plenty of rows, 1 for short-and-wide matrices)

template <class T, bool VEC, bool XLDS, int ROWS>
__global__ __launch_bounds__(BLOCK) void gemv_row_kernel(
    T* __restrict__ out, const T* __restrict__ A, const T* __restrict__ x,
    const T* __restrict__ y, long long M, long long N, long long sA0, long long sx, long long sy,
    T alpha, T beta) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* xs = (T*)smem_raw;
  constexpr int VN = Vec<T>::N;
  using V = typename Vec<T>::type;
  if constexpr (XLDS) {
    for (long long j = threadIdx.x; j < N; j += BLOCK) xs[j] = x[j * sx];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * WAVES;
  for (long long row0 = wave * ROWS; row0 < M; row0 += nwaves * ROWS) {
    T acc[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) acc[r] = T(0);
    if constexpr (VEC) {
      // (forcing 8 loads in flight per lane with an unroll pragma measured 2 us slower at
      //  4096 x 4096: the compiler's own schedule already overlaps the row passes)
      for (long long j = (long long)lane * VN; j < N; j += 64 * VN) {
        V xv;
        if constexpr (XLDS) xv = *(const V*)(xs + j);
        else {
          // x contiguous is required for VEC && !XLDS (checked on the host)
          xv = *(const V*)(x + j);
        }
        V av[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          const long long row = row0 + r < M ? row0 + r : M - 1;
          av[r] = nt_load16((const V*)(A + row * sA0 + j));
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++)
#pragma unroll
          for (int k = 0; k < VN; k++) acc[r] += vget<T>(av[r], k) * vget<T>(xv, k);
      }
    } else {
      for (long long j = lane; j < N; j += 64) {
        const T xv = XLDS ? xs[j] : x[j * sx];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
          const long long row = row0 + r < M ? row0 + r : M - 1;
          acc[r] += __builtin_nontemporal_load(A + row * sA0 + j) * xv;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) acc[r] = wave_reduce<OpAdd>(acc[r]);
    if (lane < ROWS) {
      T v = acc[0];
#pragma unroll
      for (int r = 1; r < ROWS; r++) v = (lane == r) ? acc[r] : v;
      const long long row = row0 + lane;
      if (row < M) {
        T res = alpha * v;
        if (beta != T(0)) res += beta * y[row * sy];
        out[row] = res;
      }
    }
  }
}

// ---- col kernel ------------------------------------------------------------------------
// A[i,j] at A + i + j*sA1 (sA0 == 1).  Lane owns VN consecutive outputs i; a wave walks j.
// grid = (nsplit, ceil(M / (64*VN))); partials [nsplit][M].
template <class T, bool VEC>
__global__ __launch_bounds__(BLOCK) void gemv_col_kernel(
    T* __restrict__ part, const T* __restrict__ A, const T* __restrict__ x, long long M,
    long long N, long long sA1, long long sx, long long chunk) {
  constexpr int VN = VEC ? Vec<T>::N : 1;
  using V = typename Vec<T>::type;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long long i0 = ((long long)blockIdx.y * 64 + lane) * VN;
  const long long jb0 = (long long)blockIdx.x * chunk;
  long long jb1 = jb0 + chunk;
  if (jb1 > N) jb1 = N;
  T acc[VN];
#pragma unroll
  for (int k = 0; k < VN; k++) acc[k] = T(0);
  const bool in = i0 < M;
  // waves interleave blocks of 64 columns j
  for (long long j0 = jb0 + (long long)wid * 64; j0 < jb1; j0 += WAVES * 64) {
    const long long jl = j0 + lane;
    const T xl = jl < jb1 ? x[jl * sx] : T(0);  // lane l holds x[j0 + l]
    const int m = (int)((jb1 - j0) < 64 ? (jb1 - j0) : 64);
    if (m == 64) {
#pragma unroll 8
      for (int jj = 0; jj < 64; jj++) {
        const T xj = __shfl(xl, jj, 64);  // wave-uniform broadcast (v_readlane)
        if (in) {
          if constexpr (VEC) {
            V av = nt_load16((const V*)(A + i0 + (j0 + jj) * sA1));
#pragma unroll
            for (int k = 0; k < VN; k++) acc[k] += vget<T>(av, k) * xj;
          } else {
            acc[0] += __builtin_nontemporal_load(A + i0 + (j0 + jj) * sA1) * xj;
          }
        }
      }
    } else {
      for (int jj = 0; jj < m; jj++) {
        const T xj = __shfl(xl, jj, 64);
        if (in) {
          if constexpr (VEC) {
            V av = nt_load16((const V*)(A + i0 + (j0 + jj) * sA1));
#pragma unroll
            for (int k = 0; k < VN; k++) acc[k] += vget<T>(av, k) * xj;
          } else {
            acc[0] += __builtin_nontemporal_load(A + i0 + (j0 + jj) * sA1) * xj;
          }
        }
      }
    }
  }
  // combine the block's waves in LDS in a fixed order, one partial row per block
  __shared__ T red[WAVES][64 * Vec<T>::N];
#pragma unroll
  for (int k = 0; k < VN; k++) red[wid][lane * VN + k] = acc[k];
  __syncthreads();
  if (wid == 0 && in) {
    T* dst = part + (long long)blockIdx.x * M + i0;
#pragma unroll
    for (int k = 0; k < VN; k++) {
      T v = red[0][lane * VN + k];
#pragma unroll
      for (int w = 1; w < WAVES; w++) v += red[w][lane * VN + k];
      if (i0 + k < M) dst[k] = v;
    }
  }
}

// out[i] = beta*y[i] + alpha * sum_p part[p][i]   (fixed order → deterministic)
// Block = 4 outputs x 64 slices of the partial rows (M/4 workgroups: with M = 128 and 2048
// partial rows every thread sums 32 values), slices combined through LDS in slice order.
template <class T>
__global__ __launch_bounds__(BLOCK) void gemv_finish_kernel(T* __restrict__ out,
                                                           const T* __restrict__ part,
                                                           const T* __restrict__ y, long long M,
                                                           long long nparts, long long sy, T alpha,
                                                           T beta) {
  constexpr int NO = 4, NS = BLOCK / NO;
  __shared__ T red[NS][NO + 1];
  const int oi = threadIdx.x & (NO - 1), sl = threadIdx.x / NO;
  const long long i = (long long)blockIdx.x * NO + oi;
  const long long per = (nparts + NS - 1) / NS;
  long long p0 = sl * per, p1 = p0 + per;
  if (p1 > nparts) p1 = nparts;
  T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
  if (i < M) {
    long long p = p0;
    for (; p + 3 < p1; p += 4) {
      a0 += part[p * M + i];
      a1 += part[(p + 1) * M + i];
      a2 += part[(p + 2) * M + i];
      a3 += part[(p + 3) * M + i];
    }
    for (; p < p1; p++) a0 += part[p * M + i];
  }
  red[sl][oi] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && i < M) {
    T v = red[0][oi];
#pragma unroll 8
    for (int k = 1; k < NS; k++) v += red[k][oi];
    T res = alpha * v;
    if (beta != T(0)) res += beta * y[i * sy];
    out[i] = res;
  }
}

// ---- generic strided fallback: wave per row, scalar loads ----------------------------------
template <class T>
__global__ __launch_bounds__(BLOCK) void gemv_generic_kernel(
    T* __restrict__ out, const T* __restrict__ A, const T* __restrict__ x,
    const T* __restrict__ y, long long M, long long N, long long sA0, long long sA1, long long sx,
    long long sy, T alpha, T beta) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * WAVES + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * WAVES;
  for (long long row = wave; row < M; row += nwaves) {
    T acc = T(0);
    for (long long j = lane; j < N; j += 64) acc += A[row * sA0 + j * sA1] * x[j * sx];
    acc = wave_reduce<OpAdd>(acc);
    if (lane == 0) {
      T res = alpha * acc;
      if (beta != T(0)) res += beta * y[row * sy];
      out[row] = res;
    }
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void ger_kernel(T* __restrict__ out, const T* __restrict__ A,
                                                   const T* __restrict__ x,
                                                   const T* __restrict__ y, long long M,
                                                   long long N, long long sA0, long long sA1,
                                                   long long sx, long long sy, T alpha) {
  const long long n = M * N;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * BLOCK) {
    const long long r = i / N, c = i - r * N;
    out[i] = A[r * sA0 + c * sA1] + alpha * x[r * sx] * y[c * sy];
  }
}

struct ColPlan {
  long long nsplit, chunk, ytiles;
  bool vec;
};

template <class T>
ColPlan col_plan(long long M, long long N, long long sA1, const void* A) {
  ColPlan p;
  constexpr int VN = Vec<T>::N;
  p.vec = (M % VN == 0) && (sA1 % VN == 0) && (((uintptr_t)A) % 16 == 0);
  const int vn = p.vec ? VN : 1;
  p.ytiles = (M + 64 * vn - 1) / (64 * vn);
  long long want = ((long long)pthip::kNumCU * 8 + p.ytiles - 1) / p.ytiles;
  long long max_split = (N + WAVES * 64 - 1) / (WAVES * 64);
  p.nsplit = want < max_split ? want : max_split;
  if (p.nsplit < 1) p.nsplit = 1;
  p.chunk = (N + p.nsplit - 1) / p.nsplit;
  p.chunk = (p.chunk + 63) / 64 * 64;
  p.nsplit = (N + p.chunk - 1) / p.chunk;
  if (p.nsplit < 1) p.nsplit = 1;
  return p;
}

template <class T>
int gemv_typed(long long M, long long N, double alpha_d, const void* Av, long long sA0,
               long long sA1, const void* xv, long long sx, double beta_d, const void* yv,
               long long sy, void* outv, void* ws, size_t ws_bytes) {
  hipStream_t st = pthip::ctx().stream;
  const T* A = (const T*)Av;
  const T* x = (const T*)xv;
  const T* y = (const T*)yv;
  T* out = (T*)outv;
  const T alpha = (T)alpha_d, beta = (T)beta_d;
  if (M == 0) return 0;
  constexpr int VN = Vec<T>::N;
  if (N == 0 || sA1 == 1 || (sA0 != 1 && N == 1)) {
    if (sA1 != 1 && N > 1) return pthip::set_error("gemv: internal layout error");
    const bool vec = (N % VN == 0) && (sA0 % VN == 0) && (((uintptr_t)A) % 16 == 0);
    // tuning knobs for A/B runs (tools/bench_gemv.py): PTHIP_GEMV_XLDS=0 reads x through L1/L2
    // instead of staging it in LDS, PTHIP_GEMV_ROWS=1|2|4 fixes the rows in flight per wave
    static const int knob_xlds = getenv("PTHIP_GEMV_XLDS") ? atoi(getenv("PTHIP_GEMV_XLDS")) : -1;
    static const int knob_rows = getenv("PTHIP_GEMV_ROWS") ? atoi(getenv("PTHIP_GEMV_ROWS")) : 0;
    // x in LDS only while staging it is cheap next to the rows a workgroup streams (x <= 4 KB:
    // the tall-skinny case, +2.6 % at 1e6 x 128); a longer x costs every workgroup a 32 KB copy and
    // a barrier before its first matrix load — read through L1/L2 instead (the matrix loads are
    // non-temporal and leave it cached): 4096^2 f64 24.0 -> 22.2 us, 8192^2 f32 54.1 -> 45.2 us
    // (profiles/r2k_gemv_sweep.txt)
    bool xlds = (size_t)N * sizeof(T) <= 64 * 1024;
    const bool x_direct_ok = sx == 1 && ((uintptr_t)x) % 16 == 0;
    if (x_direct_ok && (size_t)N * sizeof(T) > 4 * 1024 && knob_xlds != 1) xlds = false;
    if (knob_xlds == 0 && x_direct_ok) xlds = false;
    const bool vec_ok = vec && (xlds || (sx == 1 && ((uintptr_t)x) % 16 == 0));
    // plenty of rows: 4 rows in flight per wave; short-and-wide (e.g. 4096 x 4096): one row
    // per wave so that >= 2048 workgroups exist to fill 256 CUs
    int rows = (M / (4 * WAVES) >= (long long)pthip::kNumCU * 4) ? 4 : 1;
    if (knob_rows == 1 || knob_rows == 2 || knob_rows == 4) rows = knob_rows;
    long long waves = (M + rows - 1) / rows;
    long long blocks = (waves + WAVES - 1) / WAVES;
    long long cap = (long long)pthip::kNumCU * 8;
    if (blocks > cap) blocks = cap;
    size_t shmem = xlds ? (size_t)N * sizeof(T) : 0;
#define LAUNCH(V, X)                                                                              \
  do {                                                                                            \
    if (rows == 4)                                                                                \
      PTHIP_KLAUNCH((gemv_row_kernel<T, V, X, 4>), dim3((unsigned)blocks), dim3(BLOCK),      \
                         shmem, st, out, A, x, y, M, N, sA0, sx, sy, alpha, beta);                \
    else if (rows == 2)                                                                           \
      PTHIP_KLAUNCH((gemv_row_kernel<T, V, X, 2>), dim3((unsigned)blocks), dim3(BLOCK),      \
                         shmem, st, out, A, x, y, M, N, sA0, sx, sy, alpha, beta);                \
    else                                                                                          \
      PTHIP_KLAUNCH((gemv_row_kernel<T, V, X, 1>), dim3((unsigned)blocks), dim3(BLOCK),      \
                         shmem, st, out, A, x, y, M, N, sA0, sx, sy, alpha, beta);                \
  } while (0)
    if (vec_ok && xlds) LAUNCH(true, true);
    else if (vec_ok) LAUNCH(true, false);
    else if (xlds) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    return pthip::post_launch("gemv_row");
  }
  if (sA0 == 1) {
    ColPlan p = col_plan<T>(M, N, sA1, Av);
    const long long nparts = p.nsplit;
    if (ws_bytes < (size_t)nparts * (size_t)M * sizeof(T))
      return pthip::set_error("pthip_gemv: workspace too small (%zu < %zu)", ws_bytes,
                              (size_t)nparts * (size_t)M * sizeof(T));
    T* part = (T*)ws;
    if (p.vec)
      PTHIP_KLAUNCH((gemv_col_kernel<T, true>), dim3((unsigned)p.nsplit, (unsigned)p.ytiles),
                         dim3(BLOCK), 0, st, part, A, x, M, N, sA1, sx, p.chunk);
    else
      PTHIP_KLAUNCH((gemv_col_kernel<T, false>), dim3((unsigned)p.nsplit, (unsigned)p.ytiles),
                         dim3(BLOCK), 0, st, part, A, x, M, N, sA1, sx, p.chunk);
    int r = pthip::post_launch("gemv_col");
    if (r) return r;
    PTHIP_KLAUNCH((gemv_finish_kernel<T>), dim3((unsigned)((M + 3) / 4)),
                       dim3(BLOCK), 0, st, out, part, y, M, nparts, sy, alpha, beta);
    return pthip::post_launch("gemv_finish");
  }
  long long blocks = (M + WAVES - 1) / WAVES;
  long long cap = (long long)pthip::kNumCU * 8;
  if (blocks > cap) blocks = cap;
  PTHIP_KLAUNCH((gemv_generic_kernel<T>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, out, A,
                     x, y, M, N, sA0, sA1, sx, sy, alpha, beta);
  return pthip::post_launch("gemv_generic");
}

}  // namespace

extern "C" {

size_t pthip_gemv_workspace(int dtype, int64_t M, int64_t N, int64_t sA0, int64_t sA1) {
  if (sA1 == 1 || sA0 != 1 || M == 0 || N == 0) return 0;
  // bound valid for both the vector and scalar column plans
  long long ytiles_min = 1;
  long long want = (long long)pthip::kNumCU * 8 / ytiles_min;
  long long max_split = (N + WAVES * 64 - 1) / (WAVES * 64);
  long long ns = want < max_split ? want : max_split;
  if (ns < 1) ns = 1;
  return (size_t)(ns + 1) * (size_t)M * (size_t)pthip::dtype_size(dtype);
}

int pthip_gemv(int dtype, int64_t M, int64_t N, double alpha, const void* A, int64_t sA0,
               int64_t sA1, const void* x, int64_t sx, double beta, const void* y, int64_t sy,
               void* out, void* ws, size_t ws_bytes) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64)
    return gemv_typed<double>(M, N, alpha, A, sA0, sA1, x, sx, beta, y, sy, out, ws, ws_bytes);
  if (dtype == PTHIP_F32)
    return gemv_typed<float>(M, N, alpha, A, sA0, sA1, x, sx, beta, y, sy, out, ws, ws_bytes);
  return pthip::set_error("pthip_gemv: dtype %d not supported (float32/float64 only)", dtype);
}

int pthip_gemv_finish(int dtype, int64_t M, int64_t nparts, const void* part, double alpha,
                      double beta, const void* y, int64_t sy, void* out) {
  PTHIP_REQUIRE_INIT();
  hipStream_t st = pthip::ctx().stream;
  if (M == 0) return 0;
  if (dtype == PTHIP_F64)
    PTHIP_KLAUNCH((gemv_finish_kernel<double>), dim3((unsigned)((M + 3) / 4)), dim3(BLOCK), 0,
                       st, (double*)out, (const double*)part, (const double*)y, (long long)M,
                       (long long)nparts, (long long)sy, alpha, beta);
  else if (dtype == PTHIP_F32)
    PTHIP_KLAUNCH((gemv_finish_kernel<float>), dim3((unsigned)((M + 3) / 4)), dim3(BLOCK), 0,
                       st, (float*)out, (const float*)part, (const float*)y, (long long)M,
                       (long long)nparts, (long long)sy, (float)alpha, (float)beta);
  else
    return pthip::set_error("pthip_gemv_finish: dtype %d not supported", dtype);
  return pthip::post_launch("gemv_finish");
}

int pthip_ger(int dtype, int64_t M, int64_t N, double alpha, const void* A, int64_t sA0,
              int64_t sA1, const void* x, int64_t sx, const void* y, int64_t sy, void* out) {
  PTHIP_REQUIRE_INIT();
  hipStream_t st = pthip::ctx().stream;
  long long n = M * N;
  if (n == 0) return 0;
  long long blocks = (n + BLOCK - 1) / BLOCK;
  long long cap = (long long)pthip::kNumCU * 8;
  if (blocks > cap) blocks = cap;
  if (dtype == PTHIP_F64)
    PTHIP_KLAUNCH((ger_kernel<double>), dim3((unsigned)blocks), dim3(BLOCK), 0, st,
                       (double*)out, (const double*)A, (const double*)x, (const double*)y,
                       (long long)M, (long long)N, (long long)sA0, (long long)sA1, (long long)sx,
                       (long long)sy, alpha);
  else if (dtype == PTHIP_F32)
    PTHIP_KLAUNCH((ger_kernel<float>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, (float*)out,
                       (const float*)A, (const float*)x, (const float*)y, (long long)M,
                       (long long)N, (long long)sA0, (long long)sA1, (long long)sx, (long long)sy,
                       (float)alpha);
  else
    return pthip::set_error("pthip_ger: dtype %d not supported", dtype);
  return pthip::post_launch("ger");
}

}  // extern "C"
