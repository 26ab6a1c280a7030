// softmax.hip — Softmax / LogSoftmax over the last axis of a (rows x cols) matrix.
//
// Reference: pytensor/tensor/special.py Softmax 26 (build_inner_graph 44-47:
// e = exp(x - max(x)); e / sum(e)), LogSoftmax 67 (85-87: (x - max) - log(sum(exp(x - max)))).
// The reference inlines these into Max + Composite + Sum + Composite nodes before its fusion
// pass (rewriting/ofg.py:46-70) — five launches and four reads + two writes of the matrix on
// a GPU.  The hip linker keeps the op whole (like the JAX/PyTorch/MLX linkers, which dispatch
// their own softmax) and runs one kernel: the row is read from HBM once, the second and third
// sweep hit L1/L2, one write.
//
// wave-per-row: 64 lanes stride along the row (coalesced), wave64 butterflies for max and sum;
// four rows per workgroup.  Rows of <= 16 elements: thread per row, the workgroup's rows staged through LDS.
// float32 sums accumulate in double, as the reference's Sum does (elemwise.py:1383-1417).
#include "common.h"
#include "reduce_device.h"

namespace {

constexpr int BLOCK = 256;

template <class T> struct Acc { typedef T type; };
template <> struct Acc<float> { typedef double type; };

template <class T> __device__ __forceinline__ T dev_exp(T x);
template <> __device__ __forceinline__ double dev_exp(double x) { return exp(x); }
template <> __device__ __forceinline__ float dev_exp(float x) { return expf(x); }
template <class T> __device__ __forceinline__ T dev_log(T x);
template <> __device__ __forceinline__ double dev_log(double x) { return log(x); }
template <> __device__ __forceinline__ float dev_log(float x) { return logf(x); }

// NaN-propagating max like the reference's Maximum (scalar/basic.py:1744)
template <class T> __device__ __forceinline__ T nan_max(T a, T b) {
  return (b > a) ? b : ((a >= b) ? a : (T)__builtin_nan(""));
}

template <class T, bool LOG>
__global__ __launch_bounds__(BLOCK) void softmax_wave_kernel(T* __restrict__ out,
                                                            const T* __restrict__ x,
                                                            long long rows, long long cols) {
  typedef typename Acc<T>::type A;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
  for (long long r = wave; r < rows; r += nwaves) {
    const T* xr = x + r * cols;
    T* orow = out + r * cols;
    T m = -__builtin_huge_val();
    for (long long j = lane; j < cols; j += 64) m = nan_max(m, xr[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_xor(m, o));
    A s = A(0);
    for (long long j = lane; j < cols; j += 64) s += (A)dev_exp<T>(xr[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (LOG) {
      const T ls = dev_log<T>((T)s);
      for (long long j = lane; j < cols; j += 64) orow[j] = (xr[j] - m) - ls;
    } else {
      const T inv = T(1) / (T)s;
      for (long long j = lane; j < cols; j += 64) orow[j] = dev_exp<T>(xr[j] - m) * inv;
    }
  }
}

// rows of up to 64*VPL*V elements: the row lives in registers (VPL packs of V elements per lane, 16-byte
// loads and stores when V*sizeof(T) == 16) — one read, one exp per element, one write.  The quotient
// e / sum is e * (1 / sum): one division per row instead of ~30 fp64 instructions per element (at
// 8192 x 2048 fp64 the divisions alone were 11 us of VALU issue next to a 33 us HBM floor); the product
// is within 1.5 ulp of the reference's quotient (special.py:44-47).
template <class T, int V> struct __attribute__((aligned(sizeof(T) * V))) sm_pack { T v[V]; };

template <class T, bool LOG, int VPL, int V>
__global__ __launch_bounds__(BLOCK) void softmax_wave_reg_kernel(T* __restrict__ out,
                                                                const T* __restrict__ x,
                                                                long long rows, int cols) {
  typedef typename Acc<T>::type A;
  typedef sm_pack<T, V> P;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
  for (long long r = wave; r < rows; r += nwaves) {
    const T* xr = x + r * cols;
    T* orow = out + r * cols;
    P v[VPL];
    T m = -__builtin_huge_val();
#pragma unroll
    for (int u = 0; u < VPL; u++) {
      const int j = (lane + 64 * u) * V;
      if (j < cols) v[u] = *reinterpret_cast<const P*>(xr + j);
    }
#pragma unroll
    for (int u = 0; u < VPL; u++) {
      const int j = (lane + 64 * u) * V;
      if (j < cols) {
#pragma unroll
        for (int e = 0; e < V; e++) m = nan_max(m, v[u].v[e]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_xor(m, o));
    A s = A(0);
#pragma unroll
    for (int u = 0; u < VPL; u++) {
      const int j = (lane + 64 * u) * V;
      if (j < cols) {
#pragma unroll
        for (int e = 0; e < V; e++) {
          const T d = v[u].v[e] - m;
          const T ex = dev_exp<T>(d);
          s += (A)ex;
          v[u].v[e] = LOG ? d : ex;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const T st = (T)s;
    const T ls = LOG ? dev_log<T>(st) : T(0);
    const T inv = T(1) / st;
#pragma unroll
    for (int u = 0; u < VPL; u++) {
      const int j = (lane + 64 * u) * V;
      if (j < cols) {
#pragma unroll
        for (int e = 0; e < V; e++) v[u].v[e] = LOG ? (v[u].v[e] - ls) : (v[u].v[e] * inv);
        *reinterpret_cast<P*>(orow + j) = v[u];
      }
    }
  }
}

// few, long rows: a whole workgroup per row (a wave per row would leave most of the chip idle)
template <class T, bool LOG>
__global__ __launch_bounds__(BLOCK) void softmax_block_kernel(T* __restrict__ out,
                                                             const T* __restrict__ x,
                                                             long long rows, long long cols) {
  typedef typename Acc<T>::type A;
  __shared__ T smt[BLOCK / 64];
  __shared__ A sma[BLOCK / 64];
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const T* xr = x + r * cols;
    T* orow = out + r * cols;
    T m = -__builtin_huge_val();
    for (long long j = threadIdx.x; j < cols; j += BLOCK) m = nan_max(m, xr[j]);
    m = pthip_dev::block_reduce<pthip_dev::OpMax, T, BLOCK, true>(m, smt);
    A s = A(0);
    for (long long j = threadIdx.x; j < cols; j += BLOCK) s += (A)dev_exp<T>(xr[j] - m);
    s = pthip_dev::block_reduce<pthip_dev::OpAdd, A, BLOCK, true>(s, sma);
    if (LOG) {
      const T ls = dev_log<T>((T)s);
      for (long long j = threadIdx.x; j < cols; j += BLOCK) orow[j] = (xr[j] - m) - ls;
    } else {
      const T inv = T(1) / (T)s;
      for (long long j = threadIdx.x; j < cols; j += BLOCK) orow[j] = dev_exp<T>(xr[j] - m) * inv;
    }
  }
}

// rows of <= MAXC elements: thread per row — but a lane's row is cols*sizeof(T) bytes away from its neighbour's,
// so the workgroup's 256 consecutive rows (one contiguous chunk of memory) are staged through LDS: coalesced
// loads in, a row per thread out of LDS (odd pitch: conflict-free), results back the same way.
// (1e6 x 10 fp64: 87 us with strided per-lane rows -> profiles/r5*_hotpath*.md)
template <class T, bool LOG, int MAXC>
__global__ __launch_bounds__(BLOCK) void softmax_small_kernel(T* __restrict__ out,
                                                             const T* __restrict__ x,
                                                             long long rows, int cols) {
  typedef typename Acc<T>::type A;
  __shared__ T tile[BLOCK * (MAXC + 1)];
  const int pitch = cols | 1;
  const long long r0 = (long long)blockIdx.x * BLOCK;
  const long long left = rows - r0;
  const int nr = left < BLOCK ? (int)left : BLOCK;
  const int n = nr * cols;
  const T* src = x + r0 * cols;
  T* dst = out + r0 * cols;
#pragma unroll 4
  for (int idx = threadIdx.x; idx < n; idx += BLOCK) {
    const int r = idx / cols, j = idx - r * cols;
    tile[r * pitch + j] = src[idx];
  }
  __syncthreads();
  if ((int)threadIdx.x < nr) {
    T* row = tile + threadIdx.x * pitch;
    T v[MAXC];
    T m = -__builtin_huge_val();
#pragma unroll
    for (int j = 0; j < MAXC; j++)
      if (j < cols) { v[j] = row[j]; m = nan_max(m, v[j]); }
    A s = A(0);
#pragma unroll
    for (int j = 0; j < MAXC; j++)
      if (j < cols) {
        const T d = v[j] - m;
        const T ex = dev_exp<T>(d);
        s += (A)ex;
        v[j] = LOG ? d : ex;
      }
    const T st = (T)s;
    const T ls = LOG ? dev_log<T>(st) : T(0);
    const T inv = T(1) / st;
#pragma unroll
    for (int j = 0; j < MAXC; j++)
      if (j < cols) row[j] = LOG ? (v[j] - ls) : (v[j] * inv);
  }
  __syncthreads();
#pragma unroll 4
  for (int idx = threadIdx.x; idx < n; idx += BLOCK) {
    const int r = idx / cols, j = idx - r * cols;
    dst[idx] = tile[r * pitch + j];
  }
}

template <class T>
int softmax_typed(int log_, long long rows, long long cols, const void* x, void* out) {
  if (rows == 0 || cols == 0) return 0;
  hipStream_t st = pthip::ctx().stream;
  if (cols <= 16) {
    const unsigned grid = (unsigned)((rows + BLOCK - 1) / BLOCK);
    if (log_)
      PTHIP_KLAUNCH((softmax_small_kernel<T, true, 16>), dim3(grid), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (int)cols);
    else
      PTHIP_KLAUNCH((softmax_small_kernel<T, false, 16>), dim3(grid), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (int)cols);
    return pthip::post_launch("softmax(thread per row, LDS-staged)");
  }
  if (rows < 2 * (long long)pthip::kNumCU && cols >= 4096) {
    const unsigned grid = (unsigned)rows;
    if (log_)
      PTHIP_KLAUNCH((softmax_block_kernel<T, true>), dim3(grid), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, cols);
    else
      PTHIP_KLAUNCH((softmax_block_kernel<T, false>), dim3(grid), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, cols);
    return pthip::post_launch("softmax(workgroup per row)");
  }
  long long blocks = (rows + 3) / 4;
  const long long cap = (long long)pthip::kNumCU * 16;
  if (blocks > cap) blocks = cap;
#define LAUNCH_REG(VPL, V)                                                                       \
  do {                                                                                           \
    if (log_)                                                                                    \
      PTHIP_KLAUNCH((softmax_wave_reg_kernel<T, true, VPL, V>), dim3((unsigned)blocks),     \
                         dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (int)cols);             \
    else                                                                                         \
      PTHIP_KLAUNCH((softmax_wave_reg_kernel<T, false, VPL, V>), dim3((unsigned)blocks),    \
                         dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (int)cols);             \
    return pthip::post_launch("softmax(wave per row, registers)");                               \
  } while (0)
  constexpr int VW = 16 / (int)sizeof(T);  // elements of a 16-byte pack
  const bool packs = cols % VW == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (packs) {
    if (cols <= 64 * 2 * VW) LAUNCH_REG(2, VW);
    if (cols <= 64 * 8 * VW) LAUNCH_REG(8, VW);
    if (cols <= 64 * 16 * VW) LAUNCH_REG(16, VW);
  }
  if (cols <= 64 * 4) LAUNCH_REG(4, 1);
  if (cols <= 64 * 16) LAUNCH_REG(16, 1);
  if (cols <= 64 * 32) LAUNCH_REG(32, 1);
#undef LAUNCH_REG
  if (log_)
    PTHIP_KLAUNCH((softmax_wave_kernel<T, true>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, cols);
  else
    PTHIP_KLAUNCH((softmax_wave_kernel<T, false>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, cols);
  return pthip::post_launch("softmax(wave per row)");
}

}  // namespace

extern "C" int pthip_softmax(int dtype, int log_, int64_t rows, int64_t cols, const void* x, void* out) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64) return softmax_typed<double>(log_, rows, cols, x, out);
  if (dtype == PTHIP_F32) return softmax_typed<float>(log_, rows, cols, x, out);
  return pthip::set_error("pthip_softmax: dtype %d not supported (float32/float64 only)", dtype);
}
