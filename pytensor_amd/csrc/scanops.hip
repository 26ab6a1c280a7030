// scanops.hip — cumulative sum/product along an axis, argmax over trailing axes, integer dot.
//
// Reference: CumOp.perform (pytensor/tensor/extra_ops.py: np.cumsum / np.cumprod along
// `axis`), Argmax.perform (pytensor/tensor/math.py: reduced axes moved last, flattened,
// np.argmax -> index of the FIRST maximum, a NaN counts as the maximum).
//
// Both are order-defined: np.cumsum accumulates strictly left to right, so a work-efficient
// parallel scan would round differently from the reference.  One lane owns one line and walks
// it sequentially; lanes run along the contiguous inner axis, so every step of the walk is one
// coalesced row of loads/stores.  Lines with inner == 1 (cumsum over the last axis) are walked
// by one lane each — latency-bound by design, these are small on the hot path.
#include "common.h"

#include <type_traits>

namespace {

constexpr int BLOCK = 256;

template <class T, bool MUL>
__global__ __launch_bounds__(BLOCK) void cumulative_kernel(T* __restrict__ dst,
                                                          const T* __restrict__ src,
                                                          long long outer, long long n,
                                                          long long inner) {
  const long long line = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (line >= outer * inner) return;
  const long long o = line / inner, i = line - o * inner;
  const T* s = src + o * n * inner + i;
  T* d = dst + o * n * inner + i;
  T acc = MUL ? T(1) : T(0);
  for (long long k = 0; k < n; k++) {
    const T v = s[k * inner];
    acc = MUL ? (T)(acc * v) : (T)(acc + v);
    d[k * inner] = acc;
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void argmax_kernel(long long* __restrict__ out,
                                                      const T* __restrict__ src, long long rows,
                                                      long long R) {
  const long long row = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (row >= rows) return;
  const T* s = src + row * R;
  T best = s[0];
  long long bi = 0;
  bool nan_found = best != best;
  for (long long k = 1; k < R && !nan_found; k++) {
    const T v = s[k];
    if (v != v) { bi = k; nan_found = true; }
    else if (v > best) { best = v; bi = k; }
  }
  out[row] = bi;
}

// integer matrix product (Dot.perform = np.dot on integer arrays: no BLAS, wrap-around
// arithmetic in the result dtype).  Bit-exact tier: one thread per output element, k ascending.
template <class T>
__global__ __launch_bounds__(BLOCK) void imatmul_kernel(T* __restrict__ out,
                                                       const T* __restrict__ A, long long sA0,
                                                       long long sA1, const T* __restrict__ B,
                                                       long long sB0, long long sB1, long long M,
                                                       long long N, long long K) {
  const long long e = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (e >= M * N) return;
  const long long i = e / N, j = e - i * N;
  typedef typename std::make_unsigned<T>::type U;  // defined overflow
  U acc = 0;
  for (long long k = 0; k < K; k++) acc += (U)A[i * sA0 + k * sA1] * (U)B[k * sB0 + j * sB1];
  out[e] = (T)acc;
}

template <class T>
int imatmul_typed(long long M, long long N, long long K, const void* A, long long sA0,
                  long long sA1, const void* B, long long sB0, long long sB1, void* out) {
  if (M * N == 0) return 0;
  const unsigned grid = (unsigned)((M * N + BLOCK - 1) / BLOCK);
  PTHIP_KLAUNCH((imatmul_kernel<T>), dim3(grid), dim3(BLOCK), 0, pthip::ctx().stream, (T*)out,
                     (const T*)A, sA0, sA1, (const T*)B, sB0, sB1, M, N, K);
  return pthip::post_launch("imatmul");
}

template <class T>
int cumulative_typed(int mul, long long outer, long long n, long long inner, const void* src,
                     void* dst) {
  const long long lines = outer * inner;
  if (lines == 0 || n == 0) return 0;
  const unsigned grid = (unsigned)((lines + BLOCK - 1) / BLOCK);
  hipStream_t st = pthip::ctx().stream;
  if (mul)
    PTHIP_KLAUNCH((cumulative_kernel<T, true>), dim3(grid), dim3(BLOCK), 0, st, (T*)dst,
                       (const T*)src, outer, n, inner);
  else
    PTHIP_KLAUNCH((cumulative_kernel<T, false>), dim3(grid), dim3(BLOCK), 0, st, (T*)dst,
                       (const T*)src, outer, n, inner);
  return pthip::post_launch("cumulative");
}

template <class T>
int argmax_typed(long long rows, long long R, const void* src, void* out) {
  if (rows == 0) return 0;
  const unsigned grid = (unsigned)((rows + BLOCK - 1) / BLOCK);
  PTHIP_KLAUNCH((argmax_kernel<T>), dim3(grid), dim3(BLOCK), 0, pthip::ctx().stream,
                     (long long*)out, (const T*)src, rows, R);
  return pthip::post_launch("argmax");
}

}  // namespace

extern "C" {

int pthip_cumulative(int dtype, int mul, int64_t outer, int64_t n, int64_t inner, const void* src,
                     void* dst) {
  PTHIP_REQUIRE_INIT();
  switch (dtype) {
    case PTHIP_F64: return cumulative_typed<double>(mul, outer, n, inner, src, dst);
    case PTHIP_F32: return cumulative_typed<float>(mul, outer, n, inner, src, dst);
    case PTHIP_I64: return cumulative_typed<long long>(mul, outer, n, inner, src, dst);
    case PTHIP_I32: return cumulative_typed<int>(mul, outer, n, inner, src, dst);
    case PTHIP_I16: return cumulative_typed<short>(mul, outer, n, inner, src, dst);
    case PTHIP_I8: return cumulative_typed<signed char>(mul, outer, n, inner, src, dst);
    case PTHIP_U8: return cumulative_typed<unsigned char>(mul, outer, n, inner, src, dst);
    case PTHIP_U16: return cumulative_typed<unsigned short>(mul, outer, n, inner, src, dst);
    case PTHIP_U32: return cumulative_typed<unsigned int>(mul, outer, n, inner, src, dst);
    case PTHIP_U64: return cumulative_typed<unsigned long long>(mul, outer, n, inner, src, dst);
    default: return pthip::set_error("pthip_cumulative: unsupported dtype %d", dtype);
  }
}

int pthip_imatmul(int dtype, int64_t M, int64_t N, int64_t K, const void* A, int64_t sA0,
                  int64_t sA1, const void* B, int64_t sB0, int64_t sB1, void* out) {
  PTHIP_REQUIRE_INIT();
  switch (dtype) {
    case PTHIP_I64: return imatmul_typed<long long>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    case PTHIP_I32: return imatmul_typed<int>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    case PTHIP_I16: return imatmul_typed<short>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    case PTHIP_I8: return imatmul_typed<signed char>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    case PTHIP_U8: return imatmul_typed<unsigned char>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    case PTHIP_U16: return imatmul_typed<unsigned short>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    case PTHIP_U32: return imatmul_typed<unsigned int>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    case PTHIP_U64: return imatmul_typed<unsigned long long>(M, N, K, A, sA0, sA1, B, sB0, sB1, out);
    default: return pthip::set_error("pthip_imatmul: integer dtypes only (got %d)", dtype);
  }
}

int pthip_argmax(int dtype, int64_t rows, int64_t R, const void* src, void* out) {
  PTHIP_REQUIRE_INIT();
  if (R <= 0) return pthip::set_error("pthip_argmax: attempt to get argmax of an empty sequence");
  switch (dtype) {
    case PTHIP_F64: return argmax_typed<double>(rows, R, src, out);
    case PTHIP_F32: return argmax_typed<float>(rows, R, src, out);
    case PTHIP_I64: return argmax_typed<long long>(rows, R, src, out);
    case PTHIP_I32: return argmax_typed<int>(rows, R, src, out);
    case PTHIP_I16: return argmax_typed<short>(rows, R, src, out);
    case PTHIP_I8: return argmax_typed<signed char>(rows, R, src, out);
    case PTHIP_U8: return argmax_typed<unsigned char>(rows, R, src, out);
    case PTHIP_U16: return argmax_typed<unsigned short>(rows, R, src, out);
    case PTHIP_U32: return argmax_typed<unsigned int>(rows, R, src, out);
    case PTHIP_U64: return argmax_typed<unsigned long long>(rows, R, src, out);
    default: return pthip::set_error("pthip_argmax: unsupported dtype %d", dtype);
  }
}

}  // extern "C"
