// fill.hip — ARange and Eye generated on the device.
//
// Reference: ARange.perform (pytensor/tensor/basic.py: np.arange(start, stop, step, dtype)) and
// Eye.perform (np.eye(n, m, k, dtype)).  The host knows the scalars (they fix the output shape);
// building the array on the host and uploading it would put a pageable host-to-device copy into
// every replay of a captured plan (measured: 32 KB cost ~0.2 ms), so the values are produced by
// a kernel instead.  NumPy's fill loops are restated: integers start + i*step exactly; floating
// point `start + i*delta` with delta = (start + step) - start evaluated in the output type.
#include "common.h"

namespace {

constexpr int BLOCK = 256;

template <class T>
__global__ __launch_bounds__(BLOCK) void arange_float_kernel(T* __restrict__ out, long long n, double start, double step) {
  const T s = (T)start;
  const T delta = (T)(start + step) - s;  // NumPy: buffer[1] - buffer[0]
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) {
#pragma clang fp contract(off)
    const T scaled = (T)i * delta;  // two rounded operations, as NumPy's fill loop
    out[i] = s + scaled;
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void arange_int_kernel(T* __restrict__ out, long long n, long long start, long long step) {
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK)
    out[i] = (T)(start + i * step);
}

template <class T>
__global__ __launch_bounds__(BLOCK) void eye_kernel(T* __restrict__ out, long long n, long long m, long long k) {
  const long long total = n * m;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < total; e += (long long)gridDim.x * BLOCK) {
    const long long i = e / m, j = e - i * m;
    out[e] = (j - i == k) ? T(1) : T(0);
  }
}

int grid_for(long long n) {
  long long g = (n + BLOCK - 1) / BLOCK;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int pthip_arange(int dtype, int64_t n, double fstart, double fstep, int64_t istart, int64_t istep, void* out) {
  PTHIP_REQUIRE_INIT();
  if (n <= 0) return 0;
  hipStream_t st = pthip::ctx().stream;
  const dim3 g(grid_for(n)), b(BLOCK);
  switch (dtype) {
    case PTHIP_F64: PTHIP_KLAUNCH(arange_float_kernel<double>, g, b, 0, st, (double*)out, (long long)n, fstart, fstep); break;
    case PTHIP_F32: PTHIP_KLAUNCH(arange_float_kernel<float>, g, b, 0, st, (float*)out, (long long)n, fstart, fstep); break;
    case PTHIP_I64: PTHIP_KLAUNCH(arange_int_kernel<long long>, g, b, 0, st, (long long*)out, (long long)n, (long long)istart, (long long)istep); break;
    case PTHIP_I32: PTHIP_KLAUNCH(arange_int_kernel<int>, g, b, 0, st, (int*)out, (long long)n, (long long)istart, (long long)istep); break;
    case PTHIP_I16: PTHIP_KLAUNCH(arange_int_kernel<short>, g, b, 0, st, (short*)out, (long long)n, (long long)istart, (long long)istep); break;
    case PTHIP_I8: PTHIP_KLAUNCH(arange_int_kernel<signed char>, g, b, 0, st, (signed char*)out, (long long)n, (long long)istart, (long long)istep); break;
    case PTHIP_U8: PTHIP_KLAUNCH(arange_int_kernel<unsigned char>, g, b, 0, st, (unsigned char*)out, (long long)n, (long long)istart, (long long)istep); break;
    case PTHIP_U16: PTHIP_KLAUNCH(arange_int_kernel<unsigned short>, g, b, 0, st, (unsigned short*)out, (long long)n, (long long)istart, (long long)istep); break;
    case PTHIP_U32: PTHIP_KLAUNCH(arange_int_kernel<unsigned int>, g, b, 0, st, (unsigned int*)out, (long long)n, (long long)istart, (long long)istep); break;
    case PTHIP_U64: PTHIP_KLAUNCH(arange_int_kernel<unsigned long long>, g, b, 0, st, (unsigned long long*)out, (long long)n, (long long)istart, (long long)istep); break;
    default: return pthip::set_error("pthip_arange: unsupported dtype %d", dtype);
  }
  return pthip::post_launch("arange");
}

extern "C" int pthip_eye(int dtype, int64_t n, int64_t m, int64_t k, void* out) {
  PTHIP_REQUIRE_INIT();
  if (n <= 0 || m <= 0) return 0;
  hipStream_t st = pthip::ctx().stream;
  const dim3 g(grid_for(n * m)), b(BLOCK);
  switch (dtype) {
    case PTHIP_F64: PTHIP_KLAUNCH(eye_kernel<double>, g, b, 0, st, (double*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_F32: PTHIP_KLAUNCH(eye_kernel<float>, g, b, 0, st, (float*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_I64: PTHIP_KLAUNCH(eye_kernel<long long>, g, b, 0, st, (long long*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_I32: PTHIP_KLAUNCH(eye_kernel<int>, g, b, 0, st, (int*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_I16: PTHIP_KLAUNCH(eye_kernel<short>, g, b, 0, st, (short*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_I8: PTHIP_KLAUNCH(eye_kernel<signed char>, g, b, 0, st, (signed char*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_U8: case PTHIP_BOOL: PTHIP_KLAUNCH(eye_kernel<unsigned char>, g, b, 0, st, (unsigned char*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_U16: PTHIP_KLAUNCH(eye_kernel<unsigned short>, g, b, 0, st, (unsigned short*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_U32: PTHIP_KLAUNCH(eye_kernel<unsigned int>, g, b, 0, st, (unsigned int*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_U64: PTHIP_KLAUNCH(eye_kernel<unsigned long long>, g, b, 0, st, (unsigned long long*)out, (long long)n, (long long)m, (long long)k); break;
    case PTHIP_F16: PTHIP_KLAUNCH(eye_kernel<_Float16>, g, b, 0, st, (_Float16*)out, (long long)n, (long long)m, (long long)k); break;
    default: return pthip::set_error("pthip_eye: unsupported dtype %d", dtype);
  }
  return pthip::post_launch("eye");
}
