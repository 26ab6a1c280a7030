// misc.hip — small index / signal kernels of the widening tier: SearchsortedOp,
// Convolve1d, Convolve2d.
//
// Reference: SearchsortedOp.perform (pytensor/tensor/extra_ops.py:155-166: np.searchsorted with
// side / sorter), Convolve1d.perform (pytensor/tensor/signal/conv.py: np.convolve, "full" or
// "valid"), Convolve2d.perform (conv.py:260-263: scipy.signal.convolve, direct sums here whatever
// `method` says).  Index tier for searchsorted (bit-exact); the convolution accumulates each output in
// increasing order of the first operand's index, like np.convolve's inner dot.
#include "common.h"

namespace {

__device__ __forceinline__ double load_d(const void* p, int dt, long long e) {
  switch (dt) {
    case PTHIP_F64: return ((const double*)p)[e];
    case PTHIP_F32: return (double)((const float*)p)[e];
    case PTHIP_F16: return (double)((const _Float16*)p)[e];
    case PTHIP_I64: return (double)((const long long*)p)[e];
    case PTHIP_I32: return (double)((const int*)p)[e];
    case PTHIP_I16: return (double)((const short*)p)[e];
    case PTHIP_I8: return (double)((const signed char*)p)[e];
    case PTHIP_U8: return (double)((const unsigned char*)p)[e];
    case PTHIP_U16: return (double)((const unsigned short*)p)[e];
    case PTHIP_U32: return (double)((const unsigned int*)p)[e];
    case PTHIP_U64: return (double)((const unsigned long long*)p)[e];
    default: return (double)((const bool*)p)[e];
  }
}
__device__ __forceinline__ long long load_l(const void* p, int dt, long long e) {
  switch (dt) {
    case PTHIP_I64: return ((const long long*)p)[e];
    case PTHIP_I32: return ((const int*)p)[e];
    case PTHIP_I16: return ((const short*)p)[e];
    case PTHIP_I8: return ((const signed char*)p)[e];
    case PTHIP_U8: return ((const unsigned char*)p)[e];
    case PTHIP_U16: return ((const unsigned short*)p)[e];
    case PTHIP_U32: return ((const unsigned int*)p)[e];
    case PTHIP_U64: return (long long)((const unsigned long long*)p)[e];
    default: return ((const bool*)p)[e];
  }
}

// NumPy's order for floats: NaN behind every number
__device__ __forceinline__ bool less_d(double a, double b) { return a < b || (b != b && a == a); }

// out[j] = the insertion point of v[j] in the sorted x (through sorter, if given)
__global__ void searchsorted_kernel(const void* __restrict__ x, int xdt, long long n, const long long* __restrict__ sorter,
                                    const void* __restrict__ v, int vdt, long long m, int right, int as_int,
                                    long long* __restrict__ out) {
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (long long)gridDim.x * blockDim.x) {
    long long lo = 0, hi = n;
    if (as_int) {
      const long long key = load_l(v, vdt, j);
      while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        const long long xe = load_l(x, xdt, sorter ? sorter[mid] : mid);
        if (right ? !(key < xe) : (xe < key)) lo = mid + 1; else hi = mid;
      }
    } else {
      const double key = load_d(v, vdt, j);
      while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        const double xe = load_d(x, xdt, sorter ? sorter[mid] : mid);
        if (right ? !less_d(key, xe) : less_d(xe, key)) lo = mid + 1; else hi = mid;
      }
    }
    out[j] = lo;
  }
}

// out[i] = sum_j a[j] * b[i + lo - j] over the overlap (i + lo = index in the full convolution)
template <class T>
__global__ void convolve1d_kernel(const T* __restrict__ a, long long na, const T* __restrict__ b, long long nb, long long lo,
                                  long long nout, T* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (long long)gridDim.x * blockDim.x) {
    const long long f = i + lo;
    const long long j0 = f - (nb - 1) > 0 ? f - (nb - 1) : 0;
    const long long j1 = f < na - 1 ? f : na - 1;
    T acc = T(0);
    for (long long j = j0; j <= j1; j++) acc += a[j] * b[f - j];
    out[i] = acc;
  }
}

// out[i][j] = sum_{p,q} a[p][q] * b[i + lo_r - p][j + lo_c - q] over the overlap
template <class T>
__global__ void convolve2d_kernel(const T* __restrict__ a, int ha, int wa, const T* __restrict__ b, int hb, int wb, int lo_r,
                                  int lo_c, int ho, int wo, T* __restrict__ out) {
  const long long total = (long long)ho * wo;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int fi = (int)(e / wo) + lo_r, fj = (int)(e % wo) + lo_c;
    const int p0 = fi - (hb - 1) > 0 ? fi - (hb - 1) : 0, p1 = fi < ha - 1 ? fi : ha - 1;
    const int q0 = fj - (wb - 1) > 0 ? fj - (wb - 1) : 0, q1 = fj < wa - 1 ? fj : wa - 1;
    T acc = T(0);
    for (int p = p0; p <= p1; p++)
      for (int q = q0; q <= q1; q++) acc += a[(long long)p * wa + q] * b[(long long)(fi - p) * wb + (fj - q)];
    out[e] = acc;
  }
}

int grid_for(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int pthip_searchsorted(int x_dtype, int64_t n, const void* x, const void* sorter, int v_dtype, int64_t m,
                                  const void* v, int right, void* out) {
  PTHIP_REQUIRE_INIT();
  if (m <= 0) return 0;
  auto is_float = [](int dt) { return dt == PTHIP_F64 || dt == PTHIP_F32 || dt == PTHIP_F16; };
  const int as_int = !is_float(x_dtype) && !is_float(v_dtype);
  PTHIP_KLAUNCH(searchsorted_kernel, dim3(grid_for(m)), dim3(256), 0, pthip::ctx().stream, x, x_dtype, (long long)n,
                (const long long*)sorter, v, v_dtype, (long long)m, right, as_int, (long long*)out);
  return pthip::post_launch("searchsorted");
}

extern "C" int pthip_convolve1d(int dtype, int64_t na, const void* a, int64_t nb, const void* b, int full, void* out) {
  PTHIP_REQUIRE_INIT();
  if (na <= 0 || nb <= 0) return 0;
  // np.convolve(..., "valid"): max(na, nb) - min(na, nb) + 1 outputs starting at min(na, nb) - 1
  const long long small = na < nb ? na : nb, big = na < nb ? nb : na;
  const long long lo = full ? 0 : small - 1, nout = full ? na + nb - 1 : big - small + 1;
  hipStream_t st = pthip::ctx().stream;
  if (dtype == PTHIP_F64) PTHIP_KLAUNCH(convolve1d_kernel<double>, dim3(grid_for(nout)), dim3(256), 0, st, (const double*)a, (long long)na, (const double*)b, (long long)nb, lo, nout, (double*)out);
  else if (dtype == PTHIP_F32) PTHIP_KLAUNCH(convolve1d_kernel<float>, dim3(grid_for(nout)), dim3(256), 0, st, (const float*)a, (long long)na, (const float*)b, (long long)nb, lo, nout, (float*)out);
  else if (dtype == PTHIP_I64) PTHIP_KLAUNCH(convolve1d_kernel<long long>, dim3(grid_for(nout)), dim3(256), 0, st, (const long long*)a, (long long)na, (const long long*)b, (long long)nb, lo, nout, (long long*)out);
  else return pthip::set_error("pthip_convolve1d: dtype %d not supported (float32/float64/int64)", dtype);
  return pthip::post_launch("convolve1d");
}

extern "C" int pthip_convolve2d(int dtype, int64_t ha, int64_t wa, const void* a, int64_t hb, int64_t wb, const void* b, int full,
                                void* out) {
  PTHIP_REQUIRE_INIT();
  if (ha <= 0 || wa <= 0 || hb <= 0 || wb <= 0) return 0;
  if (!full && !((ha >= hb && wa >= wb) || (hb >= ha && wb >= wa)))
    return pthip::set_error("pthip_convolve2d: for 'valid' mode, one must be at least as large as the other in every dimension");
  auto mn = [](long long x, long long y) { return x < y ? x : y; };
  auto mx = [](long long x, long long y) { return x > y ? x : y; };
  const int lo_r = full ? 0 : (int)mn(ha, hb) - 1, lo_c = full ? 0 : (int)mn(wa, wb) - 1;
  const int ho = full ? (int)(ha + hb - 1) : (int)(mx(ha, hb) - mn(ha, hb) + 1);
  const int wo = full ? (int)(wa + wb - 1) : (int)(mx(wa, wb) - mn(wa, wb) + 1);
  hipStream_t st = pthip::ctx().stream;
  const int g = grid_for((long long)ho * wo);
  if (dtype == PTHIP_F64) PTHIP_KLAUNCH(convolve2d_kernel<double>, dim3(g), dim3(256), 0, st, (const double*)a, (int)ha, (int)wa, (const double*)b, (int)hb, (int)wb, lo_r, lo_c, ho, wo, (double*)out);
  else if (dtype == PTHIP_F32) PTHIP_KLAUNCH(convolve2d_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)a, (int)ha, (int)wa, (const float*)b, (int)hb, (int)wb, lo_r, lo_c, ho, wo, (float*)out);
  else if (dtype == PTHIP_I64) PTHIP_KLAUNCH(convolve2d_kernel<long long>, dim3(g), dim3(256), 0, st, (const long long*)a, (int)ha, (int)wa, (const long long*)b, (int)hb, (int)wb, lo_r, lo_c, ho, wo, (long long*)out);
  else return pthip::set_error("pthip_convolve2d: dtype %d not supported (float32/float64/int64)", dtype);
  return pthip::post_launch("convolve2d");
}
