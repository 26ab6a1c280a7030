// sort.hip — SortOp / ArgSortOp along the last axis: bitonic network on (key, index) pairs.
//
// Reference: SortOp.perform / ArgSortOp.perform (pytensor/tensor/sort.py: np.sort / np.argsort
// along `axis`, kind quicksort|mergesort|heapsort|stable).  Index tier: the sorted values are
// bit-exact whatever the algorithm; NaNs sort last like NumPy's; ties are ordered by position
// (a stable order), which is NumPy's answer for kind="stable"/"mergesort" and one of the valid
// answers of its unstable default.  A bitonic network is the natural GPU form: fixed, data-
// independent compare-exchange pattern, rows of up to 4096 elements entirely in LDS (one
// workgroup per row, one launch); longer rows run the far exchanges (distance >= 2048) as one
// launch per step over a padded global copy and all near exchanges of a phase in one LDS pass.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int TILE = 4096;

template <class T>
__device__ __forceinline__ bool pair_less(T ka, int ia, T kb, int ib, int n) {
  const bool pa = ia >= n, pb = ib >= n;  // padding sorts behind every real element
  if (pa != pb) return pb;
  if (!pa) {
    const bool na = ka != ka, nb = kb != kb;  // NaN behind every number
    if (na != nb) return nb;
    if (!na && ka != kb) return ka < kb;
  }
  return ia < ib;
}

template <class T>
__device__ __forceinline__ void cmp_exchange(T* k, int* x, int e, int p, bool up, int n) {
  const T ke = k[e], kp = k[p];
  const int ie = x[e], ip = x[p];
  if (pair_less(kp, ip, ke, ie, n) == up) { k[e] = kp; k[p] = ke; x[e] = ip; x[p] = ie; }
}

// rows with P <= TILE: load, sort, store in one launch
template <class T>
__global__ __launch_bounds__(BLOCK) void sort_lds_kernel(const T* __restrict__ in, int n, int P, T* __restrict__ out_vals,
                                                        long long* __restrict__ out_idx) {
  __shared__ T s_k[TILE];
  __shared__ int s_x[TILE];
  const long long row = blockIdx.x;
  const T* src = in + row * (long long)n;
  for (int e = threadIdx.x; e < P; e += BLOCK) { s_k[e] = e < n ? src[e] : T(0); s_x[e] = e; }
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (P >> 1); t += BLOCK) {
        const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        cmp_exchange(s_k, s_x, e, e + j, (e & k) == 0, n);
      }
    }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += BLOCK) {
    if (out_vals) out_vals[row * (long long)n + e] = s_k[e];
    if (out_idx) out_idx[row * (long long)n + e] = s_x[e];
  }
}

// ---- long rows: padded global copy of (key, index) ----
template <class T>
__global__ __launch_bounds__(BLOCK) void sort_init_kernel(const T* __restrict__ in, long long rows, int n, int P,
                                                         T* __restrict__ keys, int* __restrict__ idx) {
  const long long total = rows * (long long)P;
  for (long long g = (long long)blockIdx.x * BLOCK + threadIdx.x; g < total; g += (long long)gridDim.x * BLOCK) {
    const long long row = g / P;
    const int e = (int)(g - row * P);
    keys[g] = e < n ? in[row * (long long)n + e] : T(0);
    idx[g] = e;
  }
}

// one TILE of one row in LDS: phases k = 2 .. TILE (first == 1), or the near steps
// j = TILE/2 .. 1 of one phase k > TILE (first == 0)
template <class T>
__global__ __launch_bounds__(BLOCK) void sort_tile_kernel(T* __restrict__ keys, int* __restrict__ idx, int n, int P,
                                                         int k_phase, int first) {
  __shared__ T s_k[TILE];
  __shared__ int s_x[TILE];
  const int tiles = P / TILE;
  const long long row = blockIdx.x / tiles;
  const int base = (int)(blockIdx.x % tiles) * TILE;
  T* gk = keys + row * (long long)P + base;
  int* gx = idx + row * (long long)P + base;
  for (int e = threadIdx.x; e < TILE; e += BLOCK) { s_k[e] = gk[e]; s_x[e] = gx[e]; }
  const int k0 = first ? 2 : k_phase, k1 = first ? TILE : k_phase;
  for (int k = k0; k <= k1; k <<= 1)
    for (int j = (k > TILE ? TILE : k) >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (TILE >> 1); t += BLOCK) {
        const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        cmp_exchange(s_k, s_x, e, e + j, ((base + e) & k) == 0, n);
      }
    }
  __syncthreads();
  for (int e = threadIdx.x; e < TILE; e += BLOCK) { gk[e] = s_k[e]; gx[e] = s_x[e]; }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void sort_far_step_kernel(T* __restrict__ keys, int* __restrict__ idx, long long rows,
                                                             int n, int P, int k, int j) {
  const long long half = P >> 1, total = rows * half;
  for (long long g = (long long)blockIdx.x * BLOCK + threadIdx.x; g < total; g += (long long)gridDim.x * BLOCK) {
    const long long row = g / half;
    const int t = (int)(g - row * half);
    const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    cmp_exchange(keys + row * (long long)P, idx + row * (long long)P, e, e + j, (e & k) == 0, n);
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void sort_store_kernel(const T* __restrict__ keys, const int* __restrict__ idx,
                                                          long long rows, int n, int P, T* __restrict__ out_vals,
                                                          long long* __restrict__ out_idx) {
  const long long total = rows * (long long)n;
  for (long long g = (long long)blockIdx.x * BLOCK + threadIdx.x; g < total; g += (long long)gridDim.x * BLOCK) {
    const long long row = g / n;
    const long long s = row * (long long)P + (g - row * n);
    if (out_vals) out_vals[g] = keys[s];
    if (out_idx) out_idx[g] = idx[s];
  }
}

int grid_for(long long work) {
  long long g = (work + BLOCK - 1) / BLOCK;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <class T>
int sort_typed(long long rows, long long n, const void* in, void* out_vals, void* out_idx) {
  if (rows == 0 || n == 0) return 0;
  if (n > (1ll << 30)) return pthip::set_error("pthip_sort: rows longer than 2^30 are not supported");
  hipStream_t st = pthip::ctx().stream;
  int P = 2;
  while (P < n) P <<= 1;
  if (P <= TILE) {
    PTHIP_KLAUNCH(sort_lds_kernel<T>, dim3((unsigned)rows), dim3(BLOCK), 0, st, (const T*)in, (int)n, P,
                       (T*)out_vals, (long long*)out_idx);
    return pthip::post_launch("sort_lds");
  }
  void *keys = nullptr, *idx = nullptr;
  int r = pthip_alloc((size_t)rows * P * sizeof(T), &keys);
  if (r) return r;
  r = pthip_alloc((size_t)rows * P * sizeof(int), &idx);
  if (r) { pthip_free(keys); return r; }
  const unsigned tiles = (unsigned)(rows * (P / TILE));
  PTHIP_KLAUNCH(sort_init_kernel<T>, dim3(grid_for(rows * P)), dim3(BLOCK), 0, st, (const T*)in, rows, (int)n, P, (T*)keys, (int*)idx);
  PTHIP_KLAUNCH(sort_tile_kernel<T>, dim3(tiles), dim3(BLOCK), 0, st, (T*)keys, (int*)idx, (int)n, P, 0, 1);
  for (long long k = 2ll * TILE; k <= P; k <<= 1) {
    for (long long j = k >> 1; j >= TILE; j >>= 1)
      PTHIP_KLAUNCH(sort_far_step_kernel<T>, dim3(grid_for(rows * (P / 2))), dim3(BLOCK), 0, st, (T*)keys, (int*)idx, rows, (int)n, P, (int)k, (int)j);
    PTHIP_KLAUNCH(sort_tile_kernel<T>, dim3(tiles), dim3(BLOCK), 0, st, (T*)keys, (int*)idx, (int)n, P, (int)k, 0);
  }
  PTHIP_KLAUNCH(sort_store_kernel<T>, dim3(grid_for(rows * n)), dim3(BLOCK), 0, st, (const T*)keys, (const int*)idx, rows, (int)n, P, (T*)out_vals, (long long*)out_idx);
  r = pthip::post_launch("sort");
  pthip_free(keys);  // stream-ordered reuse keeps this safe
  pthip_free(idx);
  return r;
}

}  // namespace

extern "C" int pthip_sort(int dtype, int64_t rows, int64_t n, const void* in, void* out_vals, void* out_idx) {
  PTHIP_REQUIRE_INIT();
  switch (dtype) {
    case PTHIP_F64: return sort_typed<double>(rows, n, in, out_vals, out_idx);
    case PTHIP_F32: return sort_typed<float>(rows, n, in, out_vals, out_idx);
    case PTHIP_I64: return sort_typed<long long>(rows, n, in, out_vals, out_idx);
    case PTHIP_I32: return sort_typed<int>(rows, n, in, out_vals, out_idx);
    case PTHIP_I16: return sort_typed<short>(rows, n, in, out_vals, out_idx);
    case PTHIP_I8: return sort_typed<signed char>(rows, n, in, out_vals, out_idx);
    case PTHIP_U8: case PTHIP_BOOL: return sort_typed<unsigned char>(rows, n, in, out_vals, out_idx);
    case PTHIP_U16: return sort_typed<unsigned short>(rows, n, in, out_vals, out_idx);
    case PTHIP_U32: return sort_typed<unsigned int>(rows, n, in, out_vals, out_idx);
    case PTHIP_U64: return sort_typed<unsigned long long>(rows, n, in, out_vals, out_idx);
    default: return pthip::set_error("pthip_sort: unsupported dtype %d", dtype);
  }
}
