// nonzero.hip — ordered stream compaction: the flat indices of the true elements of a mask.
//
// Reference: boolean-mask indexing (AdvancedSubtensor / AdvancedIncSubtensor with a bool index,
// pytensor/tensor/subtensor.py:1932, 2275: NumPy's x[mask] == x[mask.nonzero()]) and the Nonzero
// op (pytensor/tensor/basic.py `Nonzero.perform`: np.nonzero).  Index tier: bit-exact, ascending
// C order.  Three launches: per-block counts, one-block exclusive scan of the counts (and the
// total), ordered write.  The output length is data dependent: the total is left in device
// memory and read back by the caller (such graphs are not captured into hipGraphs).
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int ITEMS = 16;  // consecutive elements per thread
constexpr int TILE = BLOCK * ITEMS;

__device__ __forceinline__ long long block_exclusive_scan(long long v, long long* total) {
  __shared__ long long s_wave[BLOCK / 64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  long long incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wid] = incl;
  __syncthreads();
  long long base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; w++) {
    if (w < wid) base += s_wave[w];
    all += s_wave[w];
  }
  __syncthreads();
  *total = all;
  return base + incl - v;
}

__global__ __launch_bounds__(BLOCK) void nz_count_kernel(const unsigned char* __restrict__ m, long long n,
                                                        long long* __restrict__ counts) {
  const long long start = (long long)blockIdx.x * TILE + (long long)threadIdx.x * ITEMS;
  long long c = 0;
#pragma unroll
  for (int j = 0; j < ITEMS; j++)
    if (start + j < n) c += m[start + j] != 0;
  long long total;
  block_exclusive_scan(c, &total);
  if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(BLOCK) void nz_scan_kernel(long long* __restrict__ counts, long long nb,
                                                       long long* __restrict__ total_out) {
  long long carry = 0;
  for (long long base = 0; base < nb; base += BLOCK) {
    const long long i = base + threadIdx.x;
    const long long v = i < nb ? counts[i] : 0;
    long long chunk;
    const long long ex = block_exclusive_scan(v, &chunk);
    if (i < nb) counts[i] = carry + ex;
    carry += chunk;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(BLOCK) void nz_write_kernel(const unsigned char* __restrict__ m, long long n,
                                                        const long long* __restrict__ offsets,
                                                        long long* __restrict__ out) {
  const long long start = (long long)blockIdx.x * TILE + (long long)threadIdx.x * ITEMS;
  unsigned bits = 0;
#pragma unroll
  for (int j = 0; j < ITEMS; j++)
    if (start + j < n && m[start + j] != 0) bits |= 1u << j;
  long long total;
  long long pos = offsets[blockIdx.x] + block_exclusive_scan((long long)__popc(bits), &total);
#pragma unroll
  for (int j = 0; j < ITEMS; j++)
    if (bits & (1u << j)) out[pos++] = start + j;
}

}  // namespace

extern "C" int pthip_nonzero(int64_t n, const void* mask, void* idx_out, void* count_out) {
  PTHIP_REQUIRE_INIT();
  hipStream_t st = pthip::ctx().stream;
  if (n <= 0) {
    PTHIP_CHECK(pthip::memset_async(count_out, 0, sizeof(long long), st));
    return 0;
  }
  const long long nb = (n + TILE - 1) / TILE;
  void* counts = nullptr;
  int r = pthip_alloc((size_t)nb * sizeof(long long), &counts);
  if (r) return r;
  PTHIP_KLAUNCH(nz_count_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, (const unsigned char*)mask, (long long)n, (long long*)counts);
  PTHIP_KLAUNCH(nz_scan_kernel, dim3(1), dim3(BLOCK), 0, st, (long long*)counts, nb, (long long*)count_out);
  PTHIP_KLAUNCH(nz_write_kernel, dim3((unsigned)nb), dim3(BLOCK), 0, st, (const unsigned char*)mask, (long long)n, (const long long*)counts, (long long*)idx_out);
  r = pthip::post_launch("nonzero");
  pthip_free(counts);  // stream-ordered reuse keeps this safe
  return r;
}
