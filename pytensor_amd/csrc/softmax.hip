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
#include "exp_device.h"

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
  // Round 6: a wave walks rows wave, wave + nwaves, ... with the NEXT row's packs requested into a second register set
  // before the current row is reduced, exponentiated and stored (two named buffers, unconditional loads at clamped
  // addresses: a branch around a load or a copy of its result makes the compiler wait for every outstanding load).
  // One row per wave and all waves resident at once (round 5) meant the whole chip loaded, then computed, then stored.
  typedef typename Acc<T>::type A;
  typedef sm_pack<T, V> P;
  const pthip_dev::ExpCtx<T> ek;  // (exp_device.h: 24-instruction fp64 exp, constants in registers)
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
  const T NEG = -__builtin_huge_val();
  const int last = cols - V > 0 ? cols - V : 0;
  auto request = [&](long long r, P (&buf)[VPL]) {
    const T* xr = x + (r < rows ? r : rows - 1) * cols;
#pragma unroll
    for (int u = 0; u < VPL; u++) {
      const int j = (lane + 64 * u) * V;
      buf[u] = *reinterpret_cast<const P*>(xr + (j < cols ? j : last));
    }
  };
  auto finish_row = [&](long long r, P (&v)[VPL]) {
    T* orow = out + r * cols;
    T m = NEG;
#pragma unroll
    for (int u = 0; u < VPL; u++) {
      const bool in = (lane + 64 * u) * V < cols;
#pragma unroll
      for (int e = 0; e < V; e++) {
        v[u].v[e] = in ? v[u].v[e] : NEG;
        m = nan_max(m, v[u].v[e]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_xor(m, o));
    A s = A(0);
#pragma unroll
    for (int u = 0; u < VPL; u++) {
#pragma unroll
      for (int e = 0; e < V; e++) {
        const T d = v[u].v[e] - m;
        const T ex = ek(d);
        s += (A)ex;
        v[u].v[e] = LOG ? d : ex;
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
#pragma unroll
      for (int e = 0; e < V; e++) v[u].v[e] = LOG ? (v[u].v[e] - ls) : (v[u].v[e] * inv);
      if (j < cols) *reinterpret_cast<P*>(orow + j) = v[u];
    }
  };
  long long r = wave;
  if (r >= rows) return;
  P bufA[VPL], bufB[VPL];
  request(r, bufA);
  while (true) {
    request(r + nwaves, bufB);
    finish_row(r, bufA);
    r += nwaves;
    if (r >= rows) break;
    request(r + nwaves, bufA);
    finish_row(r, bufB);
    r += nwaves;
    if (r >= rows) break;
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

// The workgroup's 256 consecutive rows (one contiguous chunk of n = rows * cols elements) between global memory and the
// padded LDS tile [row][pitch], 16 bytes per lane and instruction when the chunk is 16-byte aligned (round 6: the 8-byte
// loop moved half the bytes per instruction of every other streaming kernel here).  idx / cols without a division:
// magic = ceil(2^20 / cols) is exact for idx < 256 * 33.
template <class T>
__device__ __forceinline__ void stage_rows_in(T* __restrict__ tile, const T* __restrict__ src, int n, int cols, int pitch, unsigned magic, bool vec) {
  constexpr int VW = 16 / (int)sizeof(T);
  typedef sm_pack<T, VW> P;
  const int np = vec ? n / VW : 0;
#pragma unroll 4
  for (int p = threadIdx.x; p < np; p += BLOCK) {
    const P v = reinterpret_cast<const P*>(src)[p];
#pragma unroll
    for (int e = 0; e < VW; e++) {
      const int idx = p * VW + e;
      const int r = (int)(((unsigned)idx * magic) >> 20), j = idx - r * cols;
      tile[r * pitch + j] = v.v[e];
    }
  }
#pragma unroll 4
  for (int idx = np * VW + threadIdx.x; idx < n; idx += BLOCK) {  // (the whole chunk when !vec: several loads in flight)
    const int r = (int)(((unsigned)idx * magic) >> 20), j = idx - r * cols;
    tile[r * pitch + j] = src[idx];
  }
}
template <class T>
__device__ __forceinline__ void stage_rows_out(T* __restrict__ dst, const T* __restrict__ tile, int n, int cols, int pitch, unsigned magic, bool vec) {
  constexpr int VW = 16 / (int)sizeof(T);
  typedef sm_pack<T, VW> P;
  const int np = vec ? n / VW : 0;
#pragma unroll 4
  for (int p = threadIdx.x; p < np; p += BLOCK) {
    P v;
#pragma unroll
    for (int e = 0; e < VW; e++) {
      const int idx = p * VW + e;
      const int r = (int)(((unsigned)idx * magic) >> 20), j = idx - r * cols;
      v.v[e] = tile[r * pitch + j];
    }
    reinterpret_cast<P*>(dst)[p] = v;
  }
#pragma unroll 4
  for (int idx = np * VW + threadIdx.x; idx < n; idx += BLOCK) {
    const int r = (int)(((unsigned)idx * magic) >> 20), j = idx - r * cols;
    dst[idx] = tile[r * pitch + j];
  }
}

// rows of <= MAXC elements: thread per row — but a lane's row is cols*sizeof(T) bytes away from its neighbour's,
// so the workgroup's 256 consecutive rows (one contiguous chunk of memory) are staged through LDS: coalesced
// loads in, a row per thread out of LDS (odd pitch: conflict-free), results back the same way.
// (1e6 x 10 fp64: 87 us with strided per-lane rows -> profiles/r5*_hotpath*.md)
template <class T, bool LOG, int MAXC>
__global__ __launch_bounds__(BLOCK) void softmax_small_kernel(T* __restrict__ out,
                                                             const T* __restrict__ x,
                                                             long long rows, int cols, unsigned magic, int vec) {
  typedef typename Acc<T>::type A;
  const pthip_dev::ExpCtx<T> ek;
  __shared__ T tile[BLOCK * (MAXC + 1)];
  const int pitch = cols | 1;
  const long long r0 = (long long)blockIdx.x * BLOCK;
  const long long left = rows - r0;
  const int nr = left < BLOCK ? (int)left : BLOCK;
  const int n = nr * cols;
  const T* src = x + r0 * cols;
  T* dst = out + r0 * cols;
  stage_rows_in<T>(tile, src, n, cols, pitch, magic, vec != 0);
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
        const T ex = ek(d);
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
  stage_rows_out<T>(dst, tile, n, cols, pitch, magic, vec != 0);
}

template <class T>
int softmax_typed(int log_, long long rows, long long cols, const void* x, void* out) {
  if (rows == 0 || cols == 0) return 0;
  hipStream_t st = pthip::ctx().stream;
  if (cols <= 16) {
    const unsigned grid = (unsigned)((rows + BLOCK - 1) / BLOCK);
    const unsigned magic = (unsigned)(((1u << 20) + (unsigned)cols - 1) / (unsigned)cols);
    const int vec = ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((BLOCK * cols * sizeof(T)) % 16) == 0;
    // (a persistent form of this kernel — tiles walked by 2-6 workgroups per CU, the next tile's packs prefetched into
    //  registers — was measured at 41.8-42.4 us against 34.4 for one tile per workgroup at 1e6 x 10: three barriers per tile
    //  and NPK = 8 pack slots for 5 packs; not kept)
    if (log_)
      PTHIP_KLAUNCH((softmax_small_kernel<T, true, 16>), dim3(grid), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (int)cols, magic, vec);
    else
      PTHIP_KLAUNCH((softmax_small_kernel<T, false, 16>), dim3(grid), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (int)cols, magic, vec);
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
  // persistent waves with a prefetched row each: ~12 waves per CU hold two register sets of a 2048-column fp64 row
  static const int sm_per_cu = [] { const char* e = getenv("PTHIP_SOFTMAX_WG_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 3; }();
  const long long cap = (long long)pthip::kNumCU * sm_per_cu;
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


// =====================================================================================================================
// Round 6: log-sum-exp as its own kernels, and softmax / log-softmax over a NON-trailing axis of a contiguous tensor.
//
// The reference has no single op for these: ``logsumexp`` (pytensor/tensor/math.py) is rewritten into
// Max -> Composite -> Sum -> Composite (tests/benchmarks/test_logsumexp.py:9-37 is its benchmark) and a column softmax is
// ``Softmax(axis=0)`` (special.py:26; perform = scipy.special.softmax).  The hip linker's IR passes (axisfuse.py) turn the
// former into ONE reduction node whose operand is X itself; until this round that node and the column softmax ran on
// the generated N-d reduce tile (codegen_tile.py), whose per-element state update (OpLse::push: a branch, two selects
// and an exp per element) and thread mapping cost 15-40 % of HBM on the shapes below.
//
// The running state is the pair (m, s):  sum_i exp(x_i) = s * exp(sh(m)),  m = max_i x_i,  sh(m) = isinf(m) ? 0 : m —
// the reference's own stabilisation (``switch(isinf(max), 0, max)``): an all -inf slice gives -inf, a +inf term +inf,
// NaN propagates through s.  A thread takes U values per column at a time: one max over them, ONE rescale
// s *= exp(sh_old - sh_new) per U values, one exp per value — no per-element branch.
// =====================================================================================================================

template <class T> struct LseSt { T m; double s; };

template <class T> __device__ __forceinline__ T lse_shift(T m) { return __builtin_isinf(m) ? T(0) : m; }
template <class T> __device__ __forceinline__ T max_nn(T a, T b) { return a > b ? a : b; }  // (NaN never wins: it reaches s instead)

// fold the partial state b into a
template <class T> __device__ __forceinline__ LseSt<T> lse_merge(LseSt<T> a, LseSt<T> b, const pthip_dev::ExpCtx<T>& ex) {
  const T M = max_nn(a.m, b.m);
  const T sh = lse_shift(M);
  // a state with s == 0 is empty / all -inf: it contributes nothing (and 0 * exp(+big) must not become NaN)
  const double fa = a.s == 0.0 ? 0.0 : a.s * (double)ex(lse_shift(a.m) - sh);
  const double fb = b.s == 0.0 ? 0.0 : b.s * (double)ex(lse_shift(b.m) - sh);
  return LseSt<T>{M, fa + fb};
}

template <class T> __device__ __forceinline__ T lse_value(LseSt<T> a) { return (T)(log(a.s)) + lse_shift(a.m); }

// ---- rows of <= 32 elements: thread per row, the workgroup's 256 consecutive rows staged through LDS ---------------
template <class T, int MAXC>
__global__ __launch_bounds__(BLOCK) void lse_rows_small_kernel(T* __restrict__ out, const T* __restrict__ x, long long rows, int cols, unsigned magic, int vec) {
  const pthip_dev::ExpCtx<T> ex;
  extern __shared__ __attribute__((aligned(16))) unsigned char lse_lds_[];
  T* tile = reinterpret_cast<T*>(lse_lds_);
  const int pitch = cols | 1;
  const long long r0 = (long long)blockIdx.x * BLOCK;
  const long long left = rows - r0;
  const int nr = left < BLOCK ? (int)left : BLOCK;
  const int n = nr * cols;
  const T* src = x + r0 * cols;
  stage_rows_in<T>(tile, src, n, cols, pitch, magic, vec != 0);
  __syncthreads();
  if ((int)threadIdx.x < nr) {
    const T* row = tile + threadIdx.x * pitch;
    T v[MAXC];
    T m = -__builtin_huge_val();
#pragma unroll
    for (int j = 0; j < MAXC; j++)
      if (j < cols) { v[j] = row[j]; m = max_nn(m, v[j]); }
    const T sh = lse_shift(m);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < MAXC; j++)
      if (j < cols) s += (double)ex(v[j] - sh);
    out[r0 + threadIdx.x] = (T)log(s) + sh;
  }
}

// ---- rows of up to 64*VPL*V elements: wave per row, the row in registers: a max sweep, then ONE exp per element and no
//      rescaling at all (the arithmetic of the reference's stabilised graph: max, then sum(exp(x - max))) ---------------
template <class T, int VPL, int V>
__global__ __launch_bounds__(BLOCK) void lse_rows_wave_kernel(T* __restrict__ out, const T* __restrict__ x, long long rows, int cols) {
  // persistent: the next row of the wave is requested into a second register set before the current one is reduced
  // (see softmax_wave_reg_kernel)
  const pthip_dev::ExpCtx<T> ex;
  typedef sm_pack<T, V> P;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
  const T NEG = -__builtin_huge_val();
  const int last = cols - V > 0 ? cols - V : 0;
  auto request = [&](long long r, P (&buf)[VPL]) {
    const T* xr = x + (r < rows ? r : rows - 1) * cols;
#pragma unroll
    for (int u = 0; u < VPL; u++) {  // (unconditional, clamped: all VPL loads in flight together)
      const int j = (lane + 64 * u) * V;
      buf[u] = *reinterpret_cast<const P*>(xr + (j < cols ? j : last));
    }
  };
  auto finish_row = [&](long long r, P (&v)[VPL]) {
    T m = NEG;
#pragma unroll
    for (int u = 0; u < VPL; u++) {
      const bool in = (lane + 64 * u) * V < cols;
#pragma unroll
      for (int e = 0; e < V; e++) {
        v[u].v[e] = in ? v[u].v[e] : NEG;
        m = max_nn(m, v[u].v[e]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max_nn(m, __shfl_xor(m, o));
    const T sh = lse_shift(m);
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < VPL; u++) {
#pragma unroll
      for (int e = 0; e < V; e++) s += (double)ex(v[u].v[e] - sh);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = (T)log(s) + sh;
  };
  long long r = wave;
  if (r >= rows) return;
  P bufA[VPL], bufB[VPL];
  request(r, bufA);
  while (true) {
    request(r + nwaves, bufB);
    finish_row(r, bufA);
    r += nwaves;
    if (r >= rows) break;
    request(r + nwaves, bufA);
    finish_row(r, bufB);
    r += nwaves;
    if (r >= rows) break;
  }
}

// ---- rows of ANY length > 32: a wave streams its rows in visits of U packs per lane, the next visit (of this row or
//      of the wave's next row) already in flight while the current one is folded into the lane's (m, s); at the end of a
//      row the 64 lane states meet in a butterfly.  One exp per element + one rescale per visit. --------------------------
template <class T, int V, int U>
__global__ __launch_bounds__(BLOCK) void lse_rows_stream_kernel(T* __restrict__ out, const T* __restrict__ x, long long rows, long long cols, int G) {
  const pthip_dev::ExpCtx<T> ex;
  typedef sm_pack<T, V> P;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (BLOCK / 64);
  const T NEG = -__builtin_huge_val();
  // (unconditional loads at clamped addresses + two named buffers: see col_lse_partial_kernel)
  const long long last = (cols - V) > 0 ? (cols - V) : 0;
  auto load_visit = [&](long long r, int g, P (&buf)[U]) {
    const long long rc = r < rows ? r : rows - 1;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long j = ((long long)(g * U + u) * 64 + lane) * V;
      buf[u] = *reinterpret_cast<const P*>(x + rc * cols + (j < cols ? j : last));
    }
  };
  LseSt<T> st{NEG, 0.0};
  auto fold_visit = [&](long long r, int g, int gn, P (&buf)[U]) {
    T mc = NEG;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool in = ((long long)(g * U + u) * 64 + lane) * V < cols;
#pragma unroll
      for (int e = 0; e < V; e++) {
        buf[u].v[e] = in ? buf[u].v[e] : NEG;
        mc = max_nn(mc, buf[u].v[e]);
      }
    }
    const T mn = max_nn(st.m, mc);
    const T sh_old = lse_shift(st.m), sh = lse_shift(mn);
    double acc = st.s == 0.0 ? 0.0 : st.s * (double)ex(sh_old - sh);
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
      for (int e = 0; e < V; e++) acc += (double)ex(buf[u].v[e] - sh);
    }
    st.m = mn;
    st.s = acc;
    if (gn == 0) {  // the row is complete: the lanes agree on the max, rescale ONCE each, add (fixed order), write
      T M = st.m;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) M = max_nn(M, __shfl_xor(M, o));
      const T shM = lse_shift(M);
      double sl = st.s == 0.0 ? 0.0 : st.s * (double)ex(lse_shift(st.m) - shM);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sl += __shfl_xor(sl, o);
      if (lane == 0) out[r] = (T)log(sl) + shM;
      st = LseSt<T>{NEG, 0.0};
    }
  };
  long long r = wave;
  int g = 0;
  if (r >= rows) return;
  P A[U], B[U];
  load_visit(r, g, A);
  while (true) {
    long long rn = r;
    int gn = g + 1;
    if (gn == G) { gn = 0; rn = r + nwaves; }
    load_visit(rn, gn, B);
    fold_visit(r, g, gn, A);
    r = rn; g = gn;
    if (r >= rows) break;
    gn = g + 1; rn = r;
    if (gn == G) { gn = 0; rn = r + nwaves; }
    load_visit(rn, gn, A);
    fold_visit(r, g, gn, B);
    r = rn; g = gn;
    if (r >= rows) break;
  }
}

// ---- the column statistics of a contiguous (batch, R, C) tensor: reduce over R -------------------------------------
// Thread (ty, tx) of the 256: tx = its pack of V adjacent columns inside the block's column tile, ty = its row lane.
//   C / V <= 256 packs: ONE tile, TX = C / V, TY = 256 / TX — the block then reads whole rows, i.e. contiguous memory,
//                       and a thread's columns never change (1e6 x 10: TX = 5, 51 row lanes, 255 threads busy);
//   wider:              tiles of TX = 64 packs (a wave reads 1 KB of a row), TY = 4 row lanes.
// The rows are split over gridDim.y blocks; each block leaves one (m, s) pair per column in the workspace
// ([batch][split][C], two planes) and the finish kernel folds the splits.
struct ColGeom {
  long long R, C;       // reduced extent, kept (trailing) extent
  long long rows_per;   // rows of one split
  int TX, TY, nsplit;
};

template <class T, int V, int U>
__global__ __launch_bounds__(BLOCK) void col_lse_partial_kernel(double* __restrict__ pm, double* __restrict__ ps, const T* __restrict__ x, ColGeom g) {
  const pthip_dev::ExpCtx<T> ex;
  typedef sm_pack<T, V> P;
  __shared__ double sm_m[BLOCK * V];
  __shared__ double sm_s[BLOCK * V];
  const int t = threadIdx.x;
  const int ty = t / g.TX, tx = t - ty * g.TX;
  const long long col = ((long long)blockIdx.x * g.TX + tx) * V;
  const bool live = ty < g.TY && col < g.C;
  const long long b = blockIdx.z;
  const long long r_begin = (long long)blockIdx.y * g.rows_per;
  long long r_end = r_begin + g.rows_per;
  if (r_end > g.R) r_end = g.R;
  T m[V];
  double s[V];
#pragma unroll
  for (int e = 0; e < V; e++) { m[e] = -__builtin_huge_val(); s[e] = 0.0; }
  if (live) {
    // Software-pipelined: the U packs of the NEXT visit are requested before the current ones are consumed, so the
    // exps of one visit run under the loads of the next.  (Without it every wave alternated a load phase and a compute
    // phase and, all waves having started together, the memory system idled during the compute phases:
    // 31 us for 134 MB where a plain sum takes 22, profiles/r7f_lse_kernels.md.)  Rows past the split's end are
    // filled with -inf: they add exp(-inf) = 0.
    const T* base = x + (b * g.R) * g.C + col;
    const T NEG = -__builtin_huge_val();
    // (loads are UNCONDITIONAL — a row past the split's end re-reads the split's last row and is replaced by -inf
    //  afterwards — and the two buffers are named, not copied: a branch around a load or a register copy of its
    //  result makes the compiler wait for every outstanding load, vmcnt(0), and the pipeline is gone)
    const long long r_last = r_end - 1;
    auto load_visit = [&](long long r0, P (&buf)[U]) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long long rr = r0 + (long long)u * g.TY;
        buf[u] = *reinterpret_cast<const P*>(base + (rr < r_end ? rr : r_last) * g.C);
      }
    };
    auto fold_visit = [&](long long r0, P (&buf)[U]) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool in = r0 + (long long)u * g.TY < r_end;
#pragma unroll
        for (int e = 0; e < V; e++) buf[u].v[e] = in ? buf[u].v[e] : NEG;
      }
#pragma unroll
      for (int e = 0; e < V; e++) {
        T mc = buf[0].v[e];
#pragma unroll
        for (int u = 1; u < U; u++) mc = max_nn(mc, buf[u].v[e]);
        const T mn = max_nn(m[e], mc);
        const T sh_old = lse_shift(m[e]), sh = lse_shift(mn);
        double acc = s[e] == 0.0 ? 0.0 : s[e] * (double)ex(sh_old - sh);
#pragma unroll
        for (int u = 0; u < U; u++) acc += (double)ex(buf[u].v[e] - sh);
        m[e] = mn;
        s[e] = acc;
      }
    };
    const long long stride = (long long)U * g.TY;
    long long r = r_begin + ty;
    if (r < r_end) {
      P A[U], B[U];
      load_visit(r, A);
      while (true) {
        load_visit(r + stride, B);
        fold_visit(r, A);
        r += stride;
        if (r >= r_end) break;
        load_visit(r + stride, A);
        fold_visit(r, B);
        r += stride;
        if (r >= r_end) break;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < V; e++) { sm_m[t * V + e] = (double)m[e]; sm_s[t * V + e] = s[e]; }
  __syncthreads();
  // fold the row lanes of each column: a tree over ty (fixed order: deterministic)
  int h = 1;
  while (h < g.TY) h <<= 1;
  for (h >>= 1; h >= 1; h >>= 1) {
    if (ty < h && ty + h < g.TY && live) {
      const int o = (ty + h) * g.TX + tx;
#pragma unroll
      for (int e = 0; e < V; e++) {
        const LseSt<T> a{(T)sm_m[t * V + e], sm_s[t * V + e]}, c{(T)sm_m[o * V + e], sm_s[o * V + e]};
        const LseSt<T> r2 = lse_merge(a, c, ex);
        sm_m[t * V + e] = (double)r2.m;
        sm_s[t * V + e] = r2.s;
      }
    }
    __syncthreads();
  }
  if (ty == 0 && live) {
    const long long o = (b * g.nsplit + blockIdx.y) * g.C + col;
#pragma unroll
    for (int e = 0; e < V; e++) { pm[o + e] = sm_m[t * V + e]; ps[o + e] = sm_s[t * V + e]; }
  }
}

// fold the splits of each (batch, column): L lanes per column (a power of two <= 64: the lanes of a column sit in one
// wave), each takes every L-th split, then a butterfly.  MODE 0: out = the log-sum-exp (T).  MODE 1: the pair the
// normalisation needs, in the workspace's first entries of each plane: pm[b*C + c] = shift, ps[b*C + c] = sum.
template <class T, int MODE>
__global__ __launch_bounds__(BLOCK) void col_lse_finish_kernel(T* __restrict__ out, double* __restrict__ pm, double* __restrict__ ps,
                                                              double* __restrict__ fm, double* __restrict__ fs,
                                                              long long ncols_total, long long C, int nsplit, int L) {
  const pthip_dev::ExpCtx<T> ex;
  const long long gid = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const long long colid = gid / L;  // = b * C + c
  const int sub = (int)(gid - colid * L);
  LseSt<T> a{(T)(-__builtin_huge_val()), 0.0};
  if (colid < ncols_total) {
    const long long b = colid / C, c = colid - b * C;
    // eight splits at a time: their loads are in flight together (one at a time, the fold waited ~1 us per split:
    // 32 splits per lane at 1e6 x 10 were 25 of the launch's 48 us)
    constexpr int PF = 8;
    for (int sp0 = sub; sp0 < nsplit; sp0 += PF * L) {
      double vm[PF], vs[PF];
#pragma unroll
      for (int q = 0; q < PF; q++) {
        const int sp = sp0 + q * L;
        const long long o = (b * nsplit + (sp < nsplit ? sp : sub)) * C + c;
        vm[q] = pm[o];
        vs[q] = sp < nsplit ? ps[o] : 0.0;  // (an empty state: s == 0 contributes nothing)
      }
#pragma unroll
      for (int q = 0; q < PF; q++)
        if (sp0 + q * L < nsplit) a = lse_merge(a, LseSt<T>{(T)vm[q], vs[q]}, ex);
    }
  }
  for (int o = L >> 1; o > 0; o >>= 1) {
    LseSt<T> c2;
    c2.m = __shfl_xor(a.m, o);
    c2.s = __shfl_xor(a.s, o);
    a = lse_merge(a, c2, ex);
  }
  if (sub == 0 && colid < ncols_total) {
    if (MODE == 0) {
      out[colid] = lse_value(a);
    } else {
      fm[colid] = (double)lse_shift(a.m);
      fs[colid] = a.s;
    }
  }
}

// out = exp(x - shift) * (1 / sum)   |   (x - shift) - log(sum)      — same thread mapping: a thread's columns never
// change, so its statistics are loaded once
template <class T, int V, bool LOG>
__global__ __launch_bounds__(BLOCK) void col_softmax_norm_kernel(T* __restrict__ out, const T* __restrict__ x, const double* __restrict__ fm,
                                                                const double* __restrict__ fs, ColGeom g) {
  const pthip_dev::ExpCtx<T> ex;
  typedef sm_pack<T, V> P;
  const int t = threadIdx.x;
  const int ty = t / g.TX, tx = t - ty * g.TX;
  const long long col = ((long long)blockIdx.x * g.TX + tx) * V;
  if (ty >= g.TY || col >= g.C) return;
  const long long b = blockIdx.z;
  const long long r_begin = (long long)blockIdx.y * g.rows_per;
  long long r_end = r_begin + g.rows_per;
  if (r_end > g.R) r_end = g.R;
  T sh[V], k[V];
#pragma unroll
  for (int e = 0; e < V; e++) {
    sh[e] = (T)fm[b * g.C + col + e];
    const double ss = fs[b * g.C + col + e];
    k[e] = LOG ? (T)log(ss) : (T)(1.0 / (double)(T)ss);  // (the reference divides by the sum rounded to T)
  }
  const long long step = (long long)g.TY * g.C;
  const long long off = (b * g.R) * g.C + col;
  constexpr int U = 4;
  const long long r_last = r_end - 1;
  auto load_visit = [&](long long r0, P (&buf)[U]) {  // unconditional (clamped): see col_lse_partial_kernel
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long rr = r0 + (long long)u * g.TY;
      buf[u] = *reinterpret_cast<const P*>(x + off + (rr < r_end ? rr : r_last) * g.C);
    }
  };
  auto store_visit = [&](long long r0, P (&buf)[U]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long rr = r0 + (long long)u * g.TY;
#pragma unroll
      for (int e = 0; e < V; e++) buf[u].v[e] = LOG ? ((buf[u].v[e] - sh[e]) - k[e]) : (ex(buf[u].v[e] - sh[e]) * k[e]);
      if (rr < r_end) *reinterpret_cast<P*>(out + off + rr * g.C) = buf[u];
    }
  };
  const long long stride = (long long)U * g.TY;
  long long r = r_begin + ty;
  if (r >= r_end) return;
  P A[U], B[U];
  load_visit(r, A);
  while (true) {  // the next visit's loads are in flight while this one's exps and stores run
    load_visit(r + stride, B);
    store_visit(r, A);
    r += stride;
    if (r >= r_end) break;
    load_visit(r + stride, A);
    store_visit(r, B);
    r += stride;
    if (r >= r_end) break;
  }
}

// the geometry both passes share
template <class T>
ColGeom col_geometry(long long batch, long long R, long long C, int V, long long* ctiles) {
  ColGeom g{};
  g.R = R;
  g.C = C;
  const long long CP = C / V;
  if (CP <= BLOCK) {
    g.TX = (int)CP;
    g.TY = BLOCK / g.TX;
    *ctiles = 1;
  } else {
    g.TX = 64;
    g.TY = BLOCK / 64;
    *ctiles = (CP + 63) / 64;
  }
  // ~4 workgroups per CU in all (each block ends in an LDS tree + one (m, s) pair per column for the finish to fold:
  // per-block cost, so few fat blocks); a split keeps at least 16 rows per row lane
  // measured (profiles/r7_lse_kernels.md): one whole-row tile (narrow C) is best with 2 workgroups per CU, 64-pack tiles with 8
  static const int per_cu_env = [] { const char* e = getenv("PTHIP_COL_WG_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
  const int per_cu = per_cu_env ? per_cu_env : (*ctiles > 1 ? 8 : 2);
  long long want = ((long long)pthip::kNumCU * per_cu + *ctiles * batch - 1) / (*ctiles * batch);
  const long long most = (R + (long long)g.TY * 16 - 1) / ((long long)g.TY * 16);
  if (want > most) want = most;
  if (want < 1) want = 1;
  if (want > 4096) want = 4096;
  g.rows_per = (R + want - 1) / want;
  g.nsplit = (int)((R + g.rows_per - 1) / g.rows_per);
  return g;
}

template <class T> int col_pack_width(long long C, const void* x, const void* out) {
  constexpr int VW = 16 / (int)sizeof(T);
  const bool al = ((uintptr_t)x % 16) == 0 && (out == nullptr || ((uintptr_t)out % 16) == 0);
  return (C % VW == 0 && al) ? VW : 1;
}

template <class T>
size_t colstat_ws_bytes(long long batch, long long R, long long C) {
  long long ct;
  size_t most = 0;
  for (int V : {1, 16 / (int)sizeof(T)}) {
    if (C % V) continue;
    const ColGeom g = col_geometry<T>(batch, R, C, V, &ct);
    const size_t b = (size_t)batch * (size_t)g.nsplit * (size_t)C * 8 * 2 + (size_t)batch * (size_t)C * 8 * 2;
    if (b > most) most = b;
  }
  return most + 256;
}

template <class T>
int colstat_typed(int mode, int log_, long long batch, long long R, long long C, const void* x, void* out, void* ws, size_t ws_bytes) {
  if (batch == 0 || R == 0 || C == 0) return 0;
  if (ws_bytes < colstat_ws_bytes<T>(batch, R, C)) return pthip::set_error("pthip_*_cols: workspace too small (%zu)", ws_bytes);
  hipStream_t st = pthip::ctx().stream;
  const int V = col_pack_width<T>(C, x, mode == 0 ? nullptr : out);
  long long ctiles;
  const ColGeom g = col_geometry<T>(batch, R, C, V, &ctiles);
  double* pm = (double*)ws;
  double* ps = pm + (size_t)batch * g.nsplit * C;
  double* fm = ps + (size_t)batch * g.nsplit * C;
  double* fs = fm + (size_t)batch * C;
  const dim3 grid((unsigned)ctiles, (unsigned)g.nsplit, (unsigned)batch);
  constexpr int VW = 16 / (int)sizeof(T);
  if (V == VW)
    PTHIP_KLAUNCH((col_lse_partial_kernel<T, VW, (sizeof(T) == 8 ? 8 : 4)>), grid, dim3(BLOCK), 0, st, pm, ps, (const T*)x, g);
  else
    PTHIP_KLAUNCH((col_lse_partial_kernel<T, 1, 8>), grid, dim3(BLOCK), 0, st, pm, ps, (const T*)x, g);
  int r = pthip::post_launch("col_lse_partial");
  if (r) return r;
  const long long ncols = batch * C;
  int L = 64;
  while (L > 1 && (ncols * L > 32768 || L > g.nsplit)) L >>= 1;
  const unsigned fgrid = (unsigned)((ncols * L + BLOCK - 1) / BLOCK);
  if (mode == 0) {
    PTHIP_KLAUNCH((col_lse_finish_kernel<T, 0>), dim3(fgrid), dim3(BLOCK), 0, st, (T*)out, pm, ps, fm, fs, ncols, (long long)C, g.nsplit, L);
    return pthip::post_launch("col_lse_finish");
  }
  PTHIP_KLAUNCH((col_lse_finish_kernel<T, 1>), dim3(fgrid), dim3(BLOCK), 0, st, (T*)nullptr, pm, ps, fm, fs, ncols, (long long)C, g.nsplit, L);
  if ((r = pthip::post_launch("col_lse_finish"))) return r;
#define LAUNCH_NORM(VV)                                                                                                              \
  do {                                                                                                                               \
    if (log_)                                                                                                                        \
      PTHIP_KLAUNCH((col_softmax_norm_kernel<T, VV, true>), grid, dim3(BLOCK), 0, st, (T*)out, (const T*)x, (const double*)fm, (const double*)fs, g); \
    else                                                                                                                             \
      PTHIP_KLAUNCH((col_softmax_norm_kernel<T, VV, false>), grid, dim3(BLOCK), 0, st, (T*)out, (const T*)x, (const double*)fm, (const double*)fs, g); \
  } while (0)
  if (V == VW) LAUNCH_NORM(VW); else LAUNCH_NORM(1);
#undef LAUNCH_NORM
  return pthip::post_launch("col_softmax_norm");
}

template <class T>
int lse_rows_typed(long long rows, long long cols, const void* x, void* out) {
  if (rows == 0) return 0;
  if (cols <= 0) return pthip::set_error("pthip_logsumexp_rows: empty reduced axis");
  hipStream_t st = pthip::ctx().stream;
  constexpr int VW = 16 / (int)sizeof(T);
  if (cols <= 32) {
    const unsigned grid = (unsigned)((rows + BLOCK - 1) / BLOCK);
    size_t lds = (size_t)BLOCK * ((size_t)cols | 1) * sizeof(T);
    static const size_t lds_min = [] { const char* e = getenv("PTHIP_LSE_SMALL_LDS_MIN"); return e ? (size_t)atol(e) : (size_t)0; }();
    if (lds < lds_min) lds = lds_min;  // (occupancy experiment: fewer resident workgroups per CU)
    const unsigned magic = (unsigned)(((1u << 20) + (unsigned)cols - 1) / (unsigned)cols);
    // 8-byte staging here: measured (profiles/r8_lse_small_staging.txt) the 16-byte form that wins 15 % in the softmax
    // kernel (39.4 -> 33.6 us at 1e6 x 10, reads AND writes staged) loses 15 % in this read-only one (25.3 -> 29.4 us cold)
    static const int vec_env = [] { const char* e = getenv("PTHIP_LSE_SMALL_VEC"); return e ? atoi(e) : 0; }();
    const int vec = vec_env && ((uintptr_t)x % 16) == 0 && ((BLOCK * cols * sizeof(T)) % 16) == 0;
    static bool lds_raised = false;  // (up to 256 x 33 x 8 = 67.6 KB: beyond the 64 KB a launch may ask for by default)
    if (lds > 48 * 1024 && !lds_raised) {
      PTHIP_CHECK(hipFuncSetAttribute((const void*)lse_rows_small_kernel<T, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      lds_raised = true;
    }
    if (cols <= 8)
      PTHIP_KLAUNCH((lse_rows_small_kernel<T, 8>), dim3(grid), dim3(BLOCK), lds, st, (T*)out, (const T*)x, rows, (int)cols, magic, vec);
    else if (cols <= 16)
      PTHIP_KLAUNCH((lse_rows_small_kernel<T, 16>), dim3(grid), dim3(BLOCK), lds, st, (T*)out, (const T*)x, rows, (int)cols, magic, vec);
    else
      PTHIP_KLAUNCH((lse_rows_small_kernel<T, 32>), dim3(grid), dim3(BLOCK), lds, st, (T*)out, (const T*)x, rows, (int)cols, magic, vec);
    return pthip::post_launch("logsumexp(thread per row, LDS-staged)");
  }
  long long blocks = (rows + 3) / 4;
  const bool packs = cols % VW == 0 && ((uintptr_t)x % 16) == 0;
  {
    long long wb = blocks;
    static const int lw_per_cu = [] { const char* e = getenv("PTHIP_LSE_WAVE_WG_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 3; }();
    const long long wcap = (long long)pthip::kNumCU * lw_per_cu;
    if (wb > wcap) wb = wcap;
#define LAUNCH_LSE(VPL, V)                                                                                                \
  do {                                                                                                                    \
    PTHIP_KLAUNCH((lse_rows_wave_kernel<T, VPL, V>), dim3((unsigned)wb), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (int)cols); \
    return pthip::post_launch("logsumexp(wave per row, registers)");                                                      \
  } while (0)
    if (packs) {
      if (cols <= 64 * 2 * VW) LAUNCH_LSE(2, VW);
      if (cols <= 64 * 4 * VW) LAUNCH_LSE(4, VW);
      if (cols <= 64 * 8 * VW) LAUNCH_LSE(8, VW);
      if (cols <= 64 * 16 * VW) LAUNCH_LSE(16, VW);
    } else {
      if (cols <= 64 * 4) LAUNCH_LSE(4, 1);
      if (cols <= 64 * 16) LAUNCH_LSE(16, 1);
    }
#undef LAUNCH_LSE
  }
  // longer rows: streamed
  static const int rows_per_cu = [] { const char* e = getenv("PTHIP_LSE_ROWS_WG_PER_CU"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
  const long long cap = (long long)pthip::kNumCU * rows_per_cu;
  if (blocks > cap) blocks = cap;
  constexpr int U = 4;
  if (packs) {
    const int G = (int)((cols + 64LL * VW * U - 1) / (64LL * VW * U));
    PTHIP_KLAUNCH((lse_rows_stream_kernel<T, VW, U>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (long long)cols, G);
  } else {
    const int G = (int)((cols + 64LL * U - 1) / (64LL * U));
    PTHIP_KLAUNCH((lse_rows_stream_kernel<T, 1, U>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, (T*)out, (const T*)x, rows, (long long)cols, G);
  }
  return pthip::post_launch("logsumexp(wave-streamed rows)");
}

}  // namespace

extern "C" int pthip_softmax(int dtype, int log_, int64_t rows, int64_t cols, const void* x, void* out) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64) return softmax_typed<double>(log_, rows, cols, x, out);
  if (dtype == PTHIP_F32) return softmax_typed<float>(log_, rows, cols, x, out);
  return pthip::set_error("pthip_softmax: dtype %d not supported (float32/float64 only)", dtype);
}

// ---- round 6: log-sum-exp over the last / a middle axis, softmax over a middle axis (contiguous operands) ----------
extern "C" int64_t pthip_logsumexp_rows_max(int dtype) { return (dtype == PTHIP_F64 || dtype == PTHIP_F32) ? (int64_t)1 << 40 : 0; }  // (rows are streamed: any length)

extern "C" int pthip_logsumexp_rows(int dtype, int64_t rows, int64_t cols, const void* x, void* out) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64) return lse_rows_typed<double>(rows, cols, x, out);
  if (dtype == PTHIP_F32) return lse_rows_typed<float>(rows, cols, x, out);
  return pthip::set_error("pthip_logsumexp_rows: dtype %d not supported (float32/float64 only)", dtype);
}

extern "C" size_t pthip_colstat_workspace(int dtype, int64_t batch, int64_t R, int64_t C) {
  if (batch <= 0 || R <= 0 || C <= 0) return 256;
  return dtype == PTHIP_F32 ? colstat_ws_bytes<float>(batch, R, C) : colstat_ws_bytes<double>(batch, R, C);
}

extern "C" int pthip_logsumexp_cols(int dtype, int64_t batch, int64_t R, int64_t C, const void* x, void* out, void* ws, size_t ws_bytes) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64) return colstat_typed<double>(0, 0, batch, R, C, x, out, ws, ws_bytes);
  if (dtype == PTHIP_F32) return colstat_typed<float>(0, 0, batch, R, C, x, out, ws, ws_bytes);
  return pthip::set_error("pthip_logsumexp_cols: dtype %d not supported (float32/float64 only)", dtype);
}

extern "C" int pthip_softmax_cols(int dtype, int log_, int64_t batch, int64_t R, int64_t C, const void* x, void* out, void* ws,
                                  size_t ws_bytes) {
  PTHIP_REQUIRE_INIT();
  if (dtype == PTHIP_F64) return colstat_typed<double>(1, log_, batch, R, C, x, out, ws, ws_bytes);
  if (dtype == PTHIP_F32) return colstat_typed<float>(1, log_, batch, R, C, x, out, ws, ws_bytes);
  return pthip::set_error("pthip_softmax_cols: dtype %d not supported (float32/float64 only)", dtype);
}
