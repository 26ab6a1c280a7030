// index.hip — bit-exact data movement: strided copy / broadcast fill, row gather,
// row scatter (set / add).
//
// Reference semantics: Subtensor / IncSubtensor / AdvancedSubtensor /
// AdvancedIncSubtensor (pytensor/tensor/subtensor.py:868,1441,1932,2275), Alloc
// (pytensor/tensor/basic.py:1545), Join (2405), DeepCopyOp (compile/ops.py:121).
// All HBM-bound byte movers: coalesce along the innermost contiguous axis, 8/16-byte
// lanes where alignment allows; no arithmetic except the scatter-add.
#include "common.h"
#include "reduce_device.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAXD = 6;

struct CopyDesc {
  long long shape[MAXD];
  long long dstr[MAXD];
  long long sstr[MAXD];
  int ndim;
};

template <class T>
__global__ __launch_bounds__(BLOCK) void copy_strided_kernel(T* __restrict__ dst,
                                                            const T* __restrict__ src,
                                                            CopyDesc d, long long n) {
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * BLOCK) {
    long long rem = i, od = 0, os = 0;
#pragma unroll
    for (int k = MAXD - 1; k >= 0; k--) {
      if (k < d.ndim) {
        long long q = rem / d.shape[k];
        long long c = rem - q * d.shape[k];
        rem = q;
        od += c * d.dstr[k];
        os += c * d.sstr[k];
      }
    }
    dst[od] = src[os];
  }
}

// flat contiguous copy / scalar fill fast paths
template <class T>
__global__ __launch_bounds__(BLOCK) void fill_kernel(T* __restrict__ dst, const T* __restrict__ src,
                                                    long long n) {
  const T v = src[0];
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * BLOCK)
    dst[i] = v;
}

// B (K x N, any strides) -> the MFMA-operand order of the generated skinny-product kernels
// (codegen.dot_epilogue_source): Bp[N/16][K/4][16][4], Bp[ct][k4][j][q] = B[4*k4 + q][16*ct + j],
// zero-padded to multiples of 16 in both extents.  One thread per destination element: writes
// fully coalesced, reads 16-element row segments (a once-per-evaluation repack of a loop constant).
template <typename T>
__global__ void pack_b16_kernel(T* __restrict__ out, const T* __restrict__ B, long long K,
                                long long N, long long Kp, long long n, long long s0, long long s1) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const long long q = i & 3, j = (i >> 2) & 15, r = i >> 6;
    const long long k4 = r % (Kp >> 2), ct = r / (Kp >> 2);
    const long long k = 4 * k4 + q, c = 16 * ct + j;
    out[i] = (k < K && c < N) ? B[k * s0 + c * s1] : (T)0;
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void take_rows_kernel(T* __restrict__ out,
                                                         const T* __restrict__ x,
                                                         const long long* __restrict__ idx,
                                                         long long n_idx, long long inner,
                                                         long long n_rows, long long sx0,
                                                         int* status) {
  const long long n = n_idx * inner;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * BLOCK) {
    long long r = i / inner, c = i - r * inner;
    long long j = idx[r];
    if (j < 0) j += n_rows;
    if (j < 0 || j >= n_rows) {
      atomicOr(status, 1);  // IndexError
      continue;
    }
    out[i] = x[j * sx0 + c];
  }
}

// ---- scatter: set ------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void scatter_winner_kernel(long long* __restrict__ winner,
                                                              const long long* __restrict__ idx,
                                                              long long n_idx, long long n_rows,
                                                              int* status) {
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n_idx;
       i += (long long)gridDim.x * BLOCK) {
    long long j = idx[i];
    if (j < 0) j += n_rows;
    if (j < 0 || j >= n_rows) {
      atomicOr(status, 1);
      continue;
    }
    atomicMax((long long*)&winner[j], i);
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void scatter_set_kernel(T* __restrict__ out,
                                                           const long long* __restrict__ winner,
                                                           const long long* __restrict__ idx,
                                                           const T* __restrict__ y, long long n_idx,
                                                           long long inner, long long n_rows,
                                                           long long ys0) {
  const long long n = n_idx * inner;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * BLOCK) {
    long long r = i / inner, c = i - r * inner;
    long long j = idx[r];
    if (j < 0) j += n_rows;
    if (j < 0 || j >= n_rows) continue;
    if (winner[j] == r) out[j * inner + c] = y[r * ys0 + c];
  }
}

// ---- scatter: add, deterministic for few bins ------------------------------------------
// out[j*inner + c] += sum_{i: idx[i]==j} y[i*ys0 + c], summed in increasing i.
// "Compare-scan": the block stages a chunk of (idx, y) in LDS; thread t owns output bin
// t (= j*inner+c) and scans the whole chunk with LDS broadcast reads.  Work is
// n_idx * n_bins compares — cheap while n_bins <= 512 — and every partial is formed in
// index order; chunk partials [nchunk][n_bins] are then combined in chunk order
// (fixed-order second pass).  Bit-reproducible run to run, unlike atomics.
constexpr int SC_CHUNK = 2048;
constexpr int SC_MAXBINS = 256;

// nb_pad = power of two >= n_bins (<= 256); thread t owns bin t % nb_pad and scans the
// slice t / nb_pad of every staged chunk (slices are contiguous index ranges, so each
// partial is still an in-order sum).  Partials: [(block*parts + slice)][n_bins].
template <class T>
__global__ __launch_bounds__(BLOCK) void scatter_add_scan_kernel(
    T* __restrict__ part, const long long* __restrict__ idx, const T* __restrict__ y,
    long long n_idx, long long inner, long long n_rows, long long ys0, long long n_bins,
    int nb_pad, long long per_block, int* status) {
  __shared__ int s_idx[SC_CHUNK];
  __shared__ T s_y[SC_CHUNK];
  const long long i0 = (long long)blockIdx.x * per_block;
  long long i1 = i0 + per_block;
  if (i1 > n_idx) i1 = n_idx;
  const int parts = BLOCK / nb_pad;
  const int bin = threadIdx.x & (nb_pad - 1), slice = threadIdx.x / nb_pad;
  const int jb = (int)(bin / inner), cb = (int)(bin % inner);
  const int per_slice = SC_CHUNK / parts;
  T acc = T(0), acc1 = T(0), acc2 = T(0), acc3 = T(0);
  for (long long base = i0; base < i1; base += SC_CHUNK) {
    const int m = (int)((i1 - base) < SC_CHUNK ? (i1 - base) : SC_CHUNK);
    __syncthreads();
    for (int k = threadIdx.x; k < m; k += BLOCK) {
      long long j = idx[base + k];
      if (j < 0) j += n_rows;
      if (j < 0 || j >= n_rows) {
        atomicOr(status, 1);
        j = -1;
      }
      s_idx[k] = (int)j;
      if (inner == 1) s_y[k] = y[(base + k) * ys0];
    }
    __syncthreads();
    const int k0 = slice * per_slice;
    int k1 = k0 + per_slice;
    if (k1 > m) k1 = m;
    if (inner == 1) {
      // 8 staged entries per step: the LDS reads (wave-uniform addresses -> broadcast) are
      // issued together, so their latency is paid once per 8 entries instead of per entry;
      // the adds stay in index order (acc is a single in-order chain: bit-reproducible).
      int k = k0;
      for (; k + 7 < k1; k += 8) {
        int id[8];
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          id[u] = s_idx[k + u];
          v[u] = s_y[k + u];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 4) {  // four fixed interleaved chains (deterministic order)
          acc += (id[u] == jb) ? v[u] : T(0);
          acc1 += (id[u + 1] == jb) ? v[u + 1] : T(0);
          acc2 += (id[u + 2] == jb) ? v[u + 2] : T(0);
          acc3 += (id[u + 3] == jb) ? v[u + 3] : T(0);
        }
      }
      for (; k < k1; k++) {
        const T v = s_y[k];
        acc += (s_idx[k] == jb) ? v : T(0);
      }
    } else if (bin < n_bins) {
      for (int k = k0; k < k1; k++)
        if (s_idx[k] == jb) acc += y[(base + k) * ys0 + cb];
    }
  }
  if (bin < n_bins)
    part[((long long)blockIdx.x * parts + slice) * n_bins + bin] = (acc + acc1) + (acc2 + acc3);
}

// out[b] += sum_k part[k][b] in k order: 16 bins x 16 slices per block, combined in LDS
template <class T>
__global__ __launch_bounds__(BLOCK) void scatter_add_finish_kernel(T* __restrict__ out,
                                                                  const T* __restrict__ part,
                                                                  long long n_bins,
                                                                  long long nblk) {
  __shared__ T red[16][17];
  const int oi = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long b = (long long)blockIdx.x * 16 + oi;
  const long long per = (nblk + 15) / 16;
  long long k0 = sl * per, k1 = k0 + per;
  if (k1 > nblk) k1 = nblk;
  T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
  if (b < n_bins) {
    long long k = k0;
    for (; k + 3 < k1; k += 4) {
      a0 += part[k * n_bins + b];
      a1 += part[(k + 1) * n_bins + b];
      a2 += part[(k + 2) * n_bins + b];
      a3 += part[(k + 3) * n_bins + b];
    }
    for (; k < k1; k++) a0 += part[k * n_bins + b];
  }
  red[sl][oi] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && b < n_bins) {
    T v = red[0][oi];
#pragma unroll
    for (int k = 1; k < 16; k++) v += red[k][oi];
    out[b] += v;
  }
}

// many bins, integer types (and PTHIP_SCATTER_EXACT=0): native atomics — exact for integers; for floating point the
// last bits depend on the order the adds were served in
template <class T>
__global__ __launch_bounds__(BLOCK) void scatter_add_atomic_kernel(
    T* __restrict__ out, const long long* __restrict__ idx, const T* __restrict__ y,
    long long n_idx, long long inner, long long n_rows, long long ys0, int* status) {
  const long long n = n_idx * inner;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n;
       i += (long long)gridDim.x * BLOCK) {
    long long r = i / inner, c = i - r * inner;
    long long j = idx[r];
    if (j < 0) j += n_rows;
    if (j < 0 || j >= n_rows) {
      atomicOr(status, 1);
      continue;
    }
    if constexpr (sizeof(T) == 8 && !__is_integral(T))
      unsafeAtomicAdd((double*)&out[j * inner + c], (double)y[r * ys0 + c]);
    else if constexpr (sizeof(T) == 4 && !__is_integral(T))
      unsafeAtomicAdd((float*)&out[j * inner + c], (float)y[r * ys0 + c]);
    else if constexpr (sizeof(T) == 8)
      atomicAdd((unsigned long long*)&out[j * inner + c], (unsigned long long)y[r * ys0 + c]);
    else
      atomicAdd((int*)&out[j * inner + c], (int)y[r * ys0 + c]);
  }
}

// ---- scatter: add, floating point, ANY number of bins: exact and therefore order-independent -----------------------
// The atomic kernel above gives sums whose last bits depend on the order the hardware served the adds in: two
// evaluations of the same graph differ (found by the seeded regression graphs with 300 groups, tests/golden/glm_fuzz_*).
// Integer addition is associative, so: every addend becomes a 128-bit fixed-point integer in units of 2^-95 of the
// largest |y| (a first pass takes that maximum; 43 bits below the largest value's last bit are kept, anything smaller
// loses bits — an absolute error of 2^-95 max|y| per addend, far inside one rounding of any sum it could matter to;
// 30 bits of headroom: 1e9 addends of the largest magnitude), the bins add their integers with two 64-bit integer
// atomics (low word, then high word + the carry the low add produced: the total is the exact sum mod 2^128 whatever the
// order), and a last pass rounds each bin ONCE to the output type and adds it to what `out` held.  NaN / infinities go
// through per-bin flag bits.  More accurate than the reference's sequential adds (subtensor.py AdvancedIncSubtensor1:
// np.add.at), bit-identical from run to run.
__global__ __launch_bounds__(BLOCK) void scatter_absmax_kernel(const void* __restrict__ y, int f32, long long n_idx, long long inner,
                                                              long long ys0, unsigned long long* __restrict__ maxbits) {
  const long long n = n_idx * inner;
  unsigned long long best = 0;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) {
    const long long r = inner == 1 ? i : i / inner, c = i - r * inner;
    const double v = f32 ? (double)((const float*)y)[r * ys0 + c] : ((const double*)y)[r * ys0 + c];
    const unsigned long long b = (unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull;
    if ((b >> 52) != 0x7ff && b > best) best = b;  // (finite values only; non-negative doubles order like their bits)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long ob = (unsigned long long)__shfl_xor((long long)best, o);
    best = ob > best ? ob : best;
  }
  // one atomic per workgroup (a wave each was 16,000 atomics on one word for a million addends: most of the pass)
  __shared__ unsigned long long s_best[BLOCK / 64];
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < BLOCK / 64; w++) best = s_best[w] > best ? s_best[w] : best;
    if (best) atomicMax(maxbits, best);
  }
}

// one addend as a 128-bit two's-complement integer in units of 2^(Ef - 1075 - 43); false: nothing to add (zero, below the
// window) or a non-finite value (*special: 1 NaN, 2 +inf, 4 -inf)
static __device__ __forceinline__ bool scatter_fixed(double v, int Ef, unsigned long long& qlo, unsigned long long& qhi, unsigned& special) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  int ef = (int)((b >> 52) & 0x7ff);
  unsigned long long m = b & 0x000fffffffffffffull;
  const bool neg = (b >> 63) != 0;
  special = 0;
  if (ef == 0x7ff) {
    special = m ? 1u : (neg ? 4u : 2u);
    return false;
  }
  if (ef) m |= 1ull << 52; else ef = 1;
  if (m == 0) return false;
  // v = m 2^(ef - 1075); unit 2^(Ef - 1075 - 43): q = m 2^(ef - Ef + 43)
  const int sh = ef - Ef + 43;  // <= 43
  if (sh >= 0) {
    qlo = m << sh;
    qhi = sh ? (m >> (64 - sh)) : 0ull;
  } else if (sh > -53) {
    qlo = m >> (-sh);
    qhi = 0;
  } else {
    return false;
  }
  if (neg) {  // two's complement
    qlo = ~qlo + 1ull;
    qhi = ~qhi + (qlo == 0 ? 1ull : 0ull);
  }
  return (qlo | qhi) != 0;
}

// 128-bit add of (qlo, qhi) into (lo[bin], hi[bin]) with two 64-bit integer atomics: exact mod 2^128 in any order
static __device__ __forceinline__ void scatter_add128(unsigned long long* lo, unsigned long long* hi, unsigned long long qlo, unsigned long long qhi) {
  unsigned long long carry = 0;
  if (qlo) {
    const unsigned long long old = atomicAdd(lo, qlo);
    carry = (old + qlo) < old ? 1ull : 0ull;
  }
  if (qhi + carry) atomicAdd(hi, qhi + carry);
}

// LDS != 0: the workgroup first adds its share of the addends into a private table of all bins in LDS (LDS integer
// atomics: no trip to memory, no contention between workgroups) and then adds its non-zero entries to the global table
// — for a few hundred to a few thousand bins the global atomics drop from two per addend to two per (workgroup, bin)
// and no longer queue up on a few hundred addresses (300 bins, 1e6 addends: profiles/r6i_scatter_exact.txt).
template <bool LDS>
__global__ __launch_bounds__(BLOCK) void scatter_add_exact_kernel(unsigned long long* __restrict__ lo, unsigned long long* __restrict__ hi,
                                                                 unsigned* __restrict__ flags, const long long* __restrict__ idx,
                                                                 const void* __restrict__ y, int f32, long long n_idx, long long inner,
                                                                 long long n_rows, long long ys0,
                                                                 const unsigned long long* __restrict__ maxbits, int* status, int n_bins) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sc_smem[];
  unsigned long long* s_lo = (unsigned long long*)sc_smem;
  unsigned long long* s_hi = s_lo + n_bins;
  if constexpr (LDS) {
    for (int b = threadIdx.x; b < 2 * n_bins; b += BLOCK) s_lo[b] = 0;
    __syncthreads();
  }
  const long long n = n_idx * inner;
  int Ef = (int)((*maxbits >> 52) & 0x7ff);
  if (Ef == 0) Ef = 1;  // (largest value subnormal or zero: the subnormal exponent)
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * BLOCK) {
    const long long r = inner == 1 ? i : i / inner, c = i - r * inner;  // (a 64-bit division is ~100 instructions)
    long long j = idx[r];
    if (j < 0) j += n_rows;
    if (j < 0 || j >= n_rows) {
      atomicOr(status, 1);
      continue;
    }
    const long long bin = j * inner + c;
    const double v = f32 ? (double)((const float*)y)[r * ys0 + c] : ((const double*)y)[r * ys0 + c];
    unsigned long long qlo, qhi;
    unsigned special;
    if (!scatter_fixed(v, Ef, qlo, qhi, special)) {
      if (special) atomicOr(&flags[bin], special);
      continue;
    }
    if constexpr (LDS) scatter_add128(&s_lo[bin], &s_hi[bin], qlo, qhi);
    else scatter_add128(&lo[bin], &hi[bin], qlo, qhi);
  }
  if constexpr (LDS) {
    __syncthreads();
    for (int b = threadIdx.x; b < n_bins; b += BLOCK) {
      const unsigned long long l = s_lo[b], h = s_hi[b];
      if (l | h) scatter_add128(&lo[b], &hi[b], l, h);
    }
  }
}

template <class T>
__global__ __launch_bounds__(BLOCK) void scatter_exact_finish_kernel(T* __restrict__ out, const unsigned long long* __restrict__ lo,
                                                                    const unsigned long long* __restrict__ hi,
                                                                    const unsigned* __restrict__ flags, long long n_bins,
                                                                    const unsigned long long* __restrict__ maxbits) {
  const long long bin = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (bin >= n_bins) return;
  const unsigned f = flags[bin];
  double add;
  if (f) {
    add = (f & 1u) || ((f & 2u) && (f & 4u)) ? __longlong_as_double(0x7ff8000000000000ll)
                                            : ((f & 2u) ? __longlong_as_double(0x7ff0000000000000ll) : __longlong_as_double((long long)0xfff0000000000000ull));
  } else {
    unsigned long long l = lo[bin], h = hi[bin];
    // nothing landed here, or what landed cancelled exactly: out keeps its bits.  (Edge, documented in DESIGN §2: an
    // out of -0.0 under addends that cancel exactly stays -0.0 where the reference's sequential np.add.at ends on +0.0.)
    if (l == 0 && h == 0) return;
    const bool neg = (h >> 63) != 0;
    if (neg) {
      l = ~l + 1ull;
      h = ~h + (l == 0 ? 1ull : 0ull);
    }
    // magnitude h:l -> double, round to nearest even once
    const int p = h ? 127 - __clzll((long long)h) : 63 - __clzll((long long)l);  // position of the leading bit
    unsigned long long mant;
    int e2 = 0;
    if (p <= 52) {
      mant = l;
    } else {
      const int s = p - 52;  // bits to drop (1..75)
      unsigned long long kept, half_bit, rest;
      if (s < 64) {
        kept = (l >> s) | (s ? (h << (64 - s)) : 0ull);
        half_bit = (l >> (s - 1)) & 1ull;
        rest = s > 1 ? (l & ((1ull << (s - 1)) - 1ull)) : 0ull;
      } else {
        const int t = s - 64;  // 0..11
        kept = h >> t;
        half_bit = t ? ((h >> (t - 1)) & 1ull) : (l >> 63);
        rest = t ? ((h & ((1ull << (t - 1)) - 1ull)) | l) : (l & 0x7fffffffffffffffull);
      }
      mant = kept + ((half_bit && (rest || (kept & 1ull))) ? 1ull : 0ull);  // (2^53 after the carry is still exact)
      e2 = s;
    }
    int Ef = (int)((*maxbits >> 52) & 0x7ff);
    if (Ef == 0) Ef = 1;
    add = ldexp((double)mant, Ef - 1075 - 43 + e2);
    if (neg) add = -add;
  }
  // `add` is the EXACT sum of the bin's addends rounded once to double; adding it to what `out` already holds (zeros in
  // every gradient graph: the reference scatters into an Alloc of 0) is a second rounding when out != 0, and for a
  // float32 output the result is rounded to double and then to float (double rounding: <= 0.5 ulp32 + 2^-29 ulp32).
  out[bin] = (T)((double)out[bin] + add);
}

// ---- pack: gather up to 16 small contiguous buffers into one staging buffer -----------------
struct PackDesc {
  const unsigned char* src[16];
  long long nbytes[16];
  long long dst_off[16];
};

__global__ __launch_bounds__(BLOCK) void pack_kernel(unsigned char* __restrict__ dst, PackDesc d) {
  const int e = blockIdx.y;
  const long long nb = d.nbytes[e];
  const unsigned char* s = d.src[e];
  unsigned char* o = dst + d.dst_off[e];
  const long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const long long nth = (long long)gridDim.x * BLOCK;
  if ((((uintptr_t)s | (uintptr_t)o) & 7) == 0) {
    const long long nw = nb >> 3;
    for (long long i = tid; i < nw; i += nth)
      ((unsigned long long*)o)[i] = ((const unsigned long long*)s)[i];
    for (long long i = (nw << 3) + tid; i < nb; i += nth) o[i] = s[i];
  } else {
    for (long long i = tid; i < nb; i += nth) o[i] = s[i];
  }
}

inline unsigned grid_for(long long n) {
  long long b = (n + BLOCK - 1) / BLOCK;
  long long cap = (long long)pthip::kNumCU * 8;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <class T>
int copy_typed(int ndim, const int64_t* shape, void* dst, const int64_t* ds, const void* src,
               const int64_t* ss) {
  hipStream_t st = pthip::ctx().stream;
  long long n = 1;
  for (int k = 0; k < ndim; k++) n *= shape[k];
  if (n == 0) return 0;
  // collapse: drop size-1 dims, merge dims contiguous in both
  long long sh[MAXD], d[MAXD], s[MAXD];
  int nd = 0;
  for (int k = 0; k < ndim; k++) {
    if (shape[k] == 1) continue;
    if (nd > 0 && d[nd - 1] == shape[k] * ds[k] && s[nd - 1] == shape[k] * ss[k]) {
      sh[nd - 1] *= shape[k];
      d[nd - 1] = ds[k];
      s[nd - 1] = ss[k];
    } else {
      sh[nd] = shape[k];
      d[nd] = ds[k];
      s[nd] = ss[k];
      nd++;
    }
  }
  if (nd == 0) {
    sh[0] = 1; d[0] = 1; s[0] = 1; nd = 1;
  }
  if (nd == 1 && d[0] == 1 && s[0] == 1) {
    hipError_t e = pthip::memcpy_async(dst, src, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return pthip::check(e, "hipMemcpyAsync(copy_strided)");
    return 0;
  }
  if (nd == 1 && d[0] == 1 && s[0] == 0) {
    PTHIP_KLAUNCH((fill_kernel<T>), dim3(grid_for(n)), dim3(BLOCK), 0, st, (T*)dst, (const T*)src, n);
    return pthip::post_launch("fill");
  }
  if (nd > MAXD) return pthip::set_error("pthip_copy_strided: more than %d non-mergeable dims", MAXD);
  CopyDesc cd{};
  cd.ndim = nd;
  for (int k = 0; k < nd; k++) {
    cd.shape[k] = sh[k];
    cd.dstr[k] = d[k];
    cd.sstr[k] = s[k];
  }
  PTHIP_KLAUNCH((copy_strided_kernel<T>), dim3(grid_for(n)), dim3(BLOCK), 0, st, (T*)dst,
                     (const T*)src, cd, n);
  return pthip::post_launch("copy_strided");
}

}  // namespace

extern "C" {

int pthip_copy_strided(int itemsize, int ndim, const int64_t* shape, void* dst,
                       const int64_t* dst_strides, const void* src, const int64_t* src_strides) {
  PTHIP_REQUIRE_INIT();
  if (ndim > 16) return pthip::set_error("pthip_copy_strided: ndim too large");
  switch (itemsize) {
    case 1: return copy_typed<unsigned char>(ndim, shape, dst, dst_strides, src, src_strides);
    case 2: return copy_typed<unsigned short>(ndim, shape, dst, dst_strides, src, src_strides);
    case 4: return copy_typed<unsigned int>(ndim, shape, dst, dst_strides, src, src_strides);
    case 8: return copy_typed<unsigned long long>(ndim, shape, dst, dst_strides, src, src_strides);
  }
  return pthip::set_error("pthip_copy_strided: unsupported itemsize %d", itemsize);
}

int pthip_pack_b16(int itemsize, int64_t K, int64_t N, const void* B, int64_t sB0, int64_t sB1,
                   void* Bp) {
  PTHIP_REQUIRE_INIT();
  hipStream_t st = pthip::ctx().stream;
  const long long Kp = (K + 15) / 16 * 16, Np = (N + 15) / 16 * 16;
  const long long n = Kp * Np;
  if (n == 0) return 0;
#define LAUNCH(T)                                                                                  \
  PTHIP_KLAUNCH((pack_b16_kernel<T>), dim3(grid_for(n)), dim3(BLOCK), 0, st, (T*)Bp,           \
                     (const T*)B, (long long)K, (long long)N, Kp, n, (long long)sB0, (long long)sB1)
  switch (itemsize) {
    case 4: LAUNCH(unsigned int); break;
    case 8: LAUNCH(unsigned long long); break;
    default: return pthip::set_error("pthip_pack_b16: unsupported itemsize %d", itemsize);
  }
#undef LAUNCH
  return pthip::post_launch("pack_b16");
}

int pthip_take_rows(int itemsize, int64_t n_idx, int64_t inner, const void* x, int64_t n_rows,
                    int64_t sx0, const int64_t* idx, void* out) {
  PTHIP_REQUIRE_INIT();
  hipStream_t st = pthip::ctx().stream;
  long long n = n_idx * inner;
  if (n == 0) return 0;
  int* status = pthip::ctx().status_dev;
#define LAUNCH(T)                                                                                  \
  PTHIP_KLAUNCH((take_rows_kernel<T>), dim3(grid_for(n)), dim3(BLOCK), 0, st, (T*)out,        \
                     (const T*)x, (const long long*)idx, (long long)n_idx, (long long)inner,       \
                     (long long)n_rows, (long long)sx0, status)
  switch (itemsize) {
    case 1: LAUNCH(unsigned char); break;
    case 2: LAUNCH(unsigned short); break;
    case 4: LAUNCH(unsigned int); break;
    case 8: LAUNCH(unsigned long long); break;
    default: return pthip::set_error("pthip_take_rows: unsupported itemsize %d", itemsize);
  }
#undef LAUNCH
  return pthip::post_launch("take_rows");
}

size_t pthip_scatter_rows_workspace(int64_t n_idx, int64_t n_rows, int64_t inner) {
  size_t set_ws = (size_t)n_rows * 8;
  long long n_bins = n_rows * inner;
  size_t add_ws = 0;
  if (n_bins <= SC_MAXBINS) {
    long long nblk = (n_idx + SC_CHUNK - 1) / SC_CHUNK;
    long long cap = (long long)pthip::kNumCU * 8;
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    add_ws = (size_t)nblk * BLOCK * 8;  // parts * n_bins <= BLOCK partial values per block
  } else {
    add_ws = (size_t)n_bins * 24 + 64;  // the exact accumulator: two 64-bit words + a flag word per bin (padded), the maximum
  }
  return set_ws > add_ws ? set_ws : add_ws;
}

int pthip_scatter_rows(int dtype, int inc, int64_t n_idx, int64_t inner, void* out, int64_t n_rows,
                       const int64_t* idx, const void* y, int64_t ys0, void* ws, size_t ws_bytes) {
  PTHIP_REQUIRE_INIT();
  hipStream_t st = pthip::ctx().stream;
  int* status = pthip::ctx().status_dev;
  if (n_idx == 0 || inner == 0) return 0;
  if (ws_bytes < pthip_scatter_rows_workspace(n_idx, n_rows, inner))
    return pthip::set_error("pthip_scatter_rows: workspace too small");
  const int isz = pthip::dtype_size(dtype);
  const long long n = n_idx * inner;
  if (!inc) {
    PTHIP_CHECK(pthip::memset_async(ws, 0xff, (size_t)n_rows * 8, st));  // winner = -1
    PTHIP_KLAUNCH(scatter_winner_kernel, dim3(grid_for(n_idx)), dim3(BLOCK), 0, st,
                       (long long*)ws, (const long long*)idx, (long long)n_idx, (long long)n_rows, status);
#define LAUNCH(T)                                                                               \
  PTHIP_KLAUNCH((scatter_set_kernel<T>), dim3(grid_for(n)), dim3(BLOCK), 0, st, (T*)out,   \
                     (const long long*)ws, (const long long*)idx, (const T*)y, (long long)n_idx, \
                     (long long)inner, (long long)n_rows, (long long)ys0)
    switch (isz) {
      case 1: LAUNCH(unsigned char); break;
      case 2: LAUNCH(unsigned short); break;
      case 4: LAUNCH(unsigned int); break;
      case 8: LAUNCH(unsigned long long); break;
    }
#undef LAUNCH
    return pthip::post_launch("scatter_set");
  }
  const long long n_bins = n_rows * inner;
  if (n_bins <= SC_MAXBINS && (dtype == PTHIP_F64 || dtype == PTHIP_F32 || dtype == PTHIP_I64)) {
    int nb_pad = 1;
    while (nb_pad < n_bins) nb_pad <<= 1;
    const int parts = BLOCK / nb_pad;
    long long nblk = (n_idx + SC_CHUNK - 1) / SC_CHUNK;
    long long cap = (long long)pthip::kNumCU * 8;
    if (nblk > cap) nblk = cap;
    long long per_block = (n_idx + nblk - 1) / nblk;
    per_block = (per_block + SC_CHUNK - 1) / SC_CHUNK * SC_CHUNK;
    nblk = (n_idx + per_block - 1) / per_block;
#define LAUNCH(T)                                                                                 \
  do {                                                                                            \
    PTHIP_KLAUNCH((scatter_add_scan_kernel<T>), dim3((unsigned)nblk), dim3(BLOCK), 0, st,    \
                       (T*)ws, (const long long*)idx, (const T*)y, (long long)n_idx,              \
                       (long long)inner, (long long)n_rows, (long long)ys0, n_bins, nb_pad,       \
                       per_block, status);                                                        \
    PTHIP_KLAUNCH((scatter_add_finish_kernel<T>),                                            \
                       dim3((unsigned)((n_bins + 15) / 16)), dim3(BLOCK), 0, st,                  \
                       (T*)out, (const T*)ws, n_bins, nblk * parts);                              \
  } while (0)
    if (dtype == PTHIP_F64) LAUNCH(double);
    else if (dtype == PTHIP_F32) LAUNCH(float);
    else LAUNCH(long long);
#undef LAUNCH
    return pthip::post_launch("scatter_add_scan");
  }
  static const bool exact = !(getenv("PTHIP_SCATTER_EXACT") && atoi(getenv("PTHIP_SCATTER_EXACT")) == 0);
  if (exact && (dtype == PTHIP_F64 || dtype == PTHIP_F32)) {
    unsigned long long* lo = (unsigned long long*)ws;
    unsigned long long* hi = lo + n_bins;
    unsigned* flags = (unsigned*)(hi + n_bins);
    unsigned long long* maxbits = (unsigned long long*)((char*)ws + (size_t)n_bins * 24);
    PTHIP_CHECK(pthip::memset_async(ws, 0, (size_t)n_bins * 24 + 8, st));
    const int f32 = dtype == PTHIP_F32;
    const unsigned gmax = grid_for(n) < 512u ? grid_for(n) : 512u;
    PTHIP_KLAUNCH(scatter_absmax_kernel, dim3(gmax), dim3(BLOCK), 0, st, y, f32, (long long)n_idx, (long long)inner, (long long)ys0, maxbits);
    constexpr long long LDS_BINS = 8192;  // 16 bytes per bin: 128 KB of LDS
    if (n_bins <= LDS_BINS && n >= 16 * n_bins) {
      auto kl = scatter_add_exact_kernel<true>;
      const size_t sh = (size_t)n_bins * 16;
      static size_t sh_set = 0;
      if (sh > sh_set && sh > 48 * 1024) {
        PTHIP_CHECK(hipFuncSetAttribute((const void*)kl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_BINS * 16)));
        sh_set = LDS_BINS * 16;
      }
      // (one workgroup per CU at most: every workgroup ends with up to n_bins pairs of global atomics)
      long long wgs = (n + 16 * BLOCK - 1) / (16 * BLOCK);
      if (wgs > pthip::kNumCU) wgs = pthip::kNumCU;
      if (wgs < 1) wgs = 1;
      PTHIP_KLAUNCH(kl, dim3((unsigned)wgs), dim3(BLOCK), sh, st, lo, hi, flags, (const long long*)idx, y, f32, (long long)n_idx,
                    (long long)inner, (long long)n_rows, (long long)ys0, (const unsigned long long*)maxbits, status, (int)n_bins);
    } else {
      PTHIP_KLAUNCH(scatter_add_exact_kernel<false>, dim3(grid_for(n)), dim3(BLOCK), 0, st, lo, hi, flags, (const long long*)idx, y, f32,
                    (long long)n_idx, (long long)inner, (long long)n_rows, (long long)ys0, (const unsigned long long*)maxbits, status,
                    (int)(n_bins > 0x7fffffff ? 0 : n_bins));
    }
    const unsigned fg = (unsigned)((n_bins + BLOCK - 1) / BLOCK);
    if (f32)
      PTHIP_KLAUNCH(scatter_exact_finish_kernel<float>, dim3(fg), dim3(BLOCK), 0, st, (float*)out, (const unsigned long long*)lo,
                    (const unsigned long long*)hi, (const unsigned*)flags, n_bins, (const unsigned long long*)maxbits);
    else
      PTHIP_KLAUNCH(scatter_exact_finish_kernel<double>, dim3(fg), dim3(BLOCK), 0, st, (double*)out, (const unsigned long long*)lo,
                    (const unsigned long long*)hi, (const unsigned*)flags, n_bins, (const unsigned long long*)maxbits);
    return pthip::post_launch("scatter_add_exact");
  }
#define LAUNCH(T)                                                                                  \
  PTHIP_KLAUNCH((scatter_add_atomic_kernel<T>), dim3(grid_for(n)), dim3(BLOCK), 0, st,        \
                     (T*)out, (const long long*)idx, (const T*)y, (long long)n_idx,                \
                     (long long)inner, (long long)n_rows, (long long)ys0, status)
  switch (dtype) {
    case PTHIP_F64: LAUNCH(double); break;
    case PTHIP_F32: LAUNCH(float); break;
    case PTHIP_I64: LAUNCH(long long); break;
    case PTHIP_I32: LAUNCH(int); break;
    case PTHIP_U64: LAUNCH(unsigned long long); break;  // wrap-around add: same bits as the signed one
    case PTHIP_U32: LAUNCH(unsigned int); break;
    default: return pthip::set_error("pthip_scatter_rows: unsupported dtype %d for inc", dtype);
  }
#undef LAUNCH
  return pthip::post_launch("scatter_add_atomic");
}

int pthip_pack(int n, const void* const* srcs, const int64_t* nbytes, const int64_t* dst_offsets,
               void* dst) {
  PTHIP_REQUIRE_INIT();
  if (n <= 0) return 0;
  if (n > 16) return pthip::set_error("pthip_pack: at most 16 buffers per call");
  PackDesc d{};
  long long mx = 0;
  for (int i = 0; i < n; i++) {
    d.src[i] = (const unsigned char*)srcs[i];
    d.nbytes[i] = nbytes[i];
    d.dst_off[i] = dst_offsets[i];
    if (nbytes[i] > mx) mx = nbytes[i];
  }
  long long bx = (mx / 8 + BLOCK - 1) / BLOCK;
  if (bx < 1) bx = 1;
  if (bx > 64) bx = 64;
  PTHIP_KLAUNCH(pack_kernel, dim3((unsigned)bx, (unsigned)n), dim3(BLOCK), 0,
                     pthip::ctx().stream, (unsigned char*)dst, d);
  return pthip::post_launch("pack");
}

}  // extern "C"
