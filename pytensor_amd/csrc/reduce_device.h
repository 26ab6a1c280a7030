// reduce_device.h — wave64 / workgroup reduction primitives for gfx950.
// Included by the AOT reduce kernels (reduce.hip) AND prepended verbatim to the
// hiprtc-generated fused Elemwise+CAReduce kernels (pytensor_amd/codegen.py), so it
// must not include anything.
//
// Shape of a CDNA4 reduction: per-thread partial accumulators (4-way ILP) →
// wave64 butterfly with DPP/ds_swizzle-backed __shfl_xor (no LDS traffic) →
// one LDS slot per wave → first wave finishes.  No atomics: results are
// run-to-run deterministic.
#pragma once

namespace pthip_dev {

struct OpAdd {
  template <class T> static __device__ __forceinline__ T apply(T a, T b) { return a + b; }
  template <class T> static __device__ __forceinline__ T identity() { return T(0); }
};
struct OpMul {
  template <class T> static __device__ __forceinline__ T apply(T a, T b) { return a * b; }
  template <class T> static __device__ __forceinline__ T identity() { return T(1); }
};
template <class T> struct Limits;
template <> struct Limits<double> {
  static __device__ __forceinline__ double lowest() { return -__builtin_huge_val(); }
  static __device__ __forceinline__ double highest() { return __builtin_huge_val(); }
};
template <> struct Limits<float> {
  static __device__ __forceinline__ float lowest() { return -__builtin_huge_valf(); }
  static __device__ __forceinline__ float highest() { return __builtin_huge_valf(); }
};
template <> struct Limits<long long> {
  static __device__ __forceinline__ long long lowest() { return (-0x7fffffffffffffffLL - 1); }
  static __device__ __forceinline__ long long highest() { return 0x7fffffffffffffffLL; }
};
template <> struct Limits<int> {
  static __device__ __forceinline__ int lowest() { return (-0x7fffffff - 1); }
  static __device__ __forceinline__ int highest() { return 0x7fffffff; }
};
template <> struct Limits<short> {
  static __device__ __forceinline__ short lowest() { return (short)-32768; }
  static __device__ __forceinline__ short highest() { return (short)32767; }
};
template <> struct Limits<signed char> {
  static __device__ __forceinline__ signed char lowest() { return (signed char)-128; }
  static __device__ __forceinline__ signed char highest() { return (signed char)127; }
};
template <> struct Limits<unsigned char> {
  static __device__ __forceinline__ unsigned char lowest() { return 0; }
  static __device__ __forceinline__ unsigned char highest() { return 255; }
};
template <> struct Limits<unsigned short> {
  static __device__ __forceinline__ unsigned short lowest() { return 0; }
  static __device__ __forceinline__ unsigned short highest() { return 65535; }
};
template <> struct Limits<unsigned int> {
  static __device__ __forceinline__ unsigned int lowest() { return 0u; }
  static __device__ __forceinline__ unsigned int highest() { return 0xffffffffu; }
};
template <> struct Limits<unsigned long long> {
  static __device__ __forceinline__ unsigned long long lowest() { return 0ull; }
  static __device__ __forceinline__ unsigned long long highest() { return 0xffffffffffffffffull; }
};
template <> struct Limits<_Float16> {
  static __device__ __forceinline__ _Float16 lowest() { return (_Float16)(-__builtin_huge_valf()); }
  static __device__ __forceinline__ _Float16 highest() { return (_Float16)__builtin_huge_valf(); }
};
template <> struct Limits<bool> {
  static __device__ __forceinline__ bool lowest() { return false; }
  static __device__ __forceinline__ bool highest() { return true; }
};

// NumPy's maximum/minimum propagate NaN (np.maximum.reduce); so do these.
struct OpMax {
  template <class T> static __device__ __forceinline__ T apply(T a, T b) {
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
  }
  template <class T> static __device__ __forceinline__ T identity() { return Limits<T>::lowest(); }
};
struct OpMin {
  template <class T> static __device__ __forceinline__ T apply(T a, T b) {
    return (a != a) ? a : ((b != b) ? b : (a < b ? a : b));
  }
  template <class T> static __device__ __forceinline__ T identity() { return Limits<T>::highest(); }
};
struct OpAnd {
  template <class T> static __device__ __forceinline__ T apply(T a, T b) { return a & b; }
  template <class T> static __device__ __forceinline__ T identity() { return (T)~(T)0; }
};
template <> __device__ __forceinline__ bool OpAnd::apply<bool>(bool a, bool b) { return a && b; }
template <> __device__ __forceinline__ bool OpAnd::identity<bool>() { return true; }
struct OpOr {
  template <class T> static __device__ __forceinline__ T apply(T a, T b) { return a | b; }
  template <class T> static __device__ __forceinline__ T identity() { return T(0); }
};
template <> __device__ __forceinline__ bool OpOr::apply<bool>(bool a, bool b) { return a || b; }
struct OpXor {
  template <class T> static __device__ __forceinline__ T apply(T a, T b) { return a ^ b; }
  template <class T> static __device__ __forceinline__ T identity() { return T(0); }
};
template <> __device__ __forceinline__ bool OpXor::apply<bool>(bool a, bool b) { return a != b; }

// ---- cross-lane exchange for any 1/2/4-byte T and any T of whole 4-byte words (8: double, 16: a pt_lse<double>) ----
template <class T> __device__ __forceinline__ T shfl_xor_any(T v, int mask) {
  if constexpr (sizeof(T) >= 4 && sizeof(T) % 4 == 0) {
    constexpr int NW = sizeof(T) / 4;
    union { T t; int i[NW]; } u;
    u.t = v;
#pragma unroll
    for (int w = 0; w < NW; w++) u.i[w] = __shfl_xor(u.i[w], mask, 64);
    return u.t;
  } else {
    union { T t; int i; } u;
    u.i = 0;
    u.t = v;
    u.i = __shfl_xor(u.i, mask, 64);
    return u.t;
  }
}

// ---- log-sum-exp as a reduction: the running pair (m, s) with sum_i exp(x_i) = s * exp(m), m = max_i x_i ----
// One pass, one exp per element, never an overflow: what `log(sum(exp(x)))` (pytensor/tensor/math.py logsumexp,
// special.py:102 LogSumExp; stabilised by the reference's rewrites into max + log(sum(exp(x - max))): two reductions
// and an N-sized intermediate) is on a machine that would rather read x once.  NaN propagates; -inf terms add nothing;
// a +inf term gives +inf.  value() = log(s) + m.
static __device__ __forceinline__ double lse_exp(double x) { return exp(x); }
static __device__ __forceinline__ float lse_exp(float x) { return expf(x); }
static __device__ __forceinline__ double lse_log(double x) { return log(x); }
static __device__ __forceinline__ float lse_log(float x) { return logf(x); }
template <class T> struct pt_lse {
  T m, s;
  __device__ __forceinline__ T value() const { return lse_log(s) + m; }
  template <class U> explicit __device__ __forceinline__ operator U() const { return (U)value(); }
};
struct OpLse {
  // the state of an empty sum: m = -inf, s = 0 (value: log(0) + -inf = -inf, the reduction's identity)
  template <class P> static __device__ __forceinline__ P identity() { return P{-Limits<decltype(P().m)>::highest(), 0}; }
  // one more term x (one exp): d = x - m; x == m (equal infinities included) adds exactly 1
  template <class T> static __device__ __forceinline__ pt_lse<T> push(pt_lse<T> a, T x) {
    const T d = x - a.m;
    if (x == a.m) {
      a.s += T(1);
    } else {
      const T e = lse_exp(d > T(0) ? -d : d);
      a.s = d > T(0) ? a.s * e + T(1) : a.s + e;  // (NaN: both comparisons false, s + NaN)
    }
    a.m = (x != x) ? x : ((a.m != a.m) ? a.m : (x > a.m ? x : a.m));
    return a;
  }
  // two partial states (lanes, thread rows, splits)
  template <class T> static __device__ __forceinline__ pt_lse<T> apply(pt_lse<T> a, pt_lse<T> b) {
    const T M = (a.m != a.m) ? a.m : ((b.m != b.m) ? b.m : (a.m > b.m ? a.m : b.m));
    const T fa = (a.m == M) ? T(1) : lse_exp(a.m - M);  // (both -inf: M == m, factor 1, s stays 0)
    const T fb = (b.m == M) ? T(1) : lse_exp(b.m - M);
    return pt_lse<T>{M, a.s * fa + b.s * fb};
  }
};

template <class Op, class T> __device__ __forceinline__ T wave_reduce(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = Op::apply(v, shfl_xor_any(v, off));
  return v;
}

// All threads of the block must call. Result valid in every thread of wave 0
// (and broadcast to all threads when BCAST).  `smem` needs BLOCK/64 slots of T.
template <class Op, class T, int BLOCK, bool BCAST = false>
__device__ __forceinline__ T block_reduce(T v, T* smem) {
  constexpr int NW = BLOCK / 64;
  v = wave_reduce<Op>(v);
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();  // protect smem reuse across consecutive calls
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = Op::template identity<T>();
  if (BCAST || wid == 0) {
#pragma unroll
    for (int w = 0; w < NW; w++) r = Op::apply(r, smem[w]);
  }
  return r;
}

}  // namespace pthip_dev
