// random.hip — RandomVariable draws on the device from a counter-based generator.
//
// Reference: RandomVariable.perform (pytensor/tensor/random/op.py: rng_fn(rng, *params, size) on
// a numpy.random.Generator, which is returned advanced as the node's first output); the
// distributions are the classes of pytensor/tensor/random/basic.py (UniformRV 83, NormalRV 239,
// GammaRV 418, ...).  SURVEY §8f row 4: NumPy's samplers are sequential (ziggurat / rejection
// loops consuming a data-dependent number of raw draws from one stream), so a parallel device
// sampler cannot reproduce the reference's numbers; parity is distributional, and bit-exact
// against the CPU restatement of *this* algorithm (oracle/philox_ref.py).
//
// Generator: Philox4x64-10 (Salmon et al., SC'11) exactly as numpy.random.Philox implements it
// (256-bit counter, 128-bit key; a fresh generator increments the counter, then encrypts it), so
// the state travels as a real numpy Generator(Philox) on the host side.
//  * uniform: output i is word i%4 of block counter+1+i/4, (w >> 11) * 2^-53 — the numbers
//    Generator(Philox(key, counter)).random(n) returns; low + (high-low)*u without contraction.
//  * every other distribution: element i owns block counter+1+i; rejection samplers take further
//    blocks from derived keys (key1 + attempt, key0 ^ substream constant), never from the
//    neighbours' counters.  u in (0,1): ((w >> 12) + 0.5) * 2^-52; normals by Box-Muller.
// The caller advances the counter by the number of blocks consumed: ceil(n/4) or n.
#include "common.h"

namespace {

typedef unsigned long long u64;

constexpr int BLOCK = 256;

enum Dist {
  D_UNIFORM = 0, D_NORMAL, D_HALFNORMAL, D_LOGNORMAL, D_EXPONENTIAL, D_LAPLACE, D_LOGISTIC, D_CAUCHY,
  D_HALFCAUCHY, D_GUMBEL, D_WEIBULL, D_PARETO, D_TRIANGULAR, D_GAMMA, D_BETA, D_INVGAMMA, D_STUDENT_T,
  D_BERNOULLI, D_GEOMETRIC, D_POISSON, D_INTEGERS, D_BINOMIAL, D_NEGBINOMIAL, D_WALD, D_TRUNCEXPON, D_GENGAMMA,
  D_BETABINOMIAL, D_VONMISES, D_HYPERGEOMETRIC, D_COUNT
};

struct RandArgs {
  const void* p[3];
  int dt[3];
  long long st[3];
  u64 key[2];
  u64 ctr[4];
};

__device__ __forceinline__ void philox_block(u64 c0, u64 c1, u64 c2, u64 c3, u64 k0, u64 k1, u64 out[4]) {
  const u64 M0 = 0xD2E7470EE14C6C93ull, M1 = 0xCA5A826395121157ull;
  const u64 W0 = 0x9E3779B97F4A7C15ull, W1 = 0xBB67AE8584CAA73Bull;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    if (r) { k0 += W0; k1 += W1; }
    const u64 hi0 = __umul64hi(M0, c0), lo0 = M0 * c0;
    const u64 hi1 = __umul64hi(M1, c2), lo1 = M1 * c2;
    const u64 n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// block (counter + 1 + off) under key (k0 ^ substream-constant, k1 + attempt)
__device__ __forceinline__ void draw_block(const RandArgs& a, u64 off, unsigned sub, unsigned attempt, u64 out[4]) {
  u64 c0 = a.ctr[0], c1 = a.ctr[1], c2 = a.ctr[2], c3 = a.ctr[3];
  // counter + (off + 1): off + 1 may itself wrap only at 2^64, handled as two additions
  u64 t = c0 + off; u64 carry = t < c0; c0 = t;
  t = c0 + 1; carry += t < c0; c0 = t;
  t = c1 + carry; carry = t < c1; c1 = t;
  t = c2 + carry; carry = t < c2; c2 = t;
  c3 += carry;
  philox_block(c0, c1, c2, c3, a.key[0] ^ (0x9E3779B97F4A7C15ull * sub), a.key[1] + attempt, out);
}

__device__ __forceinline__ double u53(u64 w) { return (double)(w >> 11) * (1.0 / 9007199254740992.0); }
__device__ __forceinline__ double uopen(u64 w) { return ((double)(w >> 12) + 0.5) * (1.0 / 4503599627370496.0); }
__device__ __forceinline__ double box_muller(u64 w0, u64 w1) {
  return sqrt(-2.0 * log(uopen(w0))) * cos(6.283185307179586 * uopen(w1));
}

__device__ __forceinline__ double load_f(const RandArgs& a, int j, long long i) {
  const long long e = a.st[j] * i;
  switch (a.dt[j]) {
    case PTHIP_F64: return ((const double*)a.p[j])[e];
    case PTHIP_F32: return (double)((const float*)a.p[j])[e];
    case PTHIP_I64: return (double)((const long long*)a.p[j])[e];
    case PTHIP_I32: return (double)((const int*)a.p[j])[e];
    case PTHIP_I16: return (double)((const short*)a.p[j])[e];
    case PTHIP_I8: return (double)((const signed char*)a.p[j])[e];
    case PTHIP_U8: return (double)((const unsigned char*)a.p[j])[e];
    case PTHIP_U16: return (double)((const unsigned short*)a.p[j])[e];
    case PTHIP_U32: return (double)((const unsigned int*)a.p[j])[e];
    case PTHIP_U64: return (double)((const unsigned long long*)a.p[j])[e];
    case PTHIP_F16: return (double)((const _Float16*)a.p[j])[e];
    default: return (double)((const bool*)a.p[j])[e];
  }
}
__device__ __forceinline__ long long load_i(const RandArgs& a, int j, long long i) {
  const long long e = a.st[j] * i;
  switch (a.dt[j]) {
    case PTHIP_I64: return ((const long long*)a.p[j])[e];
    case PTHIP_I32: return ((const int*)a.p[j])[e];
    case PTHIP_I16: return ((const short*)a.p[j])[e];
    case PTHIP_I8: return ((const signed char*)a.p[j])[e];
    case PTHIP_U8: return ((const unsigned char*)a.p[j])[e];
    case PTHIP_U16: return ((const unsigned short*)a.p[j])[e];
    case PTHIP_U32: return ((const unsigned int*)a.p[j])[e];
    case PTHIP_U64: return (long long)((const unsigned long long*)a.p[j])[e];
    case PTHIP_F64: return (long long)((const double*)a.p[j])[e];
    case PTHIP_F32: return (long long)((const float*)a.p[j])[e];
    default: return ((const bool*)a.p[j])[e];
  }
}

template <class T> __device__ __forceinline__ void store(void* out, long long i, double v) { ((T*)out)[i] = (T)v; }

// Marsaglia & Tsang (2000) with the U^(1/a) boost below a = 1; unit scale
__device__ double gamma_mt(const RandArgs& a, u64 i, unsigned sub, double shape) {
  if (!(shape > 0.0)) return shape == 0.0 ? 0.0 : __builtin_nan("");
  u64 w[4];
  draw_block(a, i, sub, 0, w);
  double boost = 1.0;
  if (shape < 1.0) { boost = pow(uopen(w[3]), 1.0 / shape); shape += 1.0; }
  const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  for (unsigned attempt = 0; attempt < 64; attempt++) {
    if (attempt) draw_block(a, i, sub, attempt, w);
    const double z = box_muller(w[0], w[1]);
    double v = 1.0 + c * z;
    if (v <= 0.0) continue;
    v = v * v * v;
    const double u = uopen(w[2]);
    if (log(u) < 0.5 * z * z + d - d * v + d * log(v)) return d * v * boost;
  }
  return d * boost;  // (64 consecutive rejections: probability below 1e-80)
}

// Knuth's product method below lambda = 10, Hoermann's PTRS (1993) above
__device__ double poisson_draw(const RandArgs& a, u64 i, unsigned sub, double lam) {
  if (!(lam >= 0.0)) return __builtin_nan("");
  if (lam == 0.0) return 0.0;
  u64 w[4];
  if (lam < 10.0) {
    const double L = exp(-lam);
    double p = 1.0;
    long long k = 0;
    for (unsigned attempt = 0; attempt < 64; attempt++) {
      draw_block(a, i, sub, attempt, w);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        p *= uopen(w[j]);
        if (p <= L) return (double)k;
        k++;
      }
    }
    return (double)k;
  }
  const double slam = sqrt(lam), loglam = log(lam), b = 0.931 + 2.53 * slam, al = -0.059 + 0.02483 * b;
  const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  for (unsigned attempt = 0; attempt < 256; attempt++) {
    draw_block(a, i, sub, attempt, w);
    const double U = uopen(w[0]) - 0.5, V = uopen(w[1]);
    const double us = 0.5 - fabs(U);
    const double k = floor((2.0 * al / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return k;
    if (k < 0.0 || (us < 0.013 && V > us)) continue;
    if (log(V) + log(invalpha) - log(al / (us * us) + b) <= -lam + k * loglam - lgamma(k + 1.0)) return k;
  }
  return floor(lam);
}

// binomial(n, p): sequential inversion while n*min(p,1-p) < 10, Hoermann's BTRS (1993) above
__device__ double binomial_draw(const RandArgs& a, u64 i, unsigned sub, double n, double p) {
  if (!(p >= 0.0 && p <= 1.0) || !(n >= 0.0)) return __builtin_nan("");
  n = floor(n);
  const bool flip = p > 0.5;
  const double q = flip ? 1.0 - p : p;
  if (q == 0.0 || n == 0.0) return flip ? n : 0.0;
  u64 w[4];
  double x = -1.0;
  if (n * q < 10.0) {
    const double qn = exp(n * log1p(-q)), odds = q / (1.0 - q);
    const double bound = fmin(n, n * q + 10.0 * sqrt(n * q * (1.0 - q) + 1.0));
    for (unsigned attempt = 0; attempt < 64 && x < 0.0; attempt++) {
      draw_block(a, i, sub, attempt, w);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (x >= 0.0) break;
        double u = uopen(w[j]), px = qn, k = 0.0;
        bool ok = true;
        while (u > px) {
          k += 1.0;
          if (k > bound) { ok = false; break; }
          u -= px;
          px *= (n - k + 1.0) * odds / k;
        }
        if (ok) x = k;
      }
    }
    if (x < 0.0) x = floor(n * q);
  } else {
    const double spq = sqrt(n * q * (1.0 - q)), b = 1.15 + 2.53 * spq, al = -0.0873 + 0.0248 * b + 0.01 * q;
    const double c = n * q + 0.5, vr = 0.92 - 4.2 / b, alpha = (2.83 + 5.1 / b) * spq, lpq = log(q / (1.0 - q));
    const double m = floor((n + 1.0) * q), h = lgamma(m + 1.0) + lgamma(n - m + 1.0);
    for (unsigned attempt = 0; attempt < 256 && x < 0.0; attempt++) {
      draw_block(a, i, sub, attempt, w);
      const double u = uopen(w[0]) - 0.5;
      double v = uopen(w[1]);
      const double us = 0.5 - fabs(u);
      const double k = floor((2.0 * al / us + b) * u + c);
      if (k < 0.0 || k > n) continue;
      if (us >= 0.07 && v <= vr) { x = k; break; }
      v = log(v * alpha / (al / (us * us) + b));
      if (v <= h - lgamma(k + 1.0) - lgamma(n - k + 1.0) + (k - m) * lpq) x = k;
    }
    if (x < 0.0) x = m;
  }
  return flip ? n - x : x;
}

// von Mises(mu, kappa): Best & Fisher (1979) wrapped-Cauchy rejection; uniform on the circle below
// kappa = 1e-8, wrapped normal above 1e6 (where the envelope's r loses its digits); result in [-pi, pi]
__device__ double vonmises_draw(const RandArgs& a, u64 i, double mu, double kappa) {
  const double PI = 3.141592653589793;
  if (!(kappa >= 0.0)) return __builtin_nan("");
  u64 w[4];
  draw_block(a, i, 0, 0, w);
  if (kappa < 1e-8) return PI * (2.0 * uopen(w[0]) - 1.0);
  double res;
  if (kappa > 1e6) {
    res = mu + sqrt(1.0 / kappa) * box_muller(w[0], w[1]);
  } else {
    const double s = 0.5 / kappa, r = s + sqrt(1.0 + s * s);
    double W = 1.0;
    u64 sign = w[2];
    for (unsigned attempt = 0; attempt < 256; attempt++) {
      if (attempt) draw_block(a, i, 0, attempt, w);
      const double Z = cos(PI * uopen(w[0]));
      W = (1.0 + r * Z) / (r + Z);
      const double Y = kappa * (r - W), V = uopen(w[1]);
      sign = w[2];
      if (Y * (2.0 - Y) - V >= 0.0 || log(Y / V) + 1.0 - Y >= 0.0) break;
    }
    W = fmin(1.0, fmax(-1.0, W));
    res = acos(W);
    if (sign >> 63) res = -res;
    res += mu;
  }
  const bool neg = res < 0.0;
  double m = fmod(fabs(res) + PI, 2.0 * PI) - PI;
  return neg ? -m : m;
}

// hypergeometric(ngood, nbad, nsample): inversion of one uniform by chop-down from the mode, the
// pmf walked outwards with its two-term recurrences (expected steps: a few standard deviations)
__device__ double hypergeometric_draw(const RandArgs& a, u64 i, double good, double bad, double sample) {
  good = floor(good); bad = floor(bad); sample = floor(sample);
  if (!(good >= 0.0 && bad >= 0.0 && sample >= 0.0) || sample > good + bad) return __builtin_nan("");
  const double lo = fmax(0.0, sample - bad), hi = fmin(sample, good);
  if (lo == hi) return lo;
  u64 w[4];
  draw_block(a, i, 0, 0, w);
  double u = uopen(w[0]);
  double m = floor((sample + 1.0) * (good + 1.0) / (good + bad + 2.0));
  m = fmin(hi, fmax(lo, m));
  const double lpm = lgamma(good + 1.0) - lgamma(m + 1.0) - lgamma(good - m + 1.0) + lgamma(bad + 1.0) -
                     lgamma(sample - m + 1.0) - lgamma(bad - sample + m + 1.0) - lgamma(good + bad + 1.0) +
                     lgamma(sample + 1.0) + lgamma(good + bad - sample + 1.0);
  const double pm = exp(lpm);
  u -= pm;
  if (u <= 0.0) return m;
  double kd = m, ku = m, pd = pm, pu = pm;
  for (long long it = 0; it < (1ll << 40); it++) {
    const bool can_d = kd > lo, can_u = ku < hi;
    if (!can_d && !can_u) break;
    if (can_d) {
      pd *= kd * (bad - sample + kd) / ((good - kd + 1.0) * (sample - kd + 1.0));
      kd -= 1.0;
      u -= pd;
      if (u <= 0.0) return kd;
    }
    if (can_u) {
      pu *= (good - ku) * (sample - ku) / ((ku + 1.0) * (bad - sample + ku + 1.0));
      ku += 1.0;
      u -= pu;
      if (u <= 0.0) return ku;
    }
  }
  return m;  // (u fell in the rounding residue of the total mass)
}

template <class T>
__global__ __launch_bounds__(BLOCK) void random_kernel(int dist, long long n, RandArgs a, T* __restrict__ out) {
  const long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const long long nth = (long long)gridDim.x * BLOCK;
  u64 w[4];
  if (dist == D_UNIFORM) {
    for (long long blk = tid; blk * 4 < n; blk += nth) {
      draw_block(a, (u64)blk, 0, 0, w);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const long long i = blk * 4 + j;
        if (i < n) {
          // low + (high - low) * u as three rounded operations (NumPy's random_uniform): no fma
#pragma clang fp contract(off)
          const double lo = load_f(a, 0, i), hi = load_f(a, 1, i);
          const double span = hi - lo;
          const double scaled = span * u53(w[j]);
          out[i] = (T)(lo + scaled);
        }
      }
    }
    return;
  }
  for (long long i = tid; i < n; i += nth) {
    double r;
    switch (dist) {
      case D_NORMAL: case D_HALFNORMAL: case D_LOGNORMAL: {
        draw_block(a, (u64)i, 0, 0, w);
        double z = box_muller(w[0], w[1]);
        if (dist == D_HALFNORMAL) z = fabs(z);
        r = load_f(a, 0, i) + load_f(a, 1, i) * z;
        if (dist == D_LOGNORMAL) r = exp(r);
      } break;
      case D_EXPONENTIAL: draw_block(a, (u64)i, 0, 0, w); r = -load_f(a, 0, i) * log(uopen(w[0])); break;
      case D_LAPLACE: {
        draw_block(a, (u64)i, 0, 0, w);
        const double e = -log(uopen(w[0]));
        r = load_f(a, 0, i) + load_f(a, 1, i) * ((w[1] >> 63) ? e : -e);
      } break;
      case D_LOGISTIC: {
        draw_block(a, (u64)i, 0, 0, w);
        const double u = uopen(w[0]);
        r = load_f(a, 0, i) + load_f(a, 1, i) * log(u / (1.0 - u));
      } break;
      case D_CAUCHY: draw_block(a, (u64)i, 0, 0, w); r = load_f(a, 0, i) + load_f(a, 1, i) * tan(3.141592653589793 * (uopen(w[0]) - 0.5)); break;
      case D_HALFCAUCHY: draw_block(a, (u64)i, 0, 0, w); r = load_f(a, 0, i) + load_f(a, 1, i) * tan(1.5707963267948966 * uopen(w[0])); break;
      case D_GUMBEL: draw_block(a, (u64)i, 0, 0, w); r = load_f(a, 0, i) - load_f(a, 1, i) * log(-log(uopen(w[0]))); break;
      case D_WEIBULL: draw_block(a, (u64)i, 0, 0, w); r = pow(-log(uopen(w[0])), 1.0 / load_f(a, 0, i)); break;
      case D_PARETO: draw_block(a, (u64)i, 0, 0, w); r = load_f(a, 1, i) * exp(-log(uopen(w[0])) / load_f(a, 0, i)); break;
      case D_TRIANGULAR: {
        draw_block(a, (u64)i, 0, 0, w);
        const double l = load_f(a, 0, i), m = load_f(a, 1, i), h = load_f(a, 2, i), u = uopen(w[0]);
        const double fc = (m - l) / (h - l);
        r = u < fc ? l + sqrt(u * (h - l) * (m - l)) : h - sqrt((1.0 - u) * (h - l) * (h - m));
      } break;
      case D_GAMMA: r = gamma_mt(a, (u64)i, 0, load_f(a, 0, i)) * load_f(a, 1, i); break;
      case D_BETA: {
        const double x = gamma_mt(a, (u64)i, 0, load_f(a, 0, i)), y = gamma_mt(a, (u64)i, 1, load_f(a, 1, i));
        r = x / (x + y);
      } break;
      case D_INVGAMMA: r = load_f(a, 1, i) / gamma_mt(a, (u64)i, 0, load_f(a, 0, i)); break;
      case D_STUDENT_T: {
        const double df = load_f(a, 0, i);
        const double g = gamma_mt(a, (u64)i, 0, 0.5 * df);
        draw_block(a, (u64)i, 1, 0, w);
        r = load_f(a, 1, i) + load_f(a, 2, i) * (sqrt(0.5 * df) * box_muller(w[0], w[1]) / sqrt(g));
      } break;
      case D_BERNOULLI: draw_block(a, (u64)i, 0, 0, w); r = uopen(w[0]) < load_f(a, 0, i) ? 1.0 : 0.0; break;
      case D_GEOMETRIC: {
        draw_block(a, (u64)i, 0, 0, w);
        const double p = load_f(a, 0, i);
        r = p >= 1.0 ? 1.0 : ceil(log(uopen(w[0])) / log1p(-p));
        if (r < 1.0) r = 1.0;
      } break;
      case D_POISSON: r = poisson_draw(a, (u64)i, 0, load_f(a, 0, i)); break;
      case D_BINOMIAL: r = binomial_draw(a, (u64)i, 0, load_f(a, 0, i), load_f(a, 1, i)); break;
      case D_NEGBINOMIAL: {
        // gamma-Poisson mixture: lambda ~ Gamma(n, (1-p)/p), X ~ Poisson(lambda)
        const double nn = load_f(a, 0, i), p = load_f(a, 1, i);
        r = poisson_draw(a, (u64)i, 1, gamma_mt(a, (u64)i, 0, nn) * ((1.0 - p) / p));
      } break;
      case D_WALD: {
        // Michael, Schucany & Haas (1976): one chi-square(1) root, chosen by one uniform
        draw_block(a, (u64)i, 0, 0, w);
        const double mu = load_f(a, 0, i), lam = load_f(a, 1, i);
        const double z = box_muller(w[0], w[1]);
        const double y = mu * z * z, d = 0.5 * mu / lam;
        const double x = mu + d * (y - sqrt(4.0 * lam * y + y * y));
        r = uopen(w[2]) <= mu / (mu + x) ? x : mu * mu / x;
      } break;
      case D_TRUNCEXPON: {
        // inverse cdf on [0, b]: -log(1 - u (1 - e^-b)), then loc + scale x
        draw_block(a, (u64)i, 0, 0, w);
        const double b = load_f(a, 0, i);
        r = load_f(a, 1, i) + load_f(a, 2, i) * -log1p(uopen(w[0]) * expm1(-b));
      } break;
      case D_GENGAMMA: {
        // GenGammaRV.rng_fn_scipy (random/basic.py:1739): lambd * Gamma(alpha / p)^(1/p)
        const double al = load_f(a, 0, i), pw = load_f(a, 1, i);
        r = load_f(a, 2, i) * pow(gamma_mt(a, (u64)i, 0, al / pw), 1.0 / pw);
      } break;
      case D_BETABINOMIAL: {
        const double nn = load_f(a, 0, i);
        const double x = gamma_mt(a, (u64)i, 0, load_f(a, 1, i)), y = gamma_mt(a, (u64)i, 1, load_f(a, 2, i));
        r = binomial_draw(a, (u64)i, 2, nn, x / (x + y));
      } break;
      case D_VONMISES: r = vonmises_draw(a, (u64)i, load_f(a, 0, i), load_f(a, 1, i)); break;
      case D_HYPERGEOMETRIC: r = hypergeometric_draw(a, (u64)i, load_f(a, 0, i), load_f(a, 1, i), load_f(a, 2, i)); break;
      default: r = __builtin_nan(""); break;
    }
    out[i] = (T)r;
  }
}

// integers(low, high): low + floor(w * (high - low) / 2^64) in exact 64-bit arithmetic
__global__ __launch_bounds__(BLOCK) void integers_kernel(long long n, RandArgs a, long long* __restrict__ out) {
  const long long nth = (long long)gridDim.x * BLOCK;
  u64 w[4];
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += nth) {
    draw_block(a, (u64)i, 0, 0, w);
    const long long lo = load_i(a, 0, i), hi = load_i(a, 1, i);
    const u64 range = (u64)hi - (u64)lo;
    out[i] = lo + (long long)__umul64hi(w[0], range);
  }
}

// categorical: one uniform per row, first index whose running sum of p exceeds it
template <class P>
__global__ __launch_bounds__(BLOCK) void categorical_kernel(long long rows, long long k, RandArgs a,
                                                           const P* __restrict__ p, long long row_stride,
                                                           long long* __restrict__ out) {
  const long long nth = (long long)gridDim.x * BLOCK;
  u64 w[4];
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < rows; i += nth) {
    draw_block(a, (u64)i, 0, 0, w);
    const double u = uopen(w[0]);
    const P* row = p + i * row_stride;
    double acc = 0.0;
    long long pick = k - 1;
    for (long long j = 0; j < k; j++) {
      acc += (double)row[j];
      if (u < acc) { pick = j; break; }
    }
    out[i] = pick;
  }
}

// multinomial(n, p[k]): the conditional binomials X_j | X_<j ~ Binomial(n - sum X_<j, p_j / (1 - sum p_<j)),
// one row per thread, category j on substream j; the last category takes the remainder
template <class P>
__global__ __launch_bounds__(BLOCK) void multinomial_kernel(long long rows, long long k, RandArgs a,
                                                           const P* __restrict__ p, long long row_stride,
                                                           long long* __restrict__ out) {
  const long long nth = (long long)gridDim.x * BLOCK;
  for (long long i = (long long)blockIdx.x * BLOCK + threadIdx.x; i < rows; i += nth) {
    const P* row = p + i * row_stride;
    double remaining = (double)load_i(a, 0, i), mass = 1.0;
    long long* o = out + i * k;
    for (long long j = 0; j + 1 < k; j++) {
      const double pj = (double)row[j];
      double x = 0.0;
      if (remaining > 0.0) {
        const double q = mass > 0.0 ? fmin(1.0, fmax(0.0, pj / mass)) : 1.0;
        x = binomial_draw(a, (u64)i, (unsigned)j, remaining, q);
      }
      o[j] = (long long)x;
      remaining -= x;
      mass -= pj;
    }
    o[k - 1] = (long long)remaining;
  }
}

int grid_for(long long n) {
  long long g = (n + BLOCK - 1) / BLOCK;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int pthip_random(int dist, int out_dtype, int64_t n, const uint64_t* key, const uint64_t* counter,
                            int nparams, const void* const* params, const int* param_dtypes,
                            const int64_t* param_strides, void* out) {
  PTHIP_REQUIRE_INIT();
  if (dist < 0 || dist >= D_COUNT) return pthip::set_error("pthip_random: unknown distribution code %d", dist);
  if (nparams < 0 || nparams > 3) return pthip::set_error("pthip_random: %d parameters (at most 3)", nparams);
  if (n <= 0) return 0;
  RandArgs a;
  for (int j = 0; j < 3; j++) {
    a.p[j] = j < nparams ? params[j] : nullptr;
    a.dt[j] = j < nparams ? param_dtypes[j] : PTHIP_F64;
    a.st[j] = j < nparams ? param_strides[j] : 0;
  }
  a.key[0] = key[0]; a.key[1] = key[1];
  for (int j = 0; j < 4; j++) a.ctr[j] = counter[j];
  hipStream_t st = pthip::ctx().stream;
  if (dist == D_INTEGERS) {
    if (out_dtype != PTHIP_I64) return pthip::set_error("pthip_random: integers draws are int64");
    PTHIP_KLAUNCH(integers_kernel, dim3(grid_for(n)), dim3(BLOCK), 0, st, (long long)n, a, (long long*)out);
    return pthip::post_launch("random_integers");
  }
  const long long work = dist == D_UNIFORM ? (n + 3) / 4 : n;
  if (out_dtype == PTHIP_F64) PTHIP_KLAUNCH(random_kernel<double>, dim3(grid_for(work)), dim3(BLOCK), 0, st, dist, (long long)n, a, (double*)out);
  else if (out_dtype == PTHIP_F32) PTHIP_KLAUNCH(random_kernel<float>, dim3(grid_for(work)), dim3(BLOCK), 0, st, dist, (long long)n, a, (float*)out);
  else if (out_dtype == PTHIP_I64) PTHIP_KLAUNCH(random_kernel<long long>, dim3(grid_for(work)), dim3(BLOCK), 0, st, dist, (long long)n, a, (long long*)out);
  else return pthip::set_error("pthip_random: output dtype %d not supported (float64/float32/int64)", out_dtype);
  return pthip::post_launch("random");
}

extern "C" int pthip_random_categorical(int p_dtype, int64_t rows, int64_t k, const uint64_t* key,
                                        const uint64_t* counter, const void* p, int64_t row_stride, void* out) {
  PTHIP_REQUIRE_INIT();
  if (rows <= 0) return 0;
  if (k <= 0) return pthip::set_error("pthip_random_categorical: empty probability vector");
  RandArgs a = {};
  a.key[0] = key[0]; a.key[1] = key[1];
  for (int j = 0; j < 4; j++) a.ctr[j] = counter[j];
  hipStream_t st = pthip::ctx().stream;
  if (p_dtype == PTHIP_F64) PTHIP_KLAUNCH(categorical_kernel<double>, dim3(grid_for(rows)), dim3(BLOCK), 0, st, (long long)rows, (long long)k, a, (const double*)p, (long long)row_stride, (long long*)out);
  else if (p_dtype == PTHIP_F32) PTHIP_KLAUNCH(categorical_kernel<float>, dim3(grid_for(rows)), dim3(BLOCK), 0, st, (long long)rows, (long long)k, a, (const float*)p, (long long)row_stride, (long long*)out);
  else return pthip::set_error("pthip_random_categorical: probabilities must be float32/float64");
  return pthip::post_launch("random_categorical");
}

extern "C" int pthip_random_multinomial(int p_dtype, int64_t rows, int64_t k, const uint64_t* key,
                                        const uint64_t* counter, const void* n, int n_dtype, int64_t n_stride,
                                        const void* p, int64_t row_stride, void* out) {
  PTHIP_REQUIRE_INIT();
  if (rows <= 0) return 0;
  if (k <= 0) return pthip::set_error("pthip_random_multinomial: empty probability vector");
  RandArgs a = {};
  a.p[0] = n; a.dt[0] = n_dtype; a.st[0] = n_stride;
  a.key[0] = key[0]; a.key[1] = key[1];
  for (int j = 0; j < 4; j++) a.ctr[j] = counter[j];
  hipStream_t st = pthip::ctx().stream;
  if (p_dtype == PTHIP_F64) PTHIP_KLAUNCH(multinomial_kernel<double>, dim3(grid_for(rows)), dim3(BLOCK), 0, st, (long long)rows, (long long)k, a, (const double*)p, (long long)row_stride, (long long*)out);
  else if (p_dtype == PTHIP_F32) PTHIP_KLAUNCH(multinomial_kernel<float>, dim3(grid_for(rows)), dim3(BLOCK), 0, st, (long long)rows, (long long)k, a, (const float*)p, (long long)row_stride, (long long*)out);
  else return pthip::set_error("pthip_random_multinomial: probabilities must be float32/float64");
  return pthip::post_launch("random_multinomial");
}
