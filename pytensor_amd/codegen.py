"""Fused ``Elemwise``/``Composite`` → HIP kernel source for gfx950.

The reference emits the fused body with ``Composite.c_code_template``
(pytensor/scalar/basic.py:4111-4170) by concatenating each scalar op's ``c_code``
statement, and wraps it in the loop nest of ``Elemwise._c_all``
(pytensor/tensor/elemwise.py:848-1167; contiguous fast path 1083-1166).  Here the
same SSA walk emits a device expression per scalar op (same formulas as the
reference ``c_code`` strings, cited below) and wraps it in one of three loop shapes
designed for CDNA4 rather than for a CPU:

``flat``    every operand is either fully contiguous (same shape) or a scalar
            broadcast → grid-stride loop, 16-byte vector loads/stores per lane
            (coalesced 1 KiB per wave instruction), ``UNROLL`` independent packs
            in flight per thread;
``nd``      general broadcasting / arbitrary strides → collapsed N-d index with
            per-operand element strides (0 for broadcast dims);
``reduce``  ``flat`` or ``nd`` loads fused with a full reduction of selected
            outputs: per-thread accumulators (acc dtype) → wave64 butterfly →
            LDS → one partial per workgroup; reduced outputs never touch HBM.

Kernel parameters are all 8 bytes wide (pointers, ``long long``) so the host
packs the argument buffer with plain ``struct.pack("<q...")``.
"""

from __future__ import annotations

import hashlib
import math
import os

import numpy as np

CTYPE = {
    "float64": "double",
    "float32": "float",
    "int64": "long long",
    "int32": "int",
    "int16": "short",
    "int8": "signed char",
    "uint8": "unsigned char",
    "uint16": "unsigned short",
    "uint32": "unsigned int",
    "uint64": "unsigned long long",
    # storage type with per-op rounding: every SSA temporary of dtype float16 is a `_Float16`, so
    # +,-,*,/ are IEEE half operations and libm-style ops are evaluated in float and rounded —
    # what NumPy does for float16 scalars (the reference has no C code for float16:
    # Elemwise runs `perform`, tensor/elemwise.py:755-823)
    "float16": "_Float16",
    "bool": "bool",
}

_HERE = os.path.dirname(os.path.abspath(__file__))
_reduce_header_cache = None


def reduce_header() -> str:
    global _reduce_header_cache
    if _reduce_header_cache is None:
        src = open(os.path.join(_HERE, "csrc", "reduce_device.h")).read()
        _reduce_header_cache = src.replace("#pragma once", "")
    return _reduce_header_cache


PRELUDE = r"""
// ---- scalar helpers (formulas follow the reference c_code; citations in codegen.py) ----
#define PT_DEV static __device__ __forceinline__
template <class T> PT_DEV T pt_sqr(T x) { return x * x; }
template <class T> PT_DEV T pt_max(T x, T y) { return (y > x) ? y : ((x >= y) ? x : (T)__builtin_nan("")); }
template <class T> PT_DEV T pt_min(T x, T y) { return (y < x) ? y : ((x <= y) ? x : (T)__builtin_nan("")); }
PT_DEV bool pt_max(bool x, bool y) { return x || y; }
PT_DEV bool pt_min(bool x, bool y) { return x && y; }
PT_DEV double pt_sign(double x) { return (x > 0) ? 1. : ((x < 0) ? -1. : (isnan(x) ? __builtin_nan("") : 0.)); }
PT_DEV float pt_sign(float x) { return (x > 0) ? 1.f : ((x < 0) ? -1.f : (isnan(x) ? __builtin_nanf("") : 0.f)); }
template <class T> PT_DEV T pt_sign(T x) { return (x >= 0) ? ((x == 0) ? 0 : 1) : -1; }
// fp64 exp in 24 VALU instructions (the device library's is ~34; in BASELINE config #2 that one call was 41 % of
// the kernel's VALU work next to a 25 us HBM floor): n = rint(x log2 e); r = x - n ln2 in two FMAs (ln2_hi has 21
// trailing zero bits: n ln2_hi is exact); exp(r) = 1 + r + r^2 Q(r), Q a degree-9 Chebyshev fit on
// |r| <= ln2/2 computed with mpmath at 60 digits; 2^n by v_ldexp_f64.  <= 1 ulp from the correctly rounded
// value on 6e5 points in [-700, 700] (mean 0.10 ulp); Exp.c_code of the reference is libm's exp
// (pytensor/scalar/basic.py:3085-3118), itself < 1 ulp.  Overflow -> inf, underflow -> 0, NaN -> NaN.
PT_DEV double pt_exp(double x) {
  const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
  double r = __builtin_fma(n, -0x1.62e42fee00000p-1, x);
  r = __builtin_fma(n, -0x1.a39ef35793c76p-33, r);
  double q = 0x1.af38a9b0ec855p-26;
  q = __builtin_fma(q, r, 0x1.289185613a3d6p-22);
  q = __builtin_fma(q, r, 0x1.71de0dae63bb3p-19);
  q = __builtin_fma(q, r, 0x1.a019b90d2ae7ap-16);
  q = __builtin_fma(q, r, 0x1.a01a01a7c41d5p-13);
  q = __builtin_fma(q, r, 0x1.6c16c1788bd90p-10);
  q = __builtin_fma(q, r, 0x1.11111111109b3p-7);
  q = __builtin_fma(q, r, 0x1.5555555553d63p-5);
  q = __builtin_fma(q, r, 0x1.5555555555556p-3);
  q = __builtin_fma(q, r, 0x1.0000000000001p-1);
  const double p = __builtin_fma(q * r, r, r) + 1.0;
  double y = __builtin_ldexp(p, (int)n);
  y = x > 0x1.62e42fefa39efp+9 ? __builtin_huge_val() : y;
  y = x < -0x1.74910d52d3051p+9 ? 0.0 : y;
  return y;
}
// The same exp with its constants held in registers by the caller (pt_expk_load once per kernel): in straight-line code with
// dozens of exp instances (the log-sum-exp reduction: n + 1 per tile visit, fully unrolled) the compiler materialises every
// polynomial coefficient again for every instance — v_fmac overwrites its addend, so each Horner step is two v_mov_b32 of a
// literal plus the fmac: 44 instructions per exp instead of 24 (profiles/r5w_lse_pmc.md: 94 VALU instructions per element).
// With the coefficients live in VGPRs across instances each step is one v_fma_f64.
struct pt_expk { double l2e, nh, nl, c[10], hi, lo; };
PT_DEV pt_expk pt_expk_load() {
  pt_expk k = {0x1.71547652b82fep+0, -0x1.62e42fee00000p-1, -0x1.a39ef35793c76p-33,
               {0x1.af38a9b0ec855p-26, 0x1.289185613a3d6p-22, 0x1.71de0dae63bb3p-19, 0x1.a019b90d2ae7ap-16, 0x1.a01a01a7c41d5p-13,
                0x1.6c16c1788bd90p-10, 0x1.11111111109b3p-7, 0x1.5555555553d63p-5, 0x1.5555555555556p-3, 0x1.0000000000001p-1},
               0x1.62e42fefa39efp+9, -0x1.74910d52d3051p+9};
  asm volatile("" : "+v"(k.l2e), "+v"(k.nh), "+v"(k.nl), "+v"(k.hi), "+v"(k.lo));
#pragma unroll
  for (int i = 0; i < 10; i++) asm volatile("" : "+v"(k.c[i]));
  return k;
}
PT_DEV double pt_exp_k(double x, const pt_expk& k) {
  const double n = __builtin_rint(x * k.l2e);
  double r = __builtin_fma(n, k.nh, x);
  r = __builtin_fma(n, k.nl, r);
  double q = k.c[0];
#pragma unroll
  for (int i = 1; i < 10; i++) q = __builtin_fma(q, r, k.c[i]);
  const double p = __builtin_fma(q * r, r, r) + 1.0;
  double y = __builtin_ldexp(p, (int)n);
  y = x > k.hi ? __builtin_huge_val() : y;
  y = x < k.lo ? 0.0 : y;
  return y;
}
// fp64 tanh in ~45 VALU instructions (the device library's is ~85: BASELINE config #2's transcendental variant is 10 tanh + 10
// exp per element and VALU-issue bound).  |x| < 0.35: the odd Taylor series to x^27 (coefficients 2^2n (2^2n - 1) B_2n / (2n)!
// from mpmath at 60 digits; the first neglected term is < 2e-18 relative): 0.55 ulp on 2e4 points.  Otherwise
// (1 - u) / (1 + u), u = exp(-2|x|) <= 0.497 (no cancellation in 1 - u): 1.95 ulp max, 0.52 mean on 2e4 points; saturates to
// +-1 from |x| = 19.07 on because u drops below 2^-55.  Tanh.c_code of the reference is libm's tanh (scalar/basic.py:3702;
// glibc: 1.2 ulp).  NaN -> NaN.
PT_DEV double pt_tanh(double x) {
  const double ax = __builtin_fabs(x);
  if (ax < 0.35) {
    const double z = x * x;
    double p = -0x1.b0f72d3ee24e9p-18;
    p = __builtin_fma(p, z, 0x1.0b132d39a6050p-16);
    p = __builtin_fma(p, z, -0x1.497d8eea25259p-15);
    p = __builtin_fma(p, z, 0x1.967e18afcafadp-14);
    p = __builtin_fma(p, z, -0x1.f57d7734d1664p-13);
    p = __builtin_fma(p, z, 0x1.3558248036744p-11);
    p = __builtin_fma(p, z, -0x1.7da36452b75e3p-10);
    p = __builtin_fma(p, z, 0x1.d6d3d0e157de0p-9);
    p = __builtin_fma(p, z, -0x1.226e355e6c23dp-7);
    p = __builtin_fma(p, z, 0x1.664f4882c10fap-6);
    p = __builtin_fma(p, z, -0x1.ba1ba1ba1ba1cp-5);
    p = __builtin_fma(p, z, 0x1.1111111111111p-3);
    p = __builtin_fma(p, z, -0x1.5555555555555p-2);
    return __builtin_fma(x, z * p, x);
  }
  const double u = pt_exp(-2.0 * ax);
  return __builtin_copysign((1.0 - u) / (1.0 + u), x);
}
// Python-style floor division / modulo for integers (IntDiv / Mod c_code, scalar/basic.py)
template <class T> PT_DEV T pt_intdiv_i(T x, T y) {
  if (y == 0) return 0;
  T q = x / y;
  if ((x % y != 0) && ((x < 0) != (y < 0))) q -= 1;
  return q;
}
template <class T> PT_DEV T pt_mod_i(T x, T y) {
  if (y == 0) return 0;
  T r = x % y;
  if (r != 0 && ((r < 0) != (y < 0))) r += y;
  return r;
}
PT_DEV double pt_intdiv_f(double x, double y) { return floor(x / y); }
PT_DEV float pt_intdiv_f(float x, float y) { return floorf(x / y); }
PT_DEV double pt_mod_f(double x, double y) {
  if (y == 0) return __builtin_nan("");
  double r = fmod(x, y);
  if (r != 0 && ((r < 0) != (y < 0))) r += y;
  return r;
}
PT_DEV float pt_mod_f(float x, float y) {
  if (y == 0) return __builtin_nanf("");
  float r = fmodf(x, y);
  if (r != 0 && ((r < 0) != (y < 0))) r += y;
  return r;
}
// fp64 log1p in ~50 VALU instructions (the device library's is ~125; a logistic or Student-t log-density term is one log1p
// per element and VALU-issue bound: profiles/r7_wide200_pmc.md).  The classical reduction: 1 + x = 2^k (1 + f) with 1 + f in
// (sqrt(1/2), sqrt(2)], s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7 minimax R of the
// fdlibm family, k ln2 added as a hi/lo pair, and c = the rounding error of 1 + x (relative to 1 + x) added back; f = x itself
// while k = 0.  The two divisions have tame denominators (2 + f in [1.7, 2.42]; 1 + x only scales a term below one ulp), so
// they are v_rcp_f64 + Newton steps without the scale/fixup of a general fp64 division.  2.1 ulp max against long-double
// log1pl on 4e7 host-emulated points (all magnitudes, both signs, the k = 0 / 1 boundaries), 0.5 ulp typical;
// Log1p.c_code of the reference is libm's log1p (scalar/basic.py:3042; glibc < 1 ulp).  x < -1 -> NaN, -1 -> -inf,
// +inf -> +inf, NaN -> NaN, |x| < 2^-54 -> x (keeps -0.0).
PT_DEV double pt_log1p(double x) {
  const double u = 1.0 + x;
  double m = 2.0 * __builtin_amdgcn_frexp_mant(u);  // [1, 2)
  int k = __builtin_amdgcn_frexp_exp(u) - 1;
  const bool up = m > 0x1.6a09e667f3bcdp+0;
  m = up ? 0.5 * m : m;
  k = up ? k + 1 : k;
  double c = k > 0 ? 1.0 - (u - x) : x - (u - 1.0);
  c = k == 0 ? 0.0 : c * __builtin_amdgcn_rcp(u);
  const double f = k == 0 ? x : m - 1.0;
  const double d = 2.0 + f;
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  double sq = f * r;
  sq = __builtin_fma(__builtin_fma(-d, sq, f), r, sq);
  const double z = sq * sq, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3), 0x1.999999997fa04p-2);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 0x1.2f112df3e5244p-3, 0x1.7466496cb03dep-3), 0x1.2492494229359p-2), 0x1.5555555555593p-1);
  const double hf = 0.5 * f * f, dk = (double)k;
  double y = __builtin_fma(dk, 0x1.62e42fee00000p-1, f - (hf - __builtin_fma(sq, hf + (t1 + t2), __builtin_fma(dk, 0x1.a39ef35793c76p-33, c))));
  y = __builtin_fabs(x) < 0x1p-54 ? x : y;
  y = x > -1.0 ? y : (x == -1.0 ? -__builtin_huge_val() : __builtin_nan(""));
  y = x == __builtin_huge_val() ? x : y;
  return y;
}
// pow with the exact cases exact.  The device library's pow is within ~1.3 ulp but NOT exact where the result is
// representable: pow(3, 1) = 2.9999999999999996, pow(19, 3) = 6858.999999999999 — and an integer power, which Pow.c_code
// (scalar/basic.py:2250) computes as (T)pow((double)x, (double)y), then truncates to 6858.  libm's pow (the reference's) is
// correctly rounded in these cases.  So: an integer exponent |y| <= 64 of an integer-valued base is repeated squaring, taken
// when every product in it was exact (zero fma residual: always so while the result is below 2^53, and beyond for bases
// with factors of two) — and 1 / that for y < 0 (one correctly rounded division); y = +-1, +-2, +-3 of any base are products (<= 1.5 ulp); everything else is the library's value.
PT_DEV double pt_pow(double x, double y) {
  double r = pow(x, y);
  const double ay = __builtin_fabs(y);
  if (y == __builtin_rint(y) && ay >= 1.0 && ay <= 64.0) {
    const int n = (int)ay;
    if (x == __builtin_rint(x) && x != 0.0 && __builtin_fabs(x) < 0x1p53) {
      double b = __builtin_fabs(x), p = 1.0;
      bool exact = true;  // every product that went into p had a zero rounding error (fma residual)
#pragma unroll
      for (int k = 0; k < 7; k++) {
        if ((n >> k) & 1) {
          const double q = p * b;
          exact = exact && __builtin_fma(p, b, -q) == 0.0;
          p = q;
        }
        if ((n >> (k + 1)) != 0) {
          const double q = b * b;
          exact = exact && __builtin_fma(b, b, -q) == 0.0;
          b = q;
        }
      }
      if (exact && p < __builtin_huge_val()) {
        p = (x < 0.0 && (n & 1)) ? -p : p;
        r = y < 0.0 ? 1.0 / p : p;
      }
    } else if (n <= 3) {
      const double p = n == 1 ? x : n == 2 ? x * x : x * x * x;
      r = y < 0.0 ? 1.0 / p : p;
    }
  }
  return r;
}
// fp64 log by the same reduction (x = 2^k (1 + f), no rounding term): ~45 VALU instructions against the device library's ~90;
// 0.86 ulp max against long-double logl on 4e7 host-emulated points (normal and subnormal arguments, the neighbourhoods of 1,
// sqrt(2) and sqrt(1/2)).  Log.c_code of the reference is libm's log (scalar/basic.py:2896).  x < 0 -> NaN, +-0 -> -inf,
// +inf -> +inf, NaN -> NaN; subnormal arguments are normalised by v_frexp_mant_f64 / v_frexp_exp_i32_f64.
PT_DEV double pt_log(double x) {
  double m = 2.0 * __builtin_amdgcn_frexp_mant(x);
  int k = __builtin_amdgcn_frexp_exp(x) - 1;
  const bool up = m > 0x1.6a09e667f3bcdp+0;
  m = up ? 0.5 * m : m;
  k = up ? k + 1 : k;
  const double f = m - 1.0, d = 2.0 + f;
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  double sq = f * r;
  sq = __builtin_fma(__builtin_fma(-d, sq, f), r, sq);
  const double z = sq * sq, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3), 0x1.999999997fa04p-2);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 0x1.2f112df3e5244p-3, 0x1.7466496cb03dep-3), 0x1.2492494229359p-2), 0x1.5555555555593p-1);
  const double hf = 0.5 * f * f, dk = (double)k;
  double y = __builtin_fma(dk, 0x1.62e42fee00000p-1, f - (hf - __builtin_fma(sq, hf + (t1 + t2), dk * 0x1.a39ef35793c76p-33)));
  y = x > 0.0 ? y : (x == 0.0 ? -__builtin_huge_val() : __builtin_nan(""));
  y = x == __builtin_huge_val() ? x : y;
  return y;
}
PT_DEV double pt_sigmoid(double x) { return 1.0 / (1.0 + pt_exp(-x)); }
PT_DEV float pt_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
PT_DEV double pt_softplus(double x) {
  return x < -37.0 ? pt_exp(x) : x < 18.0 ? pt_log1p(pt_exp(x)) : x < 33.3 ? x + pt_exp(-x) : x;
}
// sigmoid(x) and softplus(x) of ONE argument (the logistic log-density and its gradient; a Bernoulli-logit likelihood):
// both from e = exp(-|x|) in (0, 1] — one exp instead of two or three, no overflow on either side.
//   sigmoid = 1 / (1 + e)        (x >= 0)      e / (1 + e)   (x < 0)
//   softplus = max(x, 0) + log1p(e)
// Within 2 ulp of pt_sigmoid / pt_softplus (Sigmoid.c_code: 1 / (1 + exp(-x)); Softplus.c_code: the four-branch form of
// scalar/math.py); NaN -> NaN, +inf -> (1, +inf), -inf -> (0, 0).  emit_body uses it when a body holds both of one operand.
PT_DEV void pt_sig_sp(double x, double& sg, double& sp) {
  const double e = pt_exp(-__builtin_fabs(x));
  const double inv = 1.0 / (1.0 + e);
  sg = x >= 0.0 ? inv : e * inv;
  sp = (x > 0.0 ? x : 0.0) + pt_log1p(e);
  if (x != x) { sg = x; sp = x; }
}
PT_DEV float pt_softplus(float x) {
  return x < -37.0f ? expf(x) : x < 18.0f ? log1pf(expf(x)) : x < 33.3f ? x + expf(-x) : x;
}
PT_DEV double pt_log1mexp(double x) { return x < -0.6931471805599453 ? pt_log1p(-exp(x)) : log(-expm1(x)); }
PT_DEV float pt_log1mexp(float x) { return x < -0.6931471805599453f ? log1pf(-expf(x)) : logf(-expm1f(x)); }
// RoundHalfToEven (scalar/basic.py:2737-2766 restates npy_rint with floor arithmetic): the
// hardware's v_rndne is that function exactly, and — unlike `x - floor(x)` — cannot have the
// producer of x contracted into it (x = a*b fused into an fma changes which side of a tie
// the value lands on: found by the golden vectors, 2.5 rounded to 3).
PT_DEV double pt_rint_even(double x) { return __builtin_rint(x); }
PT_DEV float pt_rint_even(float x) { return __builtin_rintf(x); }
// digamma (Psi): asymptotic series with recurrence shift, as in the reference's
// support code (scalar/math.py:403-470 `_psi`)
PT_DEV double pt_psi(double x) {
  const double S = 1.0e-5, C = 8.5, S3 = 8.333333333e-2, S4 = 8.333333333e-3, S5 = 3.968253968e-3,
               D1 = -0.5772156649;
  double y = x, psi = 0.0, R;
  if (y <= 0.0) {
    // poles at 0, -1, -2, ...: +inf (the reference's choice); elsewhere the reflection formula
    if (y == floor(y)) return __builtin_inf();
    const double pix = 3.14159265358979323846 * y;
    psi = -3.14159265358979323846 * (cos(pix) / sin(pix));
    y = 1.0 - y;
  }
  if (y <= S) return psi + D1 - 1.0 / y;
  while (y < C) { psi = psi - 1.0 / y; y = y + 1; }
  R = 1.0 / y;
  psi = psi + log(y) - .5 * R;
  R = R * R;
  psi = psi - R * (S3 - R * (S4 - R * S5));
  return psi;
}
PT_DEV float pt_psi(float x) { return (float)pt_psi((double)x); }
// trigamma: AS 121 with the 10-digit constants of TriGamma.c_support_code (scalar/math.py:518-567)
PT_DEV double pt_trigamma(double x) {
  const double b2 = 0.1666666667, b4 = -0.03333333333, b6 = 0.02380952381, b8 = -0.03333333333;
  if (x <= 0) return 0.0;  // (NaN compares false and runs through the series: NaN out)
  if (x <= 0.0001) return 1.0 / x / x;
  double value = 0.0, z = x;
  while (z < 5.0) { value += 1.0 / z / z; z += 1.0; }
  const double y = 1.0 / z / z;
  value += 0.5 * y + (1.0 + y * (b2 + y * (b4 + y * (b6 + y * b8)))) / z;
  return value;
}
PT_DEV float pt_trigamma(float x) { return (float)pt_trigamma((double)x); }
"""

# ---- incomplete gamma / beta: emitted only into kernels that use them ----


def _gamma_tables():
    """log(i!) and log(Gamma(i+1/2)) by the running products of the reference's support code
    (scalar/c_code/gamma.c:62-80; the last half-integer slot stays 0 there)."""
    logfs = [0.0] * 171
    loghs = [0.0] * 171
    x = 1.0
    for i in range(2, 171):
        x *= i
        logfs[i] = math.log(x)
    x = 1.77245385090551602729816748334
    loghs[0] = 0.5 * 1.14472988584940017414342735135
    for i in range(1, 170):
        x *= i - 0.5
        loghs[i] = math.log(x)
    return logfs, loghs


def _c_table(name, vals):
    return f"static __device__ const double {name}[{len(vals)}] = {{" + ", ".join(repr(float(v)) for v in vals) + "};\n"


_GAMMAINC_SRC = r"""
// regularised incomplete gamma P / Q as the reference's C backend computes them
// (scalar/c_code/gamma.c: logGamma 83-106, _series 143-155, _cfrac 172-189, GammaP 207-218,
//  GammaQ 222-233; called from GammaInc/GammaIncC.c_code, scalar/math.py:648-655, 695-702)
#define PT_G_EPS 2.2204460492503131e-16
#define PT_G_TINY (PT_G_EPS * PT_G_EPS * PT_G_EPS)
PT_DEV double pt_g_loggamma(double n) {
  if (n <= 0) return __builtin_nan("");
  if (n < 171 + 4 * PT_G_EPS) {
    if (fabs(n - floor(n)) < 4 * PT_G_EPS) { const int i = (int)floor(n) - 1; return pt_g_logfs[i < 0 ? 0 : i]; }
    if (fabs(2 * n - floor(2 * n)) < 4 * PT_G_EPS) return pt_g_loghs[(int)floor(n)];
  }
  double s = 0.99999999999980993227684700473478;
  s += 676.520368121885098567009190444019 / (n + 1);
  s += -1259.13921672240287047156078755283 / (n + 2);
  s += 771.3234287776530788486528258894 / (n + 3);
  s += -176.61502916214059906584551354 / (n + 4);
  s += 12.507343278686904814458936853 / (n + 5);
  s += -0.13857109526572011689554707 / (n + 6);
  s += 9.984369578019570859563e-6 / (n + 7);
  s += 1.50563273514931155834e-7 / (n + 8);
  return (n + 0.5) * log((n + 7.5) / 2.71828182845904523536028747135) + (0.918938533204672741780329736406 + log(s / n) - 7.0);
}
PT_DEV double pt_g_series(double n, double x) {
  double t = 1.0 / n, sum = t;
  for (int i = 0; i < 1024; i++) {
    n += 1.0;
    t *= x / n;
    sum += t;
    if (fabs(t) < fabs(sum) * PT_G_EPS) break;
  }
  return sum;
}
PT_DEV double pt_g_cfrac(double n, double x) {
  double b = x + 1 - n, c = 1 / PT_G_TINY, d = 1 / b, f = d;
  for (int i = 1; i < 1024; i++) {
    const double a = i * (n - i);
    b += 2;
    d = a * d + b;
    if (fabs(d) < PT_G_TINY) d = PT_G_TINY;
    c = b + a / c;
    if (fabs(c) < PT_G_TINY) c = PT_G_TINY;
    d = 1 / d;
    const double e = d * c;
    f *= e;
    if (fabs(e - 1) < PT_G_EPS) break;
  }
  return f;
}
PT_DEV double pt_gammainc(double n, double x) {
  if (isnan(n) || isnan(x)) return __builtin_nan("");
  if ((n <= 0) || (x < 0)) return __builtin_nan("");
  if (x <= 0) return 0;
  if (isinf(n)) return isinf(x) ? __builtin_nan("") : 0.0;
  if (isinf(x)) return 1;
  const double sc = exp(n * log(x) - x - pt_g_loggamma(n));
  if (x < n + 1) return pt_g_series(n, x) * sc;
  return 1 - pt_g_cfrac(n, x) * sc;
}
PT_DEV double pt_gammaincc(double n, double x) {
  if (isnan(n) || isnan(x)) return __builtin_nan("");
  if ((n <= 0) || (x < 0)) return __builtin_nan("");
  if (x <= 0) return 1;
  if (isinf(n)) return isinf(x) ? __builtin_nan("") : 1.0;
  if (isinf(x)) return 0;
  const double sc = exp(n * log(x) - x - pt_g_loggamma(n));
  if (x < n + 1) return 1 - pt_g_series(n, x) * sc;
  return pt_g_cfrac(n, x) * sc;
}
PT_DEV float pt_gammainc(float n, float x) { return (float)pt_gammainc((double)n, (double)x); }
PT_DEV float pt_gammaincc(float n, float x) { return (float)pt_gammaincc((double)n, (double)x); }
"""

_BETAINC_SRC = r"""
// regularised incomplete beta (Cephes incbet) as the reference's C backend computes it
// (scalar/c_code/incbet.c: BetaInc 33-90, incbcf 96-178, incbd 184-268, pseries 274-311;
//  called from BetaInc.c_code, scalar/math.py:1371-1381).  The reference flips (a, b, x) by
//  calling itself once; here the flipped evaluation is a second call of the same body.
#define PT_B_MINLOG -7.451332191019412076235E2
#define PT_B_MAXLOG 7.09782712893383996732E2
#define PT_B_MAXGAM 171.624376956302725
#define PT_B_EPS 1.11022302462515654042e-16
#define PT_B_BIG 4.503599627370496e15
#define PT_B_BIGINV 2.22044604925031308085e-16
// both continued fractions share one three-term recurrence: k1..k8 and their increments differ
PT_DEV double pt_b_cf(double xz, double k1, double k2, double k3, double k4, double k5, double k6,
                      double k7, double k8, double d2, double d6) {
  double pkm2 = 0.0, qkm2 = 1.0, pkm1 = 1.0, qkm1 = 1.0, ans = 1.0, r = 1.0, t;
  const double thresh = 3.0 * PT_B_EPS;
  int n = 0;
  do {
    double xk = -(xz * k1 * k2) / (k3 * k4);
    double pk = pkm1 + pkm2 * xk, qk = qkm1 + qkm2 * xk;
    pkm2 = pkm1; pkm1 = pk; qkm2 = qkm1; qkm1 = qk;
    xk = (xz * k5 * k6) / (k7 * k8);
    pk = pkm1 + pkm2 * xk; qk = qkm1 + qkm2 * xk;
    pkm2 = pkm1; pkm1 = pk; qkm2 = qkm1; qkm1 = qk;
    if (qk != 0.0) r = pk / qk;
    if (r != 0.0) { t = fabs((ans - r) / r); ans = r; } else t = 1.0;
    if (t < thresh) break;
    k1 += 1.0; k2 += d2; k3 += 2.0; k4 += 2.0; k5 += 1.0; k6 += d6; k7 += 2.0; k8 += 2.0;
    if ((fabs(qk) + fabs(pk)) > PT_B_BIG) { pkm2 *= PT_B_BIGINV; pkm1 *= PT_B_BIGINV; qkm2 *= PT_B_BIGINV; qkm1 *= PT_B_BIGINV; }
    if ((fabs(qk) < PT_B_BIGINV) || (fabs(pk) < PT_B_BIGINV)) { pkm2 *= PT_B_BIG; pkm1 *= PT_B_BIG; qkm2 *= PT_B_BIG; qkm1 *= PT_B_BIG; }
  } while (++n < 300);
  return ans;
}
PT_DEV double pt_b_pseries(double a, double b, double x) {
  const double ai = 1.0 / a;
  double u = (1.0 - b) * x, v = u / (a + 1.0), t = u, n = 2.0, s = 0.0;
  const double t1 = v, z = PT_B_EPS * ai;
  while (fabs(v) > z) {
    u = (n - b) * x / n;
    t *= u;
    v = t / (a + n);
    s += v;
    n += 1.0;
  }
  s += t1;
  s += ai;
  u = a * log(x);
  if ((a + b) < PT_B_MAXGAM && fabs(u) < PT_B_MAXLOG) {
    t = tgamma(a + b) / (tgamma(a) * tgamma(b));
    s = s * t * pow(x, a);
  } else {
    t = lgamma(a + b) - lgamma(a) - lgamma(b) + u + log(s);
    s = t < PT_B_MINLOG ? 0.0 : exp(t);
  }
  return s;
}
// everything of BetaInc() except the symmetry flip; *flip is set when the caller has to flip
PT_DEV double pt_b_body(double a, double b, double x, bool may_flip, bool* flip) {
  *flip = false;
  if (x == 0.0) return 0.0;
  if (x == 1.0) return 1.0;
  if ((b * x) <= 1.0 && x <= 0.95) return pt_b_pseries(a, b, x);
  const double xc = 1.0 - x;
  if (may_flip && x > (a / (a + b))) { *flip = true; return 0.0; }
  double y = x * (a + b - 2.0) - (a - 1.0), w, t;
  if (y < 0.0) w = pt_b_cf(x, a, a + b, a, a + 1.0, 1.0, b - 1.0, a + 1.0, a + 2.0, 1.0, -1.0);
  else w = pt_b_cf(x / (1.0 - x), a, b - 1.0, a, a + 1.0, 1.0, a + b, a + 1.0, a + 2.0, -1.0, 1.0) / xc;
  y = a * log(x);
  t = b * log(xc);
  if ((a + b) < PT_B_MAXGAM && fabs(y) < PT_B_MAXLOG && fabs(t) < PT_B_MAXLOG) {
    t = pow(xc, b);
    t *= pow(x, a);
    t /= a;
    t *= w;
    t *= tgamma(a + b) / (tgamma(a) * tgamma(b));
    return t;
  }
  y += t + lgamma(a + b) - lgamma(a) - lgamma(b);
  y += log(w / a);
  return y < PT_B_MINLOG ? 0.0 : exp(y);
}
PT_DEV double pt_betainc(double a, double b, double x) {
  if (isnan(a) || isnan(b) || isnan(x)) return __builtin_nan("");
  if (a <= 0.0 || b <= 0.0 || x < 0.0 || 1.0 < x) return __builtin_nan("");
  bool flip;
  double t = pt_b_body(a, b, x, true, &flip);
  if (!flip) return t;
  t = pt_b_body(b, a, 1.0 - x, false, &flip);
  return t <= PT_B_EPS ? 1.0 - PT_B_EPS : 1.0 - t;
}
PT_DEV float pt_betainc(float a, float b, float x) { return (float)pt_betainc((double)a, (double)b, (double)x); }
"""


_POLYGAMMA_SRC = r"""
// polygamma(n, x) as scipy.special.polygamma computes it (PolyGamma.impl, scalar/math.py:607-608):
// n = 0: digamma (reflection, recurrence to x >= 10, asymptotic series); n >= 1:
// (-1)^(n+1) n! zeta(n + 1, x), the Hurwitz zeta function by Euler-Maclaurin summation
// (Cephes zeta.c: direct terms until the argument exceeds 9, then 12 Bernoulli corrections)
PT_DEV double pt_zeta(double x, double q) {
  const double A[12] = {12.0, -720.0, 30240.0, -1209600.0, 47900160.0, -1.8924375803183791606e9, 7.47242496e10,
                        -2.950130727918164224e12, 1.1646782814350067249e14, -4.5979787224074726105e15,
                        1.8152105401943546773e17, -7.1661652561756670113e18};
  const double MACHEP = 1.11022302462515654042e-16;
  if (x == 1.0) return __builtin_inf();
  if (!(x >= 1.0)) return __builtin_nan("");
  if (q <= 0.0) {
    if (q == floor(q)) return __builtin_inf();
    if (x != floor(x)) return __builtin_nan("");
  }
  if (q > 1e8) return (1.0 / (x - 1.0) + 1.0 / (2.0 * q)) * pow(q, 1.0 - x);
  double s = pow(q, -x), a = q, b = 0.0;
  int i = 0;
  while (i < 9 || a <= 9.0) {
    i++;
    a += 1.0;
    b = pow(a, -x);
    s += b;
    if (fabs(b / s) < MACHEP) return s;
  }
  const double w = a;
  s += b * w / (x - 1.0);
  s -= 0.5 * b;
  a = 1.0;
  double k = 0.0;
  for (i = 0; i < 12; i++) {
    a *= x + k;
    b /= w;
    const double t = a * b / A[i];
    s += t;
    if (fabs(t / s) < MACHEP) return s;
    k += 1.0;
    a *= x + k;
    b /= w;
    k += 1.0;
  }
  return s;
}
PT_DEV double pt_digamma_acc(double x) {
  if (x != x || x == __builtin_inf()) return x;
  double nz = 0.0;
  bool neg = false;
  if (x <= 0.0) {
    if (x == floor(x)) return __builtin_nan("");
    neg = true;
    const double q = x;
    double p = floor(q);
    nz = q - p;
    if (nz != 0.5) {
      if (nz > 0.5) { p += 1.0; nz = q - p; }
      nz = 3.14159265358979323846 / tan(3.14159265358979323846 * nz);
    } else {
      nz = 0.0;
    }
    x = 1.0 - x;
  }
  double y;
  if (x <= 10.0 && x == floor(x)) {
    y = 0.0;
    for (int i = 1; i < (int)x; i++) y += 1.0 / i;
    y -= 0.57721566490153286061;
  } else {
    double s = x, w = 0.0;
    while (s < 10.0) { w += 1.0 / s; s += 1.0; }
    const double z = 1.0 / (s * s);
    double yy = 8.33333333333333333333E-2;
    yy = yy * z + -2.10927960927960927961E-2;
    yy = yy * z + 7.57575757575757575758E-3;
    yy = yy * z + -4.16666666666666666667E-3;
    yy = yy * z + 3.96825396825396825397E-3;
    yy = yy * z + -8.33333333333333333333E-3;
    yy = yy * z + 8.33333333333333333333E-2;
    yy *= z;
    y = log(s) - 0.5 / s - yy - w;
  }
  return neg ? y - nz : y;
}
PT_DEV double pt_polygamma(double n, double x) {
  if (n == 0.0) return pt_digamma_acc(x);
  if (!(n > 0.0) || n != floor(n)) return __builtin_nan("");
  const double sgn = (((long long)n) & 1) ? 1.0 : -1.0;
  return sgn * tgamma(n + 1.0) * pt_zeta(n + 1.0, x);
}
PT_DEV float pt_polygamma(float n, float x) { return (float)pt_polygamma((double)n, (double)x); }
"""

_NDTRIEXP_SRC = r"""
// ndtri(exp(y)) without forming exp(y) where it underflows (NdtriExp.impl, scalar/math.py:281:
// scipy.special.ndtri_exp): the upper tail through erfcinv(2 (1 - e^y)) near y = 0, erfcinv(2 e^y) down
// to y = -2, below that Newton steps on log Phi(x) = log(erfcx(-x / sqrt 2) / 2) - x^2 / 2 = y from the
// asymptotic root — quadratic, 3-5 steps
PT_DEV double pt_ndtri_exp(double y) {
  if (y != y || y > 0.0) return __builtin_nan("");
  if (y == 0.0) return __builtin_inf();
  if (y == -__builtin_inf()) return y;
  const double SQ2 = 1.4142135623730951;
  if (y >= -0.6931471805599453) return SQ2 * erfcinv(2.0 * (-expm1(y)));
  if (y >= -2.0) return -SQ2 * erfcinv(2.0 * exp(y));
  const double t = -2.0 * y;
  double x = -sqrt(t - log(6.283185307179586 * t));
  for (int it = 0; it < 8; it++) {
    const double r = erfcx(-x / SQ2);  // Phi(x) / phi(x) = r sqrt(pi / 2)
    const double dx = (log(0.5 * r) - 0.5 * x * x - y) * r * 1.2533141373155003;
    x -= dx;
    if (fabs(dx) <= 1e-16 * fabs(x)) break;
  }
  return x;
}
PT_DEV float pt_ndtri_exp(float y) { return (float)pt_ndtri_exp((double)y); }
"""

_GAMMAINCINV_SRC = r"""
// inverses of the regularised incomplete gamma functions (GammaIncInv / GammaIncCInv.impl,
// scalar/math.py: scipy.special.gammaincinv / gammainccinv): the root of P(a, x) = p or Q(a, x) = q,
// always posed on the smaller tail, by safeguarded Halley steps from a Wilson-Hilferty / power-law
// starting point, iterated to the last bit of the forward function above
PT_DEV double pt_gamma_tail_root(double a, double tail, bool is_upper) {
  const double EPS = 2.220446049250313e-16;
  const double lg = lgamma(a);
  double z = -1.4142135623730951 * erfcinv(2.0 * tail);
  if (is_upper) z = -z;
  const double t = 1.0 - 1.0 / (9.0 * a) + z / (3.0 * sqrt(a));
  double x = t > 0.0 ? a * t * t * t : 0.0;
  if (!is_upper && (a < 1.0 || x <= 0.0 || tail < 1e-3)) {
    const double x2 = exp((log(tail) + lg + log(a)) / a);
    if (x <= 0.0 || x2 < x) x = x2;
  }
  if (is_upper && x <= 0.0) x = fmax(-log(tail) - lg, 1e-3);
  if (!(x > 0.0) || isinf(x)) x = 1.0;
  double lo = 0.0, hi = __builtin_inf();
  for (int it = 0; it < 300; it++) {
    const double f = (is_upper ? pt_gammaincc(a, x) : pt_gammainc(a, x)) - tail;
    if (f == 0.0) return x;
    const bool right = is_upper ? f > 0.0 : f < 0.0;
    if (right) lo = fmax(lo, x); else hi = fmin(hi, x);
    const double dens = exp((a - 1.0) * log(x) - x - lg);
    double xn = -1.0;
    if (dens > 0.0 && !isinf(dens)) {
      const double r = f / (is_upper ? -dens : dens);
      const double h = 1.0 - 0.5 * r * ((a - 1.0) / x - 1.0);
      xn = x - (h > 0.5 ? r / h : r);
    }
    if (!(xn > lo && xn < hi)) xn = isinf(hi) ? 2.0 * x : (lo > 0.0 ? 0.5 * (lo + hi) : 0.5 * hi);
    if (fabs(xn - x) <= 2.0 * EPS * xn) return xn;
    x = xn;
  }
  return x;
}
PT_DEV double pt_gammaincinv(double a, double p) {
  if (!(a > 0.0) || !(p >= 0.0 && p <= 1.0)) return __builtin_nan("");
  if (p == 0.0) return 0.0;
  if (p == 1.0) return __builtin_inf();
  return p <= 0.5 ? pt_gamma_tail_root(a, p, false) : pt_gamma_tail_root(a, 1.0 - p, true);
}
PT_DEV double pt_gammainccinv(double a, double q) {
  if (!(a > 0.0) || !(q >= 0.0 && q <= 1.0)) return __builtin_nan("");
  if (q == 0.0) return __builtin_inf();
  if (q == 1.0) return 0.0;
  return q <= 0.5 ? pt_gamma_tail_root(a, q, true) : pt_gamma_tail_root(a, 1.0 - q, false);
}
PT_DEV float pt_gammaincinv(float a, float p) { return (float)pt_gammaincinv((double)a, (double)p); }
PT_DEV float pt_gammainccinv(float a, float q) { return (float)pt_gammainccinv((double)a, (double)q); }
"""

_BETAINCINV_SRC = r"""
// inverse of the regularised incomplete beta function (BetaIncInv.impl, scalar/math.py:
// scipy.special.betaincinv): the root is sought on the half of (0, 1) it lies in (t = x or 1 - x,
// decided by I_1/2), the residual is P - p or q - Q, whichever is known more precisely, and
// safeguarded Halley steps run to the last bit of the forward function above
PT_DEV double pt_betaincinv(double a, double b, double p) {
  const double EPS = 2.220446049250313e-16;
  if (!(a > 0.0 && b > 0.0) || !(p >= 0.0 && p <= 1.0)) return __builtin_nan("");
  if (p == 0.0) return 0.0;
  if (p == 1.0) return 1.0;
  double q = 1.0 - p;
  const bool flip = p > pt_betainc(a, b, 0.5);
  if (flip) { double s = a; a = b; b = s; s = p; p = q; q = s; }
  const double lbeta = lgamma(a) + lgamma(b) - lgamma(a + b);
  double t = fmin(0.5, a / (a + b));
  if (p <= 0.5) {
    const double lt = (log(p) + log(a) + lbeta) / a;
    if (lt < log(t)) t = fmax(exp(lt), 1e-300);
  }
  double lo = 0.0, hi = 1.0, res = t;
  for (int it = 0; it < 300; it++) {
    const double ld = (a - 1.0) * log(t) + (b - 1.0) * log1p(-t) - lbeta;
    const double dens = exp(ld);
    const double f = (p <= q + dens) ? pt_betainc(a, b, t) - p : q - pt_betainc(b, a, 1.0 - t);
    if (f == 0.0) { res = t; break; }
    if (f < 0.0) lo = fmax(lo, t); else hi = fmin(hi, t);
    double tn = -1.0;
    if (dens > 0.0 && !isinf(dens)) {
      const double r = f / dens;
      const double h = 1.0 - 0.5 * r * ((a - 1.0) / t - (b - 1.0) / (1.0 - t));
      tn = t - (h > 0.5 ? r / h : r);
    }
    if (!(tn > lo && tn < hi)) {
      tn = lo > 0.0 ? (hi > 4.0 * lo ? sqrt(lo * hi) : 0.5 * (lo + hi)) : 1e-3 * hi;
      if (tn <= 0.0) { res = 0.0; break; }
    }
    res = tn;
    if (fabs(tn - t) <= 2.0 * EPS * tn) break;
    t = tn;
  }
  return flip ? 1.0 - res : res;
}
PT_DEV float pt_betaincinv(float a, float b, float p) { return (float)pt_betaincinv((double)a, (double)b, (double)p); }
"""

_OPTIONAL_HELPERS = {"NdtriExp": ("ndtriexp",), "GammaInc": ("gammainc",), "GammaIncC": ("gammainc",), "BetaInc": ("betainc",), "PolyGamma": ("polygamma",),
                     "GammaIncInv": ("gammainc", "gammaincinv"), "GammaIncCInv": ("gammainc", "gammaincinv"),
                     "BetaIncInv": ("betainc", "betaincinv")}
_OPTIONAL_ORDER = ("gammainc", "betainc", "polygamma", "ndtriexp", "gammaincinv", "betaincinv")
_optional_src_cache = {}


def _optional_src(key: str) -> str:
    if key not in _optional_src_cache:
        if key == "gammainc":
            logfs, loghs = _gamma_tables()
            _optional_src_cache[key] = _c_table("pt_g_logfs", logfs) + _c_table("pt_g_loghs", loghs) + _GAMMAINC_SRC
        else:
            _optional_src_cache[key] = {"betainc": _BETAINC_SRC, "polygamma": _POLYGAMMA_SRC, "ndtriexp": _NDTRIEXP_SRC, "gammaincinv": _GAMMAINCINV_SRC,
                                        "betaincinv": _BETAINCINV_SRC}[key]
    return _optional_src_cache[key]


def prelude_for(*bodies) -> str:
    """PRELUDE plus the long helpers only the given scalar bodies need."""
    want = {k for b in bodies if b for op in body_ops(b) if op in _OPTIONAL_HELPERS for k in _OPTIONAL_HELPERS[op]}
    return PRELUDE + "".join(_optional_src(k) for k in _OPTIONAL_ORDER if k in want)


def body_ops(body: dict):
    """every scalar op name of a body, the inner bodies of its ``ScalarLoop`` nodes included"""
    for n in body["body"]:
        if n["op"] == "ScalarLoop":
            yield from body_ops(n["loop"]["body"])
        elif n["op"] != "LoopOut":
            yield n["op"]



class ScalarCodegenError(NotImplementedError):
    pass


def _lit(value, dtype: str) -> str:
    dt = np.dtype(dtype)
    if dt.kind == "f":
        v = float.fromhex(value) if isinstance(value, str) else float(value)
        if np.isnan(v):
            return "__builtin_nan(\"\")" if dt == np.float64 else "__builtin_nanf(\"\")"
        if np.isinf(v):
            s = "__builtin_huge_val()" if dt == np.float64 else "__builtin_huge_valf()"
            return s if v > 0 else f"(-{s})"
        if dt == np.float64:
            return f"{v.hex()}"  # C++17 hex float literal: bit exact
        return f"{float(np.float32(v)).hex()}f"
    if dt.kind == "b":
        return "true" if value else "false"
    v = int(value)
    if dt == np.uint64:
        return f"({v}ULL)"
    if dt == np.int64:
        return f"({v}LL)" if v != -(2**63) else "(-9223372036854775807LL - 1)"
    return f"(({CTYPE[str(dt)]}){v})"


def _is_float(dt):
    return np.dtype(dt).kind == "f"


def _is_int(dt):
    return np.dtype(dt).kind in "iu"


def _f(name64, name32=None):
    """libm-style unary: computed in the *output* dtype (upgrade_to_float ops)."""
    name32 = name32 or name64 + "f"

    def gen(args, in_dts, out_dt):
        ct = CTYPE[out_dt]
        fn = name64 if out_dt == "float64" else name32
        return f"{fn}(({ct}){args[0]})"

    return gen


def _chain(op):
    def gen(args, in_dts, out_dt):
        ct = CTYPE[out_dt]
        if out_dt == "bool":
            sym = {"+": "||", "*": "&&"}[op]
            return "(" + f" {sym} ".join(f"(bool){a}" for a in args) + ")"
        return "(" + f" {op} ".join(f"({ct}){a}" for a in args) + ")"

    return gen


def _binop_upcast(op):
    def gen(args, in_dts, out_dt):
        ct = CTYPE[out_dt]
        return f"(({ct}){args[0]} {op} ({ct}){args[1]})"

    return gen


def _cmp(op):
    def gen(args, in_dts, out_dt):
        # compare in the common type of the operands (C usual arithmetic conversions
        # differ from NumPy only for mixed signed/unsigned, which we upcast explicitly)
        common = str(np.result_type(*[np.dtype(d) for d in in_dts]))
        ct = CTYPE.get(common, "double")
        return f"(({ct}){args[0]} {op} ({ct}){args[1]})"

    return gen


def _bitop(op, boolop):
    def gen(args, in_dts, out_dt):
        if out_dt == "bool":
            return "(" + f" {boolop} ".join(f"(bool){a}" for a in args) + ")"
        ct = CTYPE[out_dt]
        return "(" + f" {op} ".join(f"({ct}){a}" for a in args) + ")"

    return gen


def _truediv(args, in_dts, out_dt):
    # TrueDiv.c_code (scalar/basic.py:1968+): discrete/discrete → (double)x / y
    ct = CTYPE[out_dt]
    return f"(({ct}){args[0]} / ({ct}){args[1]})"


def _intdiv(args, in_dts, out_dt):
    ct = CTYPE[out_dt]
    fn = "pt_intdiv_f" if _is_float(out_dt) else "pt_intdiv_i"
    return f"{fn}(({ct}){args[0]}, ({ct}){args[1]})"


def _mod(args, in_dts, out_dt):
    ct = CTYPE[out_dt]
    fn = "pt_mod_f" if _is_float(out_dt) else "pt_mod_i"
    return f"{fn}(({ct}){args[0]}, ({ct}){args[1]})"


def _pow(args, in_dts, out_dt):
    # Pow.c_code (scalar/basic.py:2250+): pow(x, y); integer outputs are cast back
    # (pt_pow: the library's pow with the exactly representable cases made exact — an integer power must not truncate
    #  6858.999999999999; float32 through the double: the rounded double is libm's powf value)
    return f"({CTYPE[out_dt]})pt_pow((double){args[0]}, (double){args[1]})"


def _abs(args, in_dts, out_dt):
    dt = in_dts[0]
    if _is_float(dt):
        return f"fabs({args[0]})" if dt == "float64" else f"fabsf({args[0]})"
    if dt in ("uint8", "uint16", "uint32", "uint64", "bool"):
        return args[0]
    return f"(({args[0]}) < 0 ? -({args[0]}) : ({args[0]}))"


def _switch(args, in_dts, out_dt):
    ct = CTYPE[out_dt]
    return f"(({args[0]}) ? ({ct}){args[1]} : ({ct}){args[2]})"


def _clip(args, in_dts, out_dt):
    ct = CTYPE[out_dt]
    x, lo, hi = (f"({ct}){a}" for a in args)
    return f"({x} < {lo} ? {lo} : ({x} > {hi} ? {hi} : {x}))"


def _cast(args, in_dts, out_dt):
    # Cast.c_code (scalar/basic.py:2435+)
    if out_dt == "bool":
        return f"(({args[0]}) ? true : false)"
    return f"({CTYPE[out_dt]}){args[0]}"


def _maxmin(fn):
    def gen(args, in_dts, out_dt):
        ct = CTYPE[out_dt]
        e = f"({ct}){args[0]}"
        for a in args[1:]:
            e = f"{fn}({e}, ({ct}){a})"
        return e

    return gen


def _isnan(args, in_dts, out_dt):
    return f"isnan({args[0]})" if _is_float(in_dts[0]) else "false"


def _isinf(args, in_dts, out_dt):
    return f"isinf({args[0]})" if _is_float(in_dts[0]) else "false"


def _invert(args, in_dts, out_dt):
    return f"(!{args[0]})" if out_dt == "bool" else f"(({CTYPE[out_dt]})~{args[0]})"


def _helper(fn):
    def gen(args, in_dts, out_dt):
        ct = CTYPE[out_dt]
        return f"{fn}(" + ", ".join(f"({ct}){a}" for a in args) + ")"

    return gen


_FAST_LOG = os.environ.get("PTHIP_FAST_LOG", "1") != "0"  # diagnostic: 0 = the device library's log / log1p

# op name (reference ScalarOp class) → expression generator
SCALAR_EXPR = {
    "Add": _chain("+"),  # scalar/basic.py:1835 Add.c_code
    "Mul": _chain("*"),  # 1876
    "Sub": _binop_upcast("-"),  # 1937
    "TrueDiv": _truediv,  # 1968
    "IntDiv": _intdiv,
    "Mod": _mod,
    "Pow": _pow,  # 2250
    "Neg": lambda a, i, o: f"(-({CTYPE[o]}){a[0]})",
    "Abs": _abs,  # 2524
    "Sign": _helper("pt_sign"),  # 2575
    "Sqr": _helper("pt_sqr"),  # 3202
    "Sqrt": _f("sqrt"),  # 3231
    "Exp": _f("pt_exp" if os.environ.get("PTHIP_FAST_EXP", "1") != "0" else "exp", "expf"),  # 3085
    "Exp2": _f("exp2"),
    "Expm1": _f("expm1"),
    "Log": _f("pt_log" if _FAST_LOG else "log", "logf"),  # 2907
    "Log2": _f("log2"),
    "Log10": _f("log10"),
    "Log1p": _f("pt_log1p" if _FAST_LOG else "log1p", "log1pf"),  # 3042
    "Sin": _f("sin"),
    "Cos": _f("cos"),
    "Tan": _f("tan"),
    "ArcSin": _f("asin"),
    "ArcCos": _f("acos"),
    "ArcTan": _f("atan"),
    "ArcTan2": lambda a, i, o: (
        f"{'atan2' if o == 'float64' else 'atan2f'}(({CTYPE[o]}){a[0]}, ({CTYPE[o]}){a[1]})"
    ),
    "Sinh": _f("sinh"),
    "Cosh": _f("cosh"),
    "Tanh": _f("pt_tanh" if os.environ.get("PTHIP_FAST_TANH", "1") != "0" else "tanh", "tanhf"),  # 3702
    "ArcSinh": _f("asinh"),
    "ArcCosh": _f("acosh"),
    "ArcTanh": _f("atanh"),
    "Sigmoid": _helper("pt_sigmoid"),  # scalar/math.py:1187-1198
    "Softplus": _helper("pt_softplus"),  # scalar/math.py:1250-1277
    "Log1mexp": _helper("pt_log1mexp"),  # scalar/math.py:1295+
    "Erf": _f("erf"),  # scalar/math.py:55
    "Erfc": _f("erfc"),  # 91
    "Erfinv": _f("erfinv"),
    "Erfcinv": _f("erfcinv"),
    "Erfcx": _f("erfcx"),
    "GammaLn": _f("lgamma"),  # scalar/math.py:363
    "Gamma": _f("tgamma"),
    "Psi": _helper("pt_psi"),  # scalar/math.py:403
    "TriGamma": _helper("pt_trigamma"),  # scalar/math.py:502
    "GammaInc": _helper("pt_gammainc"),  # scalar/math.py:627
    "GammaIncC": _helper("pt_gammaincc"),  # scalar/math.py:674
    "BetaInc": _helper("pt_betainc"),  # scalar/math.py:1342
    "PolyGamma": _helper("pt_polygamma"),  # scalar/math.py:595 (scipy.special.polygamma)
    "NdtriExp": _helper("pt_ndtri_exp"),  # scalar/math.py:271 (scipy.special.ndtri_exp)
    "GammaIncInv": _helper("pt_gammaincinv"),  # scipy.special.gammaincinv
    "GammaIncCInv": _helper("pt_gammainccinv"),  # scipy.special.gammainccinv
    "BetaIncInv": _helper("pt_betaincinv"),  # scipy.special.betaincinv
    # Bessel functions: J0/J1.c_code call libm's j0/j1 in double (scalar/math.py:1011-1064);
    # I0/I1 have no C code, the reference evaluates scipy.special.i0/i1 (1066-1110)
    "J0": lambda a, i, o: f"({CTYPE[o]})j0((double){a[0]})",
    "J1": lambda a, i, o: f"({CTYPE[o]})j1((double){a[0]})",
    "I0": lambda a, i, o: f"({CTYPE[o]})cyl_bessel_i0((double){a[0]})",
    "I1": lambda a, i, o: f"({CTYPE[o]})cyl_bessel_i1((double){a[0]})",
    "Reciprocal": lambda a, i, o: f"(({CTYPE[o]})1 / ({CTYPE[o]}){a[0]})",
    "Maximum": _maxmin("pt_max"),  # 1744
    "Minimum": _maxmin("pt_min"),  # 1790
    "ScalarMaximum": _maxmin("pt_max"),
    "ScalarMinimum": _maxmin("pt_min"),
    "EQ": _cmp("=="),  # 1411-1530
    "NEQ": _cmp("!="),
    "LT": _cmp("<"),
    "GT": _cmp(">"),
    "LE": _cmp("<="),
    "GE": _cmp(">="),
    "AND": _bitop("&", "&&"),
    "OR": _bitop("|", "||"),
    "XOR": _bitop("^", "!="),
    "Invert": _invert,
    "IsNan": _isnan,
    "IsInf": _isinf,
    "Switch": _switch,  # 1588
    "Clip": _clip,  # 2335
    "Identity": lambda a, i, o: f"({CTYPE[o]}){a[0]}",
    "Second": lambda a, i, o: f"({CTYPE[o]}){a[1]}",
    "Floor": _f("floor"),
    "Ceil": _f("ceil"),
    "Trunc": _f("trunc"),
    "RoundHalfToEven": _helper("pt_rint_even"),
    "RoundHalfAwayFromZero": _f("round"),
    "Cast": _cast,  # 2435
    "Deg2Rad": lambda a, i, o: f"(({CTYPE[o]}){a[0]} * ({CTYPE[o]})0.017453292519943295)",
    "Rad2Deg": lambda a, i, o: f"(({CTYPE[o]}){a[0]} * ({CTYPE[o]})57.29577951308232)",
}


def supported(body: dict) -> bool:
    def dtypes(b):
        yield from b["in_dtypes"] + b["out_dtypes"]
        for n in b["body"]:
            if n["op"] == "ScalarLoop":
                yield from dtypes(n["loop"]["body"])

    return all(op in SCALAR_EXPR for op in body_ops(body)) and all(d in CTYPE for d in dtypes(body))


_SHARE_SIG_SP = os.environ.get("PTHIP_SHARE_SIG_SP", "1") != "0"
_EMIT_CTX = {"share_recip": False}  # set by flat_kernel_source for bodies all of whose outputs are summed


def emit_body(body: dict, in_names, out_names, indent="      ", tp="t") -> str:
    """SSA statements computing ``out_names`` from ``in_names`` (one element).  ``tp`` prefixes
    the temporaries (the inner body of a loop lives in a nested scope with its own prefix)."""
    lines = []
    tdt = []

    def ref(r):
        if r[0] == "i":
            return in_names[r[1]], body["in_dtypes"][r[1]]
        if r[0] == "t":
            return f"{tp}{r[1]}", tdt[r[1]]
        return _lit(r[1], r[2]), r[2]

    # sigmoid and softplus of the same float64 operand: one shared evaluation (pt_sig_sp)
    shared = {}
    if _SHARE_SIG_SP:
        by_arg = {}
        for k, n in enumerate(body["body"]):
            if n["op"] in ("Sigmoid", "Softplus") and n["dtype"] == "float64" and len(n["in"]) == 1 and n["in"][0][0] in ("i", "t"):
                src_dt = body["in_dtypes"][n["in"][0][1]] if n["in"][0][0] == "i" else body["body"][n["in"][0][1]]["dtype"]
                if src_dt == "float64":
                    by_arg.setdefault((n["in"][0][0], n["in"][0][1]), {}).setdefault(n["op"], k)
        for d in by_arg.values():
            if len(d) == 2:
                first = min(d.values())
                for op, k in d.items():
                    shared[k] = (first, "sg" if op == "Sigmoid" else "sp")
    recip = {}  # TrueDiv node -> first node of its denominator group
    if _EMIT_CTX["share_recip"]:
        by_den = {}
        for k, n in enumerate(body["body"]):
            if n["op"] == "TrueDiv" and n["dtype"] == "float64" and len(n["in"]) == 2 and n["in"][1][0] in ("i", "t"):
                den = n["in"][1]
                den_dt = body["in_dtypes"][den[1]] if den[0] == "i" else body["body"][den[1]]["dtype"]
                num = n["in"][0]
                num_dt = (body["in_dtypes"][num[1]] if num[0] == "i" else body["body"][num[1]]["dtype"]) if num[0] in ("i", "t") else num[2]
                if den_dt == "float64" and num_dt == "float64":
                    by_den.setdefault((den[0], den[1]), []).append(k)
        for ks in by_den.values():
            if len(ks) >= 2:
                for k in ks:
                    recip[k] = ks[0]
    for k, n in enumerate(body["body"]):
        ct = CTYPE[n["dtype"]]
        if k in recip:
            first = recip[k]
            if k == first:
                den, _ = ref(n["in"][1])
                lines.append(f"{indent}const double {tp}{first}_rcp = 1.0 / (double){den};")
            num, _ = ref(n["in"][0])
            lines.append(f"{indent}const {ct} {tp}{k} = ({ct})((double){num} * {tp}{first}_rcp);")
            tdt.append(n["dtype"])
            continue
        if k in shared:
            first, which = shared[k]
            if k == first:
                arg, _ = ref(n["in"][0])
                lines.append(f"{indent}double {tp}{first}_sg, {tp}{first}_sp; pt_sig_sp((double){arg}, {tp}{first}_sg, {tp}{first}_sp);")
            lines.append(f"{indent}const {ct} {tp}{k} = {tp}{first}_{which};")
            tdt.append(n["dtype"])
            continue
        if n["op"] == "ScalarLoop":
            lines.append(_emit_loop(n, [ref(r) for r in n["in"]], f"{tp}{k}_", indent))
            lines.append(f"{indent}const {ct} {tp}{k} = {tp}{k}_s0;")
        elif n["op"] == "LoopOut":
            assert n["in"][0][0] == "t" and body["body"][n["in"][0][1]]["op"] == "ScalarLoop"
            loop = body["body"][n["in"][0][1]]["loop"]
            which = "done" if (loop["is_while"] and n["k"] == loop["n_state"]) else f"s{n['k']}"
            lines.append(f"{indent}const {ct} {tp}{k} = ({ct}){tp}{n['in'][0][1]}_{which};")
        else:
            gen = SCALAR_EXPR.get(n["op"])
            if gen is None:
                raise ScalarCodegenError(f"no device expression for scalar op {n['op']}")
            pairs = [ref(r) for r in n["in"]]
            odt = n["dtype"]
            if odt == "float16" or any(p[1] == "float16" for p in pairs):
                # half is a storage type: operands widen to float, the op runs in float and the
                # assignment below rounds to half — bit-identical to IEEE half +,-,*,/ (float has
                # 24 >= 2*11+2 significand bits, so the double rounding is innocuous) and what
                # NumPy does for every float16 ufunc
                pairs = [(f"(float){a}", "float32") if d == "float16" else (a, d) for a, d in pairs]
                odt = "float32" if odt == "float16" else odt
            expr = gen([p[0] for p in pairs], [p[1] for p in pairs], odt)
            lines.append(f"{indent}const {ct} {tp}{k} = ({ct})({expr});")
        tdt.append(n["dtype"])
    for name, r, dt in zip(out_names, body["outs"], body["out_dtypes"]):
        e, _ = ref(r)
        lines.append(f"{indent}{name} = ({CTYPE[dt]})({e});")
    return "\n".join(lines)


def _emit_loop(n: dict, pairs, P: str, indent: str) -> str:
    """``ScalarLoop.c_code_template`` (pytensor/scalar/loop.py:181-290) restated: carried
    copies of the initial states, ``for (i < n_steps)`` around the inner body, the carries
    overwritten after the whole body ran, ``until`` starting true and breaking after the update."""
    loop = n["loop"]
    inner = loop["body"]
    S = loop["n_state"]
    L = []
    for j in range(S):
        ct = CTYPE[inner["in_dtypes"][j]]
        L.append(f"{indent}{ct} {P}s{j} = ({ct})({pairs[1 + j][0]});")
    names = [f"{P}s{j}" for j in range(S)]
    for j in range(S, len(inner["in_dtypes"])):
        ct = CTYPE[inner["in_dtypes"][j]]
        L.append(f"{indent}const {ct} {P}c{j} = ({ct})({pairs[1 + j][0]});")
        names.append(f"{P}c{j}")
    if loop["is_while"]:
        L.append(f"{indent}bool {P}done = true;")
    L.append(f"{indent}for (long long {P}it = 0, {P}n = (long long)({pairs[0][0]}); {P}it < {P}n; ++{P}it) {{")
    ind2 = indent + "  "
    outs = []
    for j, dt in enumerate(inner["out_dtypes"]):
        L.append(f"{ind2}{CTYPE[dt]} {P}n{j};")
        outs.append(f"{P}n{j}")
    L.append(emit_body(inner, names, outs, ind2, tp=P + "t"))
    for j in range(S):
        L.append(f"{ind2}{P}s{j} = {P}n{j};")
    if loop["is_while"]:
        L.append(f"{ind2}{P}done = {P}n{S};")
        L.append(f"{ind2}if ({P}done) break;")
    L.append(f"{indent}}}")
    return "\n".join(L)


# ---------------------------------------------------------------------------
# kernels
# ---------------------------------------------------------------------------

BLOCK = 256
REDUCE_OPS = {"Add": "OpAdd", "Mul": "OpMul", "Maximum": "OpMax", "Minimum": "OpMin"}


def _vec_width(dtypes) -> int:
    """elements per 16-byte pack, limited by the widest participating dtype"""
    w = max(np.dtype(d).itemsize for d in dtypes)
    return max(1, 16 // w)


def _vec_type(ctype: str, n: int) -> str:
    return f"pt_vec<{ctype}, {n}>"


VEC_HELPERS = r"""
template <class T, int N> struct __attribute__((aligned(sizeof(T) * N))) pt_vec { T v[N]; };
template <class P> static __device__ __forceinline__ P pthip_nt_pack(const P* p) {
  typedef unsigned int pt_u4 __attribute__((ext_vector_type(4)));
  P o;
  if constexpr (sizeof(P) == 16) { const pt_u4 r = __builtin_nontemporal_load((const pt_u4*)p); __builtin_memcpy(&o, &r, 16); }
  else if constexpr (sizeof(P) == 8) { const unsigned long long r = __builtin_nontemporal_load((const unsigned long long*)p); __builtin_memcpy(&o, &r, 8); }
  else if constexpr (sizeof(P) == 4) { const unsigned int r = __builtin_nontemporal_load((const unsigned int*)p); __builtin_memcpy(&o, &r, 4); }
  else o = *p;
  return o;
}
"""


def _stream_load(ptr_expr: str, struct=False) -> str:
    """Load of an operand that is streamed once (the vector inputs of a flat fused kernel, the
    matrix of ``gchain``): non-temporal (``nt``) — the line is not retained in L2/MALL, which is
    what a pass over 0.16-1 GB wants (MI355X_MICROARCH.md price list, nt-weights row).  Measured
    (profiles/r2f_nt_loads.txt): ``gchain`` 184.5 -> 171.2 us (5.64 -> 6.08 TB/s), config #4
    4072 -> 4366 evals/s; 52-op Composite+Sum at N=1e7 39.9 -> 38.1 us.  ``PTHIP_NT_LOADS=0``
    restores plain loads."""
    if os.environ.get("PTHIP_NT_LOADS", "1") != "0":
        if struct:  # the pack types are structs: reinterpret as one 16/8/4-byte word
            return f"pthip_nt_pack({ptr_expr})"
        return f"__builtin_nontemporal_load({ptr_expr})"
    return f"*({ptr_expr})"


def _flat_params(body: dict, modes: str, reduce_spec, vec: int, finish=None):
    params = ["long long n"]
    for k, dt in enumerate(body["in_dtypes"]):
        if modes[k] == "C":  # host-known scalar: travels by value in the argument block
            params.append(f"const long long in{k}")
        elif modes[k] == "G":  # in{k}[gx{k}[i]]: a gather (AdvancedSubtensor on axis 0) read in the loop
            params += [f"const {CTYPE[dt]}* __restrict__ in{k}", f"const long long* __restrict__ gx{k}", f"long long gn{k}"]
        else:
            params.append(f"const {CTYPE[dt]}* __restrict__ in{k}")
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None:
            params.append(f"{CTYPE[dt]}* __restrict__ out{k}")
        else:
            params.append(f"{CTYPE[reduce_spec[k][1]]}* __restrict__ part{k}")
    if "G" in modes:
        assert vec == 1, "gather inputs use the scalar loop"
        params.append("int* __restrict__ status")
    if finish:
        for k, dt in enumerate(finish):
            if dt is not None:
                params.append(f"{CTYPE[dt]}* __restrict__ fin{k}")
        params += ["int* __restrict__ ticket", "int* __restrict__ pt_status"]
    return params


def flat_kernel_source(name: str, body: dict, modes: str, vec: int, reduce_spec=None, unroll=2, device_fn=False, prefetch=False, finish=None, finish_per_thread=None) -> str:
    """:func:`_flat_kernel_source` with the emit context set: when EVERY output of the body is summed (a logp term and its
    gradients in a many-term launch, a fused Elemwise + Sum) no element is ever seen — only sums over ~1e6 of them, held to
    rtol 1e-12 — so divisions that share a float64 denominator may share ONE reciprocal (x * (1/d) is within 1.5 ulp of
    x / d; an fp64 division is ~28 VALU instructions).  Never when an element-wise output is stored: 6 * (1/3) != 2."""
    all_summed = bool(reduce_spec) and all(rs is not None and rs[0] == "Add" and rs[1] == "float64" for rs in reduce_spec)
    old = _EMIT_CTX["share_recip"]
    _EMIT_CTX["share_recip"] = bool(all_summed and os.environ.get("PTHIP_SHARE_RECIP", "1") != "0")
    try:
        return _flat_kernel_source(name, body, modes, vec, reduce_spec, unroll, device_fn, prefetch, finish, finish_per_thread)
    finally:
        _EMIT_CTX["share_recip"] = old


def _flat_kernel_source(name: str, body: dict, modes: str, vec: int, reduce_spec=None, unroll=2, device_fn=False, prefetch=False, finish=None, finish_per_thread=None) -> str:
    """``flat`` loop.  ``modes[k]`` ∈ {'V' contiguous vector, 'S' scalar broadcast} per input.

    reduce_spec: None or list (per output) of None | (op_name, acc_dtype): reduced
    outputs are accumulated instead of stored; the kernel then takes one partial
    pointer per reduced output (laid out [gridDim.x]).

    ``device_fn``: emit the loop as a ``__device__`` function taking the workgroup index and count as
    two trailing arguments, without the headers (``multi_flat_source`` dispatches several of them
    from one launch).

    ``prefetch``: software-pipelined main loop — the packs of iteration i+1 are requested before the
    scalar graph of iteration i runs, so a wave keeps ``unroll`` x 16 B per operand in flight WHILE
    it computes.  For a long fp64 body (BASELINE config #2: ~270 VALU instructions per 4 elements,
    17 us of issue next to 25 us of HBM time) the plain loop alternates between the two — every
    wave either waits or computes, and the bytes in flight per CU drop with the share of waves
    that are computing; pipelined, the two overlap.
    """
    nin = len(body["in_dtypes"])
    nout = len(body["out_dtypes"])
    reduce_spec = reduce_spec or [None] * nout
    params = _flat_params(body, modes, reduce_spec, vec, finish)
    if device_fn:
        # (inlined into every case of the dispatching switch: as ONE shared copy per family (__noinline__) the kernel of
        #  north_star's 48-term graph shrank from 40,000 to 5,500 instructions but needed 99 VGPRs instead of 50 for the
        #  call and ran 218 us instead of 196 — measured, profiles/r5m notes in DESIGN.md)
        src = [f"static __device__ __forceinline__ void {name}({', '.join(params)}, const unsigned pt_bidx, const unsigned pt_gdim) {{"]
    else:
        src = [reduce_header() if any(reduce_spec) else "", prelude_for(body), VEC_HELPERS, PT_PAIR_HELPERS if finish else ""]
        src.append(f'extern "C" __global__ __launch_bounds__({BLOCK}) void {name}({", ".join(params)}) {{')
    # scalars
    for k, m in enumerate(modes):
        if m == "S":
            src.append(f"  const {CTYPE[body['in_dtypes'][k]]} s{k} = in{k}[0];")
        elif m == "C":
            ct = CTYPE[body["in_dtypes"][k]]
            src.append(f"  {ct} s{k}; {{ const long long b = in{k}; __builtin_memcpy(&s{k}, &b, sizeof({ct})); }}")
    for k, rs in enumerate(reduce_spec):
        if rs is not None:
            act = CTYPE[rs[1]]
            for u in range(unroll):
                src.append(f"  {act} acc{k}_{u} = pthip_dev::{REDUCE_OPS[rs[0]]}::identity<{act}>();")
    src.append(f"  const long long tid = (long long)blockIdx.x * {BLOCK} + threadIdx.x;")
    src.append(f"  const long long nthreads = (long long)gridDim.x * {BLOCK};")
    V = vec
    if V > 1:
        src.append(f"  const long long npack = n / {V};")
        # main vector loop, `unroll` packs per iteration
        src.append(f"  long long p = tid;")
        vins = [(k, CTYPE[body["in_dtypes"][k]]) for k, m in enumerate(modes) if m == "V"]

        def _ld(k, ct, u, pv):
            return _stream_load(f'reinterpret_cast<const {_vec_type(ct, V)}*>(in{k}) + ({pv} + {u} * nthreads)', struct=True)

        if prefetch and vins:
            for u in range(unroll):
                for k, ct in vins:
                    src.append(f"  {_vec_type(ct, V)} n{k}_{u};")
            src.append(f"  bool pt_more = p + {unroll - 1} * nthreads < npack;")
            src.append("  if (pt_more) {")
            for u in range(unroll):
                for k, ct in vins:
                    src.append(f"    n{k}_{u} = {_ld(k, ct, u, 'p')};")
            src.append("  }")
            src.append("  while (pt_more) {")
            for u in range(unroll):
                for k, ct in vins:
                    src.append(f"    const {_vec_type(ct, V)} a{k}_{u} = n{k}_{u};")
            src.append(f"    const long long pn = p + {unroll} * nthreads;")
            src.append(f"    pt_more = pn + {unroll - 1} * nthreads < npack;")
            src.append("    if (pt_more) {")
            for u in range(unroll):
                for k, ct in vins:
                    src.append(f"      n{k}_{u} = {_ld(k, ct, u, 'pn')};")
            src.append("    }")
            # (the machine scheduler would otherwise sink the requests next to their first use)
            src.append("    __builtin_amdgcn_sched_barrier(0);")
        else:
            src.append(f"  for (; p + {unroll - 1} * nthreads < npack; p += {unroll} * nthreads) {{")
            for u in range(unroll):
                for k, ct in vins:
                    src.append(f"    const {_vec_type(ct, V)} a{k}_{u} = {_ld(k, ct, u, 'p')};")
        for u in range(unroll):
            for k, dt in enumerate(body["out_dtypes"]):
                if reduce_spec[k] is None:
                    src.append(f"    {_vec_type(CTYPE[dt], V)} r{k}_{u};")
            src.append(f"#pragma unroll\n    for (int e = 0; e < {V}; e++) {{")
            in_names = [(f"a{k}_{u}.v[e]" if m == "V" else f"s{k}") for k, m in enumerate(modes)]
            out_names = []
            for k, dt in enumerate(body["out_dtypes"]):
                if reduce_spec[k] is None:
                    out_names.append(f"r{k}_{u}.v[e]")
                else:
                    src.append(f"      {CTYPE[dt]} o{k};")
                    out_names.append(f"o{k}")
            src.append(emit_body(body, in_names, out_names))
            for k, rs in enumerate(reduce_spec):
                if rs is not None:
                    src.append(f"      acc{k}_{u} = pthip_dev::{REDUCE_OPS[rs[0]]}::apply(acc{k}_{u}, ({CTYPE[rs[1]]})o{k});")
            src.append("    }")
            for k, dt in enumerate(body["out_dtypes"]):
                if reduce_spec[k] is None:
                    src.append(f"    reinterpret_cast<{_vec_type(CTYPE[dt], V)}*>(out{k})[p + {u} * nthreads] = r{k}_{u};")
        if prefetch and vins:
            src.append("    p = pn;")
        src.append("  }")
        # remaining packs one at a time, then the scalar tail
        src.append(f"  for (; p < npack; p += nthreads) {{")
        src.append(_flat_scalar_block(body, modes, reduce_spec, V, "p"))
        src.append("  }")
        src.append(f"  for (long long i = npack * {V} + tid; i < n; i += nthreads) {{")
        src.append(_flat_elem(body, modes, reduce_spec, "i"))
        src.append("  }")
    else:
        src.append(f"  for (long long i = tid; i < n; i += nthreads) {{")
        src.append(_flat_elem(body, modes, reduce_spec, "i"))
        src.append("  }")
    src.append(_reduce_epilogue(reduce_spec, unroll, finish, finish_per_thread))
    src.append("}")
    text = "\n".join(src)
    if device_fn:
        text = text.replace("blockIdx.x", "pt_bidx").replace("gridDim.x", "pt_gdim")
    return text


def multi_finish_layout(nred: int, groups: int):
    """One term's block of a self-finishing many-term launch, in 8-byte words: ``nred`` pair arrays of ``2 * groups`` words,
    then the finished values; returns ``(offset of the finals, words in all — even, so the next term's pairs stay 16-byte
    aligned)``."""
    fin = 2 * groups * nred
    return fin, (fin + nred + 1) // 2 * 2


def multi_flat_source(name: str, terms, finish: bool = False) -> str:
    """Several independent flat kernels in ONE launch (widefuse.fuse_independent_reductions):
    ``(blockIdx.x + blockIdx.y) % gridDim.x`` selects the term, ``blockIdx.y`` / the term's ``groups`` are the
    workgroup index / count within it.  ``terms``: ``[{body, modes, vec, rs, unroll}]``; terms with the same (body, modes,
    vec, reductions) share one device function.  Arguments: the terms' flat-kernel arguments, one
    term after the other.

    ``finish``: every term finishes its own reductions (``_reduce_epilogue``'s one-pass form: the term's last workgroup
    folds its <= BLOCK pairs).  A term's pair arrays and finished values then live in ONE block (``multi_finish_layout``)
    passed as a single pointer, its ticket is ``tickets[term]`` and the status word is shared: the argument block shrinks
    instead of growing (4 KB limit: north_star's 48 terms)."""
    bodies = [t["body"] for t in terms]
    head = [reduce_header(), prelude_for(*bodies), VEC_HELPERS, PT_PAIR_HELPERS if finish else ""]
    fns, fn_of = {}, []
    for t in terms:
        key = source_key(repr((t["body"], t["modes"], t["vec"], t["rs"], t["unroll"], bool(t.get("prefetch")), finish)))
        if key not in fns:
            fname = f"mt_{key[:12]}"
            fin = [rs[1] for rs in t["rs"]] if finish else None  # (finished in the accumulator dtype: 8-byte words)
            fns[key] = (fname, flat_kernel_source(fname, t["body"], t["modes"], t["vec"], t["rs"], t["unroll"], device_fn=True, prefetch=bool(t.get("prefetch")),
                                                  finish=fin, finish_per_thread=1 if finish else None))
        fn_of.append(fns[key][0])
    P, calls = [], []
    for ti, t in enumerate(terms):
        ps = _flat_params(t["body"], t["modes"], t["rs"], t["vec"])
        names = []
        nred = len(t["rs"])
        for q, prm in enumerate(ps):
            decl, nm = prm.rsplit(" ", 1)
            if finish and nm.startswith("part"):
                if nm == "part0":
                    P.append(f"double* __restrict__ t{ti}_blk")
                k = int(nm[4:])
                names.append(f"({decl})(t{ti}_blk + {2 * int(t['groups']) * k})")
                continue
            P.append(f"{decl} t{ti}_{nm}")
            names.append(f"t{ti}_{nm}")
        gt = t.get("groups")  # this term's workgroup count (cost-proportional, dispatch/wide.py); default: the whole grid column
        if finish:
            fin0, _ = multi_finish_layout(nred, int(gt))
            names += [f"({CTYPE[rs[1]]}*)(t{ti}_blk + {fin0 + k})" for k, rs in enumerate(t["rs"])] + [f"pt_tickets + {ti}", "pt_status"]
        if gt:
            calls.append(f"    case {ti}: if (blockIdx.y < {int(gt)}) {fn_of[ti]}({', '.join(names)}, blockIdx.y, {int(gt)}u); break;")
        else:
            calls.append(f"    case {ti}: {fn_of[ti]}({', '.join(names)}, blockIdx.y, gridDim.y); break;")
    if finish:
        P += ["int* __restrict__ pt_tickets", "int* __restrict__ pt_status"]
    L = head + [f for _, f in fns.values()]
    L.append(f'extern "C" __global__ __launch_bounds__({BLOCK}) void {name}({", ".join(P)}) {{')
    # grid (terms, workgroups per term), the term rotated by the round: consecutive workgroup ids — which the
    # dispatcher deals out round-robin over XCDs and CUs — belong to different terms AND the ids a CU's slots receive
    # (c, c + 256, ...) to different families.  With the term on blockIdx.y and four families cycling through the terms
    # every CU's eight slots held the SAME family: the CUs of the expensive one ran 3x longer than the rest idled
    # (north_star's 48-term graph: mean occupancy 7 of 32 waves per CU, profiles/r5h_wide_multi_pmc.md).
    L.append("  switch ((blockIdx.x + blockIdx.y) % gridDim.x) {")
    L += calls
    L.append("    default: break;")
    L.append("  }")
    L.append("}")
    return "\n".join(L)


def _flat_scalar_block(body, modes, reduce_spec, V, pvar):
    lines = []
    for k, m in enumerate(modes):
        if m == "V":
            ct = CTYPE[body["in_dtypes"][k]]
            lines.append(f"    const {_vec_type(ct, V)} a{k} = reinterpret_cast<const {_vec_type(ct, V)}*>(in{k})[{pvar}];")
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None:
            lines.append(f"    {_vec_type(CTYPE[dt], V)} r{k};")
    lines.append(f"#pragma unroll\n    for (int e = 0; e < {V}; e++) {{")
    in_names = [(f"a{k}.v[e]" if m == "V" else f"s{k}") for k, m in enumerate(modes)]
    out_names = []
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None:
            out_names.append(f"r{k}.v[e]")
        else:
            lines.append(f"      {CTYPE[dt]} o{k};")
            out_names.append(f"o{k}")
    lines.append(emit_body(body, in_names, out_names))
    for k, rs in enumerate(reduce_spec):
        if rs is not None:
            lines.append(f"      acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::apply(acc{k}_0, ({CTYPE[rs[1]]})o{k});")
    lines.append("    }")
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None:
            lines.append(f"    reinterpret_cast<{_vec_type(CTYPE[dt], V)}*>(out{k})[{pvar}] = r{k};")
    return "\n".join(lines)


def _flat_elem(body, modes, reduce_spec, ivar):
    lines = []
    in_names = []
    for k, m in enumerate(modes):
        if m == "V":
            in_names.append(f"in{k}[{ivar}]")
        elif m == "G":
            # NumPy index semantics: negative wraps once, out of range is an IndexError (raised
            # by the host from the device flag; the lane reads entry 0 meanwhile)
            lines.append(f"      long long gi{k} = gx{k}[{ivar}];")
            lines.append(f"      if (gi{k} < 0) gi{k} += gn{k};")
            lines.append(f"      if (gi{k} < 0 || gi{k} >= gn{k}) {{ atomicOr(status, 1); gi{k} = 0; }}")
            in_names.append(f"in{k}[gi{k}]")
        else:
            in_names.append(f"s{k}")
    out_names = []
    for k, dt in enumerate(body["out_dtypes"]):
        lines.append(f"      {CTYPE[dt]} o{k};")
        out_names.append(f"o{k}")
    lines.append(emit_body(body, in_names, out_names))
    for k, rs in enumerate(reduce_spec):
        if rs is None:
            lines.append(f"      out{k}[{ivar}] = o{k};")
        else:
            lines.append(f"      acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::apply(acc{k}_0, ({CTYPE[rs[1]]})o{k});")
    return "\n".join(lines)


FINISH_MAX_PER_THREAD = 8  # pairs a thread of the last workgroup folds: one-pass reductions need gridDim.x <= 8 * BLOCK

PT_PAIR_HELPERS = r"""
// one-pass reductions: a workgroup's partial travels as a self-validating 16-byte pair {bits, bits ^ MAGIC} in one
// write-through store; the last workgroup polls the pairs (agent-scope loads) — no fence anywhere.  (A fence per
// workgroup — __threadfence() before a ticket — made BASELINE config #2 ten times slower: every agent-scope
// release / acquire writes back and invalidates the XCD's L2 under 2000 streaming workgroups,
// profiles/r4i_c2_ab.txt.)
typedef unsigned long long pt_u64;
typedef pt_u64 pt_u2 __attribute__((ext_vector_type(2)));
static constexpr pt_u64 PT_PAIR_MAGIC = 0x7ff4c0de5ea1ed03ull;
static __device__ __forceinline__ void pt_pair_store(pt_u64* slot, pt_u64 lo, pt_u64 hi) {
  pt_u2 pr = {lo, hi};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" : : "v"((pt_u2*)slot), "v"(pr) : "memory");
}
static __device__ __forceinline__ bool pt_pair_poll(const pt_u64* slot, pt_u64& bits) {
  bits = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const pt_u64 b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (bits ^ b) == PT_PAIR_MAGIC;
}
"""


def _reduce_epilogue(reduce_spec, unroll, finish=None, per_thread=None):
    """``finish`` (per output: final dtype | None): ONE pass — every workgroup publishes its partial as a
    self-validating pair and takes a ticket (a relaxed device atomic); the last one to arrive polls all pairs, folds
    them in a fixed order (thread t takes partials t, t+BLOCK, ... in order, then the block combine: deterministic
    for a given grid), stores the final value and zeroes the pairs for the next launch of this kernel.  The
    second-stage launch (~3 us + a launch gap behind a 30 us streaming kernel) goes away, and no fence is needed.
    ``part{k}`` is then the pair array (2 x 8 bytes per workgroup)."""
    if not any(reduce_spec):
        return ""
    FM = per_thread or FINISH_MAX_PER_THREAD
    lines = []
    for k, rs in enumerate(reduce_spec):
        if rs is None:
            continue
        act = CTYPE[rs[1]]
        op = f"pthip_dev::{REDUCE_OPS[rs[0]]}"
        lines.append(f"  __shared__ {act} smem{k}[{BLOCK // 64}];")
        e = f"acc{k}_0"
        for u in range(1, unroll):
            e = f"{op}::apply({e}, acc{k}_{u})"
        lines.append(f"  {act} tot{k} = pthip_dev::block_reduce<{op}, {act}, {BLOCK}>({e}, smem{k});")
        if finish:
            lines.append(f"  if (threadIdx.x == 0) {{ pt_u64 b = 0; __builtin_memcpy(&b, &tot{k}, sizeof(tot{k})); pt_pair_store((pt_u64*)part{k} + 2 * blockIdx.x, b, b ^ PT_PAIR_MAGIC); }}")
        else:
            lines.append(f"  if (threadIdx.x == 0) part{k}[blockIdx.x] = tot{k};")
    if finish:
        lines.append("  __shared__ int pt_last;")
        lines.append("  if (threadIdx.x == 0) pt_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;")
        lines.append("  __syncthreads();")
        lines.append("  if (pt_last) {")
        lines.append("    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);")
        for k, rs in enumerate(reduce_spec):
            if rs is None:
                continue
            act = CTYPE[rs[1]]
            op = f"pthip_dev::{REDUCE_OPS[rs[0]]}"
            lines.append(f"    {act} fa{k} = {op}::identity<{act}>();")
            lines.append("    {")
            # all of a thread's pairs are requested before the first is examined: one memory round trip for the
            # whole fold (polled one after the other the 8 pairs of a thread cost ~1 us each)
            lines.append(f"      pt_u64 b[{FM}];")
            lines.append(f"      bool got[{FM}];")
            lines.append(f"#pragma unroll\n      for (int u = 0; u < {FM}; u++) {{ b[u] = 0; got[u] = threadIdx.x + u * {BLOCK} >= gridDim.x; }}")
            lines.append("      for (long long spins = 0;; spins++) {")
            lines.append("        bool all = true;")
            lines.append(f"#pragma unroll\n        for (int u = 0; u < {FM}; u++)")
            lines.append(f"          if (!got[u]) {{ got[u] = pt_pair_poll((const pt_u64*)part{k} + 2 * (threadIdx.x + u * {BLOCK}), b[u]); all = all && got[u]; }}")
            lines.append("        if (all) break;")
            lines.append("        if (spins > (1ll << 22)) { atomicOr(pt_status, 16); break; }")
            lines.append("      }")
            lines.append(f"#pragma unroll\n      for (int u = 0; u < {FM}; u++)")
            lines.append(f"        if (threadIdx.x + u * {BLOCK} < gridDim.x) {{")
            lines.append(f"          {act} v; __builtin_memcpy(&v, &b[u], sizeof(v));")
            lines.append(f"          fa{k} = {op}::apply(fa{k}, v);")
            lines.append(f"          pt_pair_store((pt_u64*)part{k} + 2 * (threadIdx.x + u * {BLOCK}), 0, 0);  // clean for the next launch")
            lines.append("        }")
            lines.append("    }")
            lines.append("    __syncthreads();")
            lines.append(f"    fa{k} = pthip_dev::block_reduce<{op}, {act}, {BLOCK}>(fa{k}, smem{k});")
            lines.append(f"    if (threadIdx.x == 0) fin{k}[0] = ({CTYPE[finish[k]]})fa{k};")
        lines.append("  }")
    return "\n".join(lines)


MAX_ND = 5


def nd_kernel_source(name: str, body: dict, ndim: int, reduce_spec=None, partial=(), byvalue=()) -> str:
    """General broadcasting loop: collapsed ``ndim``-d index (row-major over the output
    shape), per-operand element strides (0 on broadcast dims); outputs contiguous.

    ``partial``: input positions that arrive as unfinished split-K slabs ``[np][n]`` (contiguous,
    same iteration space as the output): the value is their sum in ascending slab order — the
    order ``splitk_finish_kernel`` uses — folded into this kernel instead of a launch of its own."""
    nin = len(body["in_dtypes"])
    nout = len(body["out_dtypes"])
    reduce_spec = reduce_spec or [None] * nout
    partial = set(partial)
    byvalue = set(byvalue)
    params = ["long long n"]
    params += [f"long long d{j}" for j in range(ndim)]
    for k, dt in enumerate(body["in_dtypes"]):
        if k in byvalue:  # host-known scalar: travels in the argument block
            params.append(f"const long long in{k}")
            continue
        params.append(f"const {CTYPE[dt]}* __restrict__ in{k}")
        if k in partial:
            # np slabs, slab stride ps; rows of pn elements, pld apart (a column block of wider slabs)
            params += [f"long long np{k}", f"long long ps{k}", f"long long pn{k}", f"long long pld{k}"]
        else:
            params += [f"long long s{k}_{j}" for j in range(ndim)]
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None:
            params.append(f"{CTYPE[dt]}* __restrict__ out{k}")
        else:
            params.append(f"{CTYPE[reduce_spec[k][1]]}* __restrict__ part{k}")
    src = [reduce_header() if any(reduce_spec) else "", prelude_for(body)]
    src.append(f'extern "C" __global__ __launch_bounds__({BLOCK}) void {name}({", ".join(params)}) {{')
    for k, rs in enumerate(reduce_spec):
        if rs is not None:
            act = CTYPE[rs[1]]
            src.append(f"  {act} acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::identity<{act}>();")
    for k in sorted(byvalue):
        ct = CTYPE[body["in_dtypes"][k]]
        src.append(f"  {ct} bv{k}; {{ const long long b = in{k}; __builtin_memcpy(&bv{k}, &b, sizeof({ct})); }}")
    src.append(f"  for (long long i = (long long)blockIdx.x * {BLOCK} + threadIdx.x; i < n; i += (long long)gridDim.x * {BLOCK}) {{")
    src.append("      long long rem = i;")
    for j in range(ndim - 1, 0, -1):
        src.append(f"      const long long c{j} = rem % d{j}; rem /= d{j};")
    src.append("      const long long c0 = rem;")
    in_names = []
    for k in range(nin):
        if k in byvalue:
            in_names.append(f"bv{k}")
            continue
        if k in partial:
            ct = CTYPE[body["in_dtypes"][k]]
            # eight independent loads in flight, added in ascending slab order (deterministic)
            src.append(f"      const long long pi{k} = (pld{k} == pn{k}) ? i : (i / pn{k}) * pld{k} + (i % pn{k});")
            src.append(f"      {ct} p{k} = in{k}[pi{k}];")
            src.append(f"      long long sl{k} = 1;")
            src.append(f"      for (; sl{k} + 7 < np{k}; sl{k} += 8) {{")
            src.append(f"        {ct} q{k}[8];")
            src.append(f"#pragma unroll\n        for (int u = 0; u < 8; u++) q{k}[u] = in{k}[(sl{k} + u) * ps{k} + pi{k}];")
            src.append(f"#pragma unroll\n        for (int u = 0; u < 8; u++) p{k} += q{k}[u];")
            src.append("      }")
            src.append(f"      for (; sl{k} < np{k}; sl{k}++) p{k} += in{k}[sl{k} * ps{k} + pi{k}];")
            in_names.append(f"p{k}")
            continue
        off = " + ".join(f"c{j} * s{k}_{j}" for j in range(ndim)) or "0"
        in_names.append(f"in{k}[{off}]")
    out_names = []
    for k, dt in enumerate(body["out_dtypes"]):
        src.append(f"      {CTYPE[dt]} o{k};")
        out_names.append(f"o{k}")
    src.append(emit_body(body, in_names, out_names))
    for k, rs in enumerate(reduce_spec):
        if rs is None:
            src.append(f"      out{k}[i] = o{k};")
        else:
            src.append(f"      acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::apply(acc{k}_0, ({CTYPE[rs[1]]})o{k});")
    src.append("  }")
    src.append(_reduce_epilogue(reduce_spec, 1))
    src.append("}")
    return "\n".join(src)


def gemv_chain_source(name, body, e_modes, reduce_spec, w_out, C, RG, store_r, has_y1,
                      out_store=None, scatter_out=None, scatter_groups=2, pack=2, atype="float64") -> str:
    """One-pass ``r = b1*y1 + a1*A@x ; outs = body(.., r, ..) ; partial += A.T@w`` (fp64).

    Work decomposition (wave64): a wave owns groups of ``RG`` consecutive rows.  Lane l
    holds columns {2l, 2l+1} + 128c (c < C) of every row of the group in registers
    (16-byte coalesced loads: one wave instruction = one 1 KiB row chunk), so the
    matrix is read from HBM exactly once and used twice:

    1. per-row partial dot products (2C FMAs per row per lane);
    2. a *transposing* butterfly: log2(RG) exchange steps in which every lane gives away
       half of its rows (RG-1 exchanges instead of 6 per row), then 6-log2(RG) plain
       steps — lanes (row << s .. ) end up owning one finished row each;
    3. the scalar graph runs once per row on the owning lanes (other row inputs are
       coalesced loads or table gathers), reductions accumulate per lane, vector outputs
       are stored only if something outside the fused node reads them;
    4. w[row] is broadcast back with ``v_readlane`` (compile-time lane) and multiplied
       into the still-resident row registers: acc[c] += row * w;
    5. optionally the scatter-add ``out[sidx[row]] += o[row]`` (gradient of a gather) is
       accumulated in the same pass: lane b owns bins b, b+64, ... (``scatter_groups`` x 64 <= 256 bins), rows are
       visited in order, per-workgroup partials are combined in a fixed order afterwards
       (deterministic, like every other reduction here).

    ``pack`` = 2: lane l holds columns {2l, 2l+1} of every 128-column chunk (one 16-byte load; rows must start on
    16-byte boundaries: even ``lda``, even K).  ``pack`` = 1: columns {l, l+64} (two 8-byte loads, each a coalesced
    512-byte row piece): any K, any ``lda`` — the odd-K instance.  ``C`` > 8 chunks (K > 1024): the multiplier vector
    ``x`` lives in LDS instead of registers, and fewer rows ride per group (``RG`` = 2: K <= 2048, 1: K <= 4096) so that
    the row registers (``RG*C`` <= 32 packs) and the ``A.T@w`` accumulators (``C`` packs) still fit.

    ``atype`` = "float32": the matrix, ``x``, ``y1`` and the stored Gemv result are float arrays — converted on load
    (an 8-byte ``float2`` per lane and chunk with ``pack`` = 2), everything between the loads and the stores stays the
    double-precision kernel (dot products, butterfly, ``A.T@w`` accumulators, partial slabs); the scalar graph gets the
    Gemv result rounded to float, as the reference's float32 ``Gemv`` output would be.

    ``e_modes[k]`` ∈ {'R' the Gemv result, 'V' N-vector, 'S' scalar, 'G' gather
    ``table[gidx[row]]``} per elementwise input.
    Kernel params (all 8 bytes): N, K, A, lda, x, y1, alpha1, beta1, <per elementwise input
    except R: ptr (and for 'G': index ptr, table length)>, [r_out], <per output: stored ptr
    (if stored) | partial ptr (if reduced)>, partT, [sidx, sbins, partS], status.
    """
    import math

    nout = len(body["out_dtypes"])
    out_store = list(out_store) if out_store is not None else [True] * nout
    lg = int(math.log2(RG))
    assert 1 << lg == RG and 1 <= RG <= 32 and RG * C <= 32 and pack in (1, 2, 4)
    assert pack != 4 or (atype == "float32" and C % 2 == 0)
    b_lds = C > 8
    if pack == 4:
        # float32 only: a lane's 16-byte load is FOUR columns {4l .. 4l+3} of a 256-column chunk PAIR; the two halves are
        # chunks c (even) and c + 1 of the double-precision register image.  (8-byte loads — float2 per lane — fetch the
        # same bytes per instruction and run 2.6x slower: the waves sit at s_waitcnt 6x as long, profiles/r6a_gchain_f32_pmc.md)
        col0 = "(c >> 1) * 256 + 4 * lane + 2 * (c & 1)"
        col1 = col0 + " + 1"
    else:
        col0 = "c * 128 + 2 * lane" if pack == 2 else "c * 128 + lane"  # first column of lane's pack in chunk c
        col1 = "c * 128 + 2 * lane + 1" if pack == 2 else "c * 128 + 64 + lane"

    at = CTYPE[atype]

    def ld_pack(base, stream=True):  # the lane's two columns of chunk c from `base` (a pointer to `atype`)
        if pack == 2 and atype == "float64":
            ld = _stream_load(f"(const pt_d2*)({base} + {col0})") if stream else f"*(const pt_d2*)({base} + {col0})"
            return f"(({col0}) < K) ? {ld} : (pt_d2){{0.0, 0.0}}"
        if pack == 2:
            ld = _stream_load(f"(const pt_f2*)({base} + {col0})") if stream else f"*(const pt_f2*)({base} + {col0})"
            return f"(({col0}) < K) ? pt_widen({ld}) : (pt_d2){{0.0, 0.0}}"
        if pack == 4 and not stream:  # (the short multiplier vector: element loads)
            return f"(pt_d2){{(({col0}) < K) ? (double){base}[{col0}] : 0.0, (({col1}) < K) ? (double){base}[{col1}] : 0.0}}"
        assert pack != 4, "the matrix rows of the four-column form are loaded pairwise (below)"
        return f"(pt_d2){{(({col0}) < K) ? (double){base}[{col0}] : 0.0, (({col1}) < K) ? (double){base}[{col1}] : 0.0}}"

    rest = 6 - lg  # plain butterfly steps after the transposing ones
    params = [
        "long long N", "long long K", f"const {CTYPE[atype]}* __restrict__ A", "long long lda",
        f"const {CTYPE[atype]}* __restrict__ x", f"const {CTYPE[atype]}* __restrict__ y1", "double alpha1", "double beta1",
    ]
    for k, m in enumerate(e_modes):
        if m == "R":
            continue
        params.append(f"const {CTYPE[body['in_dtypes'][k]]}* __restrict__ in{k}")
        if m == "G":
            params += [f"const long long* __restrict__ gidx{k}", f"long long glen{k}"]
    if store_r:
        params.append(f"{CTYPE[atype]}* __restrict__ r_out")
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is not None:
            params.append(f"{CTYPE[reduce_spec[k][1]]}* __restrict__ part{k}")
        elif out_store[k]:
            params.append(f"{CTYPE[dt]}* __restrict__ out{k}")
    params.append("double* __restrict__ partT")
    if scatter_out is not None:
        params += ["const long long* __restrict__ sidx", "long long sbins", "double* __restrict__ partS"]
    params.append("int* __restrict__ status")
    L = [reduce_header(), prelude_for(body)]
    L.append("typedef double pt_d2 __attribute__((ext_vector_type(2)));")
    L.append("typedef float pt_f2 __attribute__((ext_vector_type(2)));")
    L.append("typedef float pt_f4 __attribute__((ext_vector_type(4)));")
    L.append("static __device__ __forceinline__ pt_d2 pt_widen(pt_f2 v) { return (pt_d2){(double)v.x, (double)v.y}; }")
    L.append("static __device__ __forceinline__ double pt_shfl_xor(double v, int m) { return pthip_dev::shfl_xor_any(v, m); }")
    L.append("static __device__ __forceinline__ double pt_readlane(double v, int l) {")
    L.append("  union { double d; int i[2]; } u; u.d = v;")
    L.append("  u.i[0] = __builtin_amdgcn_readlane(u.i[0], l); u.i[1] = __builtin_amdgcn_readlane(u.i[1], l); return u.d; }")
    L.append(f'extern "C" __global__ __launch_bounds__({BLOCK}) void {name}({", ".join(params)}) {{')
    L.append(f"  constexpr int C = {C}, RG = {RG};")
    L.append("  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;")
    L.append("  pt_d2 accT[C];")
    red_w = f"(128 * C > {64 * int(scatter_groups)} ? 128 * C : {64 * int(scatter_groups)})" if scatter_out is not None else "128 * C"
    L.append(f"  __shared__ double redT[{BLOCK // 64}][{red_w}];")
    if b_lds:
        # the multiplier vector: read per use (registers hold the rows and the accumulators); it lives in the memory the
        # block combine uses after the row loop
        L.append("  pt_d2 (*b)[64] = reinterpret_cast<pt_d2 (*)[64]>(&redT[0][0]);")
        L.append(f"  for (int j = threadIdx.x; j < C * 64; j += {BLOCK}) {{ const int c = j >> 6, lane = j & 63; b[c][lane] = " + ld_pack("x", stream=False) + "; }")
        L.append("  __syncthreads();")
        L.append("#pragma unroll\n  for (int c = 0; c < C; c++) accT[c] = (pt_d2){0.0, 0.0};")
        bref = "b[c][lane]"
    else:
        L.append("  pt_d2 b[C];")
        L.append("#pragma unroll\n  for (int c = 0; c < C; c++) {")
        L.append("    b[c] = " + ld_pack("x", stream=False) + ";")
        L.append("    accT[c] = (pt_d2){0.0, 0.0};\n  }")
        bref = "b[c]"
    if scatter_out is not None:
        SG = int(scatter_groups)
        assert 1 <= SG <= 4
        L.append("  double " + ", ".join(f"accS{q} = 0.0" for q in range(SG)) + ";  // bins lane, lane + 64, ...")
    for k, m in enumerate(e_modes):
        if m == "S":
            L.append(f"  const {CTYPE[body['in_dtypes'][k]]} s{k} = in{k}[0];")
    for k, rs in enumerate(reduce_spec):
        if rs is not None:
            act = CTYPE[rs[1]]
            L.append(f"  {act} acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::identity<{act}>();")
    L.append(f"  const int myrow = (lane >> {rest}) & (RG - 1);   // row of the group this lane finishes")
    L.append(f"  const bool owner = (lane & {(1 << rest) - 1}) == 0;")
    L.append("  const long long ngroups = (N + RG - 1) / RG;")
    L.append(f"  for (long long g = (long long)blockIdx.x * {BLOCK // 64} + wid; g < ngroups; g += (long long)gridDim.x * {BLOCK // 64}) {{")
    L.append("    const long long row0 = g * RG;")
    L.append("    pt_d2 xr[RG][C];")
    L.append("#pragma unroll\n    for (int r = 0; r < RG; r++) {")
    L.append("      const long long row = (row0 + r < N) ? row0 + r : N - 1;")
    L.append(f"      const {at}* __restrict__ Ar = A + row * lda;")
    if pack == 4:
        L.append("#pragma unroll\n      for (int c = 0; c < C; c += 2) {")
        L.append("        const long long cq = (c >> 1) * 256 + 4 * lane;")
        L.append("        pt_f4 t4 = {0.f, 0.f, 0.f, 0.f};")
        L.append("        if (cq < K) t4 = " + _stream_load("(const pt_f4*)(Ar + cq)") + ";  // (K % 4 == 0: a pack is inside the row or outside)")
        L.append("        xr[r][c] = (pt_d2){(double)t4.x, (double)t4.y};")
        L.append("        xr[r][c + 1] = (pt_d2){(double)t4.z, (double)t4.w};\n      }\n    }")
    else:
        L.append("#pragma unroll\n      for (int c = 0; c < C; c++) {")
        L.append("        xr[r][c] = " + ld_pack("Ar") + ";\n      }\n    }")
    L.append("    double p[RG];")
    L.append("#pragma unroll\n    for (int r = 0; r < RG; r++) {")
    L.append("      double s = 0.0;")
    L.append(f"#pragma unroll\n      for (int c = 0; c < C; c++) {{ const pt_d2 bc = {bref}; s += xr[r][c].x * bc.x + xr[r][c].y * bc.y; }}")
    L.append("      p[r] = s;\n    }")
    half = RG // 2
    mask = 32
    while half >= 1:
        L.append(f"    {{ const bool up = (lane & {mask}) != 0;")
        L.append(f"#pragma unroll\n      for (int i = 0; i < {half}; i++) {{")
        L.append(f"        const double send = up ? p[i] : p[i + {half}];")
        L.append(f"        const double keep = up ? p[i + {half}] : p[i];")
        L.append(f"        p[i] = keep + pt_shfl_xor(send, {mask});\n      }} }}")
        half //= 2
        mask //= 2
    while mask >= 1:
        L.append(f"    p[0] += pt_shfl_xor(p[0], {mask});")
        mask //= 2
    L.append("    const long long row = row0 + myrow;")
    L.append("    const bool valid = row < N;")
    L.append("    const long long rowc = valid ? row : N - 1;")
    L.append("    double res = alpha1 * p[0];")
    if has_y1:
        L.append("    if (beta1 != 0.0) res += beta1 * y1[rowc];")
    if store_r:
        L.append("    if (valid && owner) r_out[row] = res;")
    in_names = []
    for k, m in enumerate(e_modes):
        if m == "R":
            in_names.append("res" if body["in_dtypes"][k] == "float64" else f"(({CTYPE[body['in_dtypes'][k]]})res)")
        elif m == "S":
            in_names.append(f"s{k}")
        elif m == "G":
            L.append(f"    long long gi{k} = gidx{k}[rowc];")
            L.append(f"    if (gi{k} < 0) gi{k} += glen{k};")
            L.append(f"    if (gi{k} < 0 || gi{k} >= glen{k}) {{ atomicOr(status, 1); gi{k} = 0; }}  // IndexError, reported by the host")
            in_names.append(f"in{k}[gi{k}]")
        else:
            in_names.append(f"in{k}[rowc]")
    out_names = []
    for k, dt in enumerate(body["out_dtypes"]):
        L.append(f"    {CTYPE[dt]} o{k};")
        out_names.append(f"o{k}")
    L.append(emit_body(body, in_names, out_names, indent="    "))
    for k, rs in enumerate(reduce_spec):
        if rs is not None:
            L.append(f"    if (valid && owner) acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::apply(acc{k}_0, ({CTYPE[rs[1]]})o{k});")
        elif out_store[k]:
            L.append(f"    if (valid && owner) out{k}[row] = o{k};")
    L.append(f"    const double w = valid ? (double)o{w_out} : 0.0;")
    if scatter_out is not None:
        if scatter_out != w_out:
            L.append(f"    const double sv = valid ? (double)o{scatter_out} : 0.0;")
        L.append("    long long si_ = sidx[rowc];")
        L.append("    if (si_ < 0) si_ += sbins;")
        L.append("    if (valid && (si_ < 0 || si_ >= sbins)) { atomicOr(status, 1); }")
        L.append("    const int si = (valid && si_ >= 0 && si_ < sbins) ? (int)si_ : -1;")
    L.append("#pragma unroll\n    for (int r = 0; r < RG; r++) {")
    L.append(f"      const double wr = pt_readlane(w, r << {rest});")
    L.append("#pragma unroll\n      for (int c = 0; c < C; c++) { accT[c].x += xr[r][c].x * wr; accT[c].y += xr[r][c].y * wr; }")
    if scatter_out is not None:
        sval = "wr" if scatter_out == w_out else f"pt_readlane(sv, r << {rest})"
        L.append(f"      const int ir = __builtin_amdgcn_readlane(si, r << {rest});")
        L.append(f"      const double svr = {sval};")
        for q in range(SG):
            L.append(f"      accS{q} += (ir == lane + {64 * q}) ? svr : 0.0;")
    L.append("    }")
    L.append("  }")
    # block combine of accT (fixed wave order), of the scatter bins and of the reductions
    if b_lds:
        L.append("  __syncthreads();  // every wave is done reading the multiplier vector out of this memory")
    L.append(f"#pragma unroll\n  for (int c = 0; c < C; c++) {{ redT[wid][{col0}] = accT[c].x; redT[wid][{col1}] = accT[c].y; }}")
    L.append("  __syncthreads();")
    L.append(f"  for (int j = threadIdx.x; j < 128 * C; j += {BLOCK}) {{")
    L.append("    double v = redT[0][j];")
    L.append(f"#pragma unroll\n    for (int q = 1; q < {BLOCK // 64}; q++) v += redT[q][j];")
    L.append("    if (j < K) partT[(long long)blockIdx.x * K + j] = v;\n  }")
    if scatter_out is not None:
        L.append("  __syncthreads();")
        L.append("  " + " ".join(f"redT[wid][lane + {64 * q}] = accS{q};" for q in range(SG)))
        L.append("  __syncthreads();")
        L.append(f"  if (threadIdx.x < {64 * SG}) {{")
        L.append("    double v = redT[0][threadIdx.x];")
        L.append(f"#pragma unroll\n    for (int q = 1; q < {BLOCK // 64}; q++) v += redT[q][threadIdx.x];")
        L.append("    if (threadIdx.x < sbins) partS[(long long)blockIdx.x * sbins + threadIdx.x] = v;\n  }")
    L.append(_reduce_epilogue(reduce_spec, 1))
    L.append("}")
    return "\n".join(L)


def source_key(src: str) -> str:
    return hashlib.sha256(src.encode()).hexdigest()[:24]


# ---------------------------------------------------------------------------
# Skinny product + epilogue (gemmfuse.fuse_dot_epilogue): out = body(.., A@B, ..)
# ---------------------------------------------------------------------------

DOTEW_CHUNK = 8  # k-groups (16 k each) per register buffer; two buffers in flight
DOTEW_MAX_K = 16384
_MFMA16 = {"float32": "__builtin_amdgcn_mfma_f32_16x16x4f32", "float64": "__builtin_amdgcn_mfma_f64_16x16x4f64"}


def dot_epilogue_source(name: str, body: dict, dot_pos, K: int, byvalue=(), chunk: int = DOTEW_CHUNK, share=None, lds_a: bool = False, var: str = "",
                        packed_a=(), pack_outs=()) -> str:
    """One 16x16 output tile per workgroup of ``out = body(.., A_d @ B_d, ..)``, full K.

    The recurrent products of a Scan step (``h @ U``: M = batch <= a few hundred rows, K = N =
    hidden) followed by their gate ``Composite``: reference ``Dot22``/``Gemm`` (blas/gemm.py:
    76, 248) + ``Elemwise`` (elemwise.py:755) of one step in ONE launch, no split-K slabs through
    HBM, no finish pass.  MI355X mapping: 256 tiles for (64, 1024) = one per CU; the four waves
    split K, each lane streams its operands with 16-byte loads straight into the MFMA operand
    registers (``v_mfma_*_16x16x4``: lane (i = l%16, q = l/16) supplies A[i][k] and B[k][i] for
    k = 16g + 4q + j, j = 0..3 — one 4-vector load per operand feeds four MFMAs).  ``B`` arrives
    packed by ``pthip_pack_b16`` as ``[N/16][K/4][16][4]`` so that a wave's load is 1 KiB
    contiguous; A is row-major (16 rows x 64 B per instruction).  The wave partials are added in
    wave order through LDS (deterministic), then thread t owns element (t/16, t%16) of the tile
    and runs the scalar graph; operands of the epilogue are requested before the K loop.

    ``share``: ``{follower dot position: leader dot position}`` — products with the SAME left
    operand (``h @ U_r`` and ``h @ U_z``): the leader's A registers feed both MFMA chains, the
    follower loads only its packed B (64 KB less per tile, and two independent accumulator chains).

    ``lds_a`` (float32, K % 128 == 0): the left operand is fetched in full 128-byte lines by
    LDS-DMA (``global_load_lds`` x4: per wave and pair of k-groups two 1 KiB copies, lane = (row l/8,
    16-byte piece l%8 XOR row&7 on the source side so that the linear LDS image is bank-swizzled) and
    the MFMA fragments are read back with ``ds_read_b128`` — instead of fragment-shaped loads (16 rows x
    64 B per instruction), which the texture addresser serves at half rate.  Wave-local: no barrier.

    ``var``: ``"acc4"`` (the default of dispatch/dotew.py) / ``"acc2"`` split every product's
    accumulator into 4 / 2 chains (k-groups round-robin; two chains each when two products share
    their left operand), added in a fixed order at the end.  Measured: no time (the kernels wait on
    memory, profiles/r3a_dotew_variants.txt) but accuracy — an MFMA chain is a k-ordered fma chain,
    and 1000 GRU steps of 256-term chains drifted 1.3x further from the fp64 trajectory than
    OpenBLAS's blocked sums; shorter chains close most of that (tests/test_gpu_fullsize.py).
    The other values are TIMING-ONLY decompositions (wrong results; tools/dotew_variants.py):
    ``nomfma`` (VALU stand-ins for the MFMAs), ``noload`` (operands from a kernel argument),
    ``apacked`` (the left operand fetched with the packed operand's 1-KiB-contiguous pattern).

    ``packed_a``: dot positions whose LEFT operand arrives in the MFMA operand order as well
    (``Ap[M/16][K/4][16][4]``, ``Ap[rt][k4][i][j] = A[16 rt + i][4 k4 + j]``): a wave's load is then
    1 KiB contiguous like the packed right operand, instead of 16 row segments of 64 B (measured
    with the timing-only ``apacked`` variant: -1.1 us per GRU step, profiles/r3a_dotew_variants.txt).
    ``pack_outs``: outputs this kernel ALSO stores in that order (one extra pointer each, after the
    regular outputs) because a later step kernel multiplies them from the left: the 16x16 tile a
    workgroup owns is one contiguous 1 KiB piece of the packed image.  Needs N % 16 == 0.

    Arguments: M, N, then per body input — dot: (A, lda, Bp) | by value: bits | other:
    (ptr, stride0, stride1) — then per output (ptr, row stride), then per packed output its pointer."""
    dot_pos = list(dot_pos)
    packed_a = set(packed_a)
    pack_outs = list(pack_outs)
    nacc = 4 if "acc4" in var else 2 if "acc2" in var else 1
    byvalue = set(byvalue)
    T = body["in_dtypes"][dot_pos[0]]
    assert T in _MFMA16 and all(body["in_dtypes"][p] == T for p in dot_pos)
    assert K % 16 == 0 and 0 < K <= DOTEW_MAX_K
    ct = CTYPE[T]
    G = K // 16
    GW = (G + 3) // 4
    guard = G % 4 != 0
    nd = len(dot_pos)
    P = ["long long M", "long long N"]
    for k, dt in enumerate(body["in_dtypes"]):
        if k in dot_pos:
            P += [f"const {ct}* __restrict__ A{k}", f"long long lda{k}", f"const {ct}* __restrict__ Bp{k}"]
        elif k in byvalue:
            P.append(f"const long long in{k}")
        else:
            P += [f"const {CTYPE[dt]}* __restrict__ in{k}", f"long long s{k}_0", f"long long s{k}_1"]
    for k, dt in enumerate(body["out_dtypes"]):
        P += [f"{CTYPE[dt]}* __restrict__ out{k}", f"long long ldo{k}"]
    for k in pack_outs:
        P.append(f"{CTYPE[body['out_dtypes'][k]]}* __restrict__ pk{k}")
    L = [prelude_for(body)]
    L.append(f"typedef {ct} __attribute__((ext_vector_type(4))) dvec4;")
    L.append(f'extern "C" __global__ __launch_bounds__({BLOCK}) void {name}({", ".join(P)}) {{')
    lds_a = bool(lds_a) and T == "float32" and K % 128 == 0 and chunk % 2 == 0
    L.append(f"  __shared__ {ct} red_[{nd}][4][256];")
    if lds_a:
        # per wave: two buffers of `chunk` k-groups = chunk/2 line pairs of 16 rows x 128 B
        L.append(f"  __shared__ __attribute__((aligned(16))) float lda_[4][2][{chunk // 2}][16 * 32];")
    L.append("  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;")
    L.append("  const long long ctile = blockIdx.x, r0 = (long long)blockIdx.y * 16;")
    L.append("  const long long er = r0 + (tid >> 4), ec = ctile * 16 + (tid & 15);")
    L.append("  const bool live = er < M && ec < N;")
    L.append("  const long long err = live ? er : 0, ecc = live ? ec : 0;")
    in_names, early, ew_loads = [], [], []
    for k, dt in enumerate(body["in_dtypes"]):
        if k in dot_pos:
            in_names.append(f"dot{k}")
        elif k in byvalue:
            c = CTYPE[dt]
            L.append(f"  {c} bv{k}; {{ const long long b = in{k}; __builtin_memcpy(&bv{k}, &b, sizeof({c})); }}")
            in_names.append(f"bv{k}")
        else:
            ew_loads.append(f"  {CTYPE[dt]} e{k} = in{k}[err * s{k}_0 + ecc * s{k}_1];")
            in_names.append(f"e{k}")
            if dt in ("float32", "float64", "int32", "int64", "uint32", "uint64"):
                early.append(f"e{k}")
    # (the machine scheduler otherwise sinks every load next to its use: 48 VGPRs, one load in
    #  flight per MFMA group, and the epilogue operands requested after the barrier)
    SB = "  __builtin_amdgcn_sched_barrier(0);"
    L.append("  long long arow = r0 + li; if (arow >= M) arow = M - 1;")
    if "noload" in var:
        L.append(f"  const {ct} fake_ = ({ct})M;")
    for p in dot_pos:
        L.append(f"  dvec4 acc{p} = {{0, 0, 0, 0}};")
        for a in range(1, nacc):
            L.append(f"  dvec4 acc{p}_{a} = {{0, 0, 0, 0}};")
        if "apacked" in var or p in packed_a:
            L.append(f"  const dvec4* ap{p} = (const dvec4*)A{p} + (((long long)blockIdx.y * {K // 4} + (long long)wave * {GW * 4} + kq) * 16 + li);")
        else:
            L.append(f"  const dvec4* ap{p} = (const dvec4*)(A{p} + arow * lda{p}) + ((long long)wave * {GW * 4} + kq);")
        L.append(f"  const dvec4* bp{p} = (const dvec4*)Bp{p} + ((ctile * {K // 4} + (long long)wave * {GW * 4} + kq) * 16 + li);")
    # the stream of (dot group, chunk) register buffers, double-buffered; a group = a leader and
    # the dots that share its left operand (at most one follower: register budget)
    share = dict(share or {})
    groups = []
    for p in dot_pos:
        if p in share:
            continue
        fol = [f for f in dot_pos if share.get(f) == p][:1]
        for f in [f for f in dot_pos if share.get(f) == p][1:]:
            share.pop(f)  # further followers stream their own copy of A
        groups.append([p] + fol)
    groups += [[p] for p in dot_pos if p not in {q for g in groups for q in g}]
    chunks = []
    for gr in groups:
        for c0 in range(0, GW, chunk):
            chunks.append((gr, c0, min(chunk, GW - c0)))
    L.append(f"  dvec4 ra_[2][{chunk}], rb_[2][{chunk}];")
    if any(len(gr) > 1 for gr in groups):
        L.append(f"  dvec4 rc_[2][{chunk}];")
    bufs = ["rb_", "rc_"]

    def loads(s):
        gr, c0, n = chunks[s]
        out = []
        for u in range(n):
            g = c0 + u
            if "ntb" in var:  # (timing experiment: non-temporal weight loads)
                bl = " ".join(f"{bufs[q]}[{s & 1}][{u}] = __builtin_nontemporal_load(bp{p} + {g * 64});" for q, p in enumerate(gr))
            else:
                bl = " ".join(f"{bufs[q]}[{s & 1}][{u}] = bp{p}[{g * 64}];" for q, p in enumerate(gr))
            if lds_a:
                la = bl
                if u % 2 == 0:
                    # the 128-byte lines of k-groups g, g+1 of this wave's slice: rows 0-7, then 8-15
                    for hr in (0, 1):
                        la += (f" __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ag{gr[0]}_{hr} + {g * 16}), "
                               f"(__attribute__((address_space(3))) void*)&lda_[wave][{s & 1}][{u // 2}][{hr * 256}], 16, 0, 0);")
            else:
                la = f"ra_[{s & 1}][{u}] = ap{gr[0]}[{g * (64 if ('apacked' in var or gr[0] in packed_a) else 4)}]; " + bl
                if "noload" in var:
                    la = (f"ra_[{s & 1}][{u}] = dvec4{{fake_, fake_, fake_, fake_}}; "
                          + " ".join(f"{bufs[q]}[{s & 1}][{u}] = dvec4{{fake_, fake_, fake_, fake_}};" for q in range(len(gr))))
            if guard:
                zero = f"ra_[{s & 1}][{u}] = dvec4{{0, 0, 0, 0}}; " + " ".join(f"{bufs[q]}[{s & 1}][{u}] = dvec4{{0, 0, 0, 0}};" for q in range(len(gr)))
                la = f"if (wave * {GW} + {g} < {G}) {{ {la} }} else {{ {zero} }}"
            out.append("  " + la)
        return out

    def frags(s):
        """LDS path: the DMAs of chunk s have landed (vmcnt(0)); read its MFMA fragments"""
        gr, c0, n = chunks[s]
        out = ['  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");']
        for u in range(n):
            out.append(f"  ra_[{s & 1}][{u}] = *(const dvec4*)&lda_[wave][{s & 1}][{u // 2}][li * 32 + ((({4 * (u % 2)} + kq) ^ (li & 7)) << 2)];")
        return out

    def mfmas(s):
        gr, c0, n = chunks[s]
        out = []
        for u in range(n):
            for j in range(4):
                for q, p in enumerate(gr):
                    a = u % (nacc if len(gr) == 1 else max(nacc // 2, 1))
                    acc = f"acc{p}_{a}" if a else f"acc{p}"
                    if "nomfma" in var:
                        out.append(f"  acc{p}[{j}] += ra_[{s & 1}][{u}][{j}] * {bufs[q]}[{s & 1}][{u}][{j}];")
                    else:
                        out.append(f"  {acc} = {_MFMA16[T]}(ra_[{s & 1}][{u}][{j}], {bufs[q]}[{s & 1}][{u}][{j}], {acc}, 0, 0, 0);")
        return out

    # issue order: operand chunks 0 and 1, then the epilogue operands (vmcnt retires in order:
    # requested first, a load from HBM would hold up the first MFMA group), then the MFMA stream
    if lds_a:
        # source addresses of the two DMAs per line pair: lane l -> row l/8 (+8), piece (l%8) ^ (row&7)
        L.append("  const int lr_ = lane >> 3, lc_ = lane & 7;")
        for p in sorted({gr[0] for gr, _, _ in chunks}):
            for hr in (0, 1):
                L.append(f"  long long arow{p}_{hr} = r0 + lr_ + {8 * hr}; if (arow{p}_{hr} >= M) arow{p}_{hr} = M - 1;")
                L.append(f"  const {ct}* ag{p}_{hr} = A{p} + arow{p}_{hr} * lda{p} + (long long)wave * {GW * 16} + ((lc_ ^ (lr_ & 7)) << 2);")
        assert not guard
        # (the DMA wait is a plain vmcnt(0): the epilogue operands are requested first so that it
        #  never waits for anything younger than the chunk it needs)
        L += ew_loads + [SB] + loads(0) + [SB]
        for s in range(len(chunks)):
            L += frags(s) + [SB]
            if s + 1 < len(chunks):
                L += loads(s + 1) + [SB]
            L += mfmas(s) + [SB]
    else:
        L += loads(0) + [SB]
        for s in range(len(chunks)):
            if s + 1 < len(chunks):
                L += loads(s + 1) + [SB]
            if s == 0:
                L += ew_loads + [SB]
            L += mfmas(s) + [SB]
    # pin the epilogue operands here: without a use in this block the whole scalar graph, loads
    # included, is sunk into `if (live)` behind the barrier
    for e in early:
        L.append(f'  asm volatile("" : "+v"({e}));')
    if nacc > 1:
        for p in dot_pos:
            # fixed order: ((a0 + a1) + (a2 + a3)); chains a product never used stay zero
            if nacc == 4:
                L.append(f"  acc{p} = (acc{p} + acc{p}_1) + (acc{p}_2 + acc{p}_3);")
            else:
                L.append(f"  acc{p} += acc{p}_1;")
    for d, p in enumerate(dot_pos):
        # accumulator register v of lane (li, kq): f32 16x16x4 -> row 4*kq + v; f64 -> row kq + 4*v
        row = "4 * kq + v" if T == "float32" else "kq + 4 * v"
        L.append(f"#pragma unroll\n  for (int v = 0; v < 4; v++) red_[{d}][wave][({row}) * 16 + li] = acc{p}[v];")
    L.append("  __syncthreads();")
    for d, p in enumerate(dot_pos):
        L.append(f"  const {ct} dot{p} = ((red_[{d}][0][tid] + red_[{d}][1][tid]) + red_[{d}][2][tid]) + red_[{d}][3][tid];")
    out_names = []
    for k, dt in enumerate(body["out_dtypes"]):
        L.append(f"  {CTYPE[dt]} o{k};")
        out_names.append(f"o{k}")
    L.append(emit_body(body, in_names, out_names, indent="  "))
    L.append("  if (live) {")
    for k in range(len(body["out_dtypes"])):
        L.append(f"    out{k}[er * ldo{k} + ec] = o{k};")
    L.append("  }")
    for k in pack_outs:
        # element (i = tid/16, column ec) of row tile blockIdx.y: k4 = ec/4, j = ec%4 (all 256 threads: rows
        # past M hold zeros, so that a consumer's MFMA never multiplies uninitialised memory)
        L.append(f"  pk{k}[((blockIdx.y * (N >> 2) + (ctile * 4 + ((tid & 15) >> 2))) * 16 + (tid >> 4)) * 4 + (tid & 3)] = live ? o{k} : ({CTYPE[body['out_dtypes'][k]]})0;")
    L.append("}")
    return "\n".join(L)


# ---------------------------------------------------------------------------
# Tail kernel: a chain of small nodes in ONE single-workgroup launch (tailfuse.py)
# ---------------------------------------------------------------------------

TAIL_BLOCK = 256


_tail_header_cache = None


def tail_header() -> str:
    global _tail_header_cache
    if _tail_header_cache is None:
        src = open(os.path.join(_HERE, "csrc", "tail_device.h")).read()
        _tail_header_cache = src.replace("#pragma once", "")
    return _tail_header_cache


TAIL_SHRINK_MAX_TASKS = 4  # the by-value task table of the fused form (csrc/tail_device.h TailTasksT<4>, 192 bytes)


def tail_shrink_pack(tasks):
    """``TailTasksT<4>`` as kernel-argument bytes: ``tasks`` = [(op code, part ptr, nparts, M, S, out ptr)];
    returns (bytes, number of blocks)."""
    import struct

    n = len(tasks)
    assert 1 <= n <= TAIL_SHRINK_MAX_TASKS
    pad = TAIL_SHRINK_MAX_TASKS - n
    blk0, nb = [], 0
    for _, _, _, M, S, _ in tasks:
        blk0.append(nb)
        nb += (int(M) + 15) // 16 * int(S)
    blk0 += [nb] * (pad + 1)
    col = lambda k, fill=0: [int(t[k]) for t in tasks] + [fill] * pad
    buf = struct.pack("<i4i4x4Q4q4q4i4Q5i4x", n, *col(0), *col(1), *col(2), *col(3, 1), *col(4, 1), *col(5), *blk0)
    assert len(buf) == 192
    return buf, nb


def _tail_prologue(L, shrink):
    """``shrink`` = {"dtype": accumulator dtype}: the launch has one workgroup per slab piece; each shrinks its
    piece (csrc/tail_device.h, the code of pthip_multi_finish), takes a ticket, and only the LAST one to finish
    goes on to the chain (release: fence before the ticket; acquire: fence after it) and puts the ticket back."""
    # device-side join of a segmented plan's two streams (csrc/tail_device.h; include/pthip.h pthip_join_signal): wait
    # for the other stream's signal word and put it back.  Null outside such a plan.  Where kernels of different streams
    # cannot overlap (a counter-collecting profiler serialises them) the signal launch cannot run while this one spins:
    # the wait gives up after 1 ms and says so through the done word.
    L.append("  __shared__ int join_fail_;")
    L.append("  if (join_src != nullptr) {")
    L.append("    if (tid == 0) {")
    L.append("      const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();")
    L.append("      // 1 ms when the host can run this segment again (it polls done_dst), else 3 s and the status bit")
    L.append("      const unsigned long long lim_ = done_dst != nullptr ? 100000ull : 300000000ull;")
    L.append("      int ok_ = 1;")
    L.append("      while (__hip_atomic_load(join_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {")
    L.append("        __builtin_amdgcn_s_sleep(2);")
    L.append("        if (__builtin_amdgcn_s_memrealtime() - t0_ > lim_) { ok_ = 0; break; }")
    L.append("      }")
    L.append("      if (ok_) __hip_atomic_store(join_src, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);")
    L.append("      else if (done_dst != nullptr) __hip_atomic_store(done_dst, 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);")
    L.append("      else if (status_src != nullptr) atomicOr((int*)status_src, 16);")
    L.append("      join_fail_ = !ok_ && done_dst != nullptr;")
    L.append("    }")
    if os.environ.get("PTHIP_JOIN_FENCE", "0") == "1":
        # opt-in: the formally ordered form — an agent-scope acquire behind the wait.  It invalidates this XCD's L2, so
        # the slab this launch shrinks next comes back from HBM (4.8 -> 11 us, profiles/r4_c4_device_join.txt): the
        # default leans on the invalidate every kernel start performs instead, checked by pthip_join_probe per process.
        L.append("    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\");")
    L.append("    __syncthreads();  // (default: no acquire fence — see csrc/tail_device.h plan_join_wait; PTHIP_JOIN_FENCE=1 adds one)")
    L.append("    if (join_fail_) return;  // done word 2: pthip_plan_replay4 waits for the other stream and runs this segment again")
    L.append("  }")
    if not shrink:
        return
    ct = CTYPE[shrink["dtype"]]
    L.append("  {  // prologue: the partial slabs shrink in THIS launch; the last workgroup to finish runs the chain")
    L.append(f"    __shared__ {ct} shr_[{TAIL_BLOCK}];")
    L.append("    __shared__ int last_;")
    L.append(f"    pthip_dev::tail_shrink_block<{ct}, {TAIL_SHRINK_MAX_TASKS}>(tasks_, (int)blockIdx.x, shr_);")
    L.append("    __threadfence();")
    L.append("    __syncthreads();")
    L.append("    if (tid == 0) last_ = atomicAdd(ticket_, 1) == (int)gridDim.x - 1;")
    L.append("    __syncthreads();")
    L.append("    if (!last_) return;")
    L.append("    if (tid == 0) __hip_atomic_store(ticket_, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);")
    L.append("    __threadfence();")
    L.append("  }")


def _tail_epilogue(L, spec):
    for k, o in enumerate(spec["outs"]):
        L.append(f"  for (long long i = tid; i < len{k}; i += {TAIL_BLOCK}) dst{k}[i] = l{o}[i];")
    L.append("  if (tid == 0 && status_dst != nullptr) *status_dst = __hip_atomic_load(status_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);")
    # the plan's completion word (pinned host memory, polled by pthip_plan_replay4): behind every result store
    L.append("  if (done_dst != nullptr) {")
    L.append("    __threadfence_system();")
    L.append("    __syncthreads();")
    L.append("    if (tid == 0) __hip_atomic_store(done_dst, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);")
    L.append("  }")
    L.append("}")


def tail_chain_source(name: str, spec: dict, sizes: dict | None = None, shrink: dict | None = None) -> str:
    """One workgroup runs ``spec["steps"]`` in order, intermediates in LDS.

    ``spec`` is purely structural (extents are kernel arguments, so one code object serves every
    size):

    * ``ext``   — external operands: ``{"kind": "V" (vector: pointer + element stride) | "S"
      (device scalar) | "C" (host scalar by value) | "P" (row-major partial slab / partials),
      "dtype"}``;
    * ``slots`` — LDS values ``{"dtype"}`` (byte offset and length arrive as arguments);
    * ``steps`` — ``finish``: ``out[i] = beta*y[i] + alpha*sum_s src[s*M+i]`` (the second stage +
      epilogue of a split Gemv / scatter-add, blas/gemv.py:64-108; ``y`` optional),
      ``rsum``: a deferred full reduction over ``rows`` partials (elemwise.py:1233 second stage),
      ``ew``: an ``Elemwise`` / ``ElemwiseReduce`` over ``n`` elements with per-operand modes;
    * ``outs``  — LDS slots copied to their destinations at the end; optional status-word copy.

    Operand references are ``("e", k)`` (external) or ("l", k) (LDS slot).  Argument order =
    the order of ``tail_chain_args``.
    """
    ext, slots, steps = spec["ext"], spec["slots"], spec["steps"]
    P = []
    for k, e in enumerate(ext):
        ct = CTYPE[e["dtype"]]
        if e["kind"] == "C":
            P.append(f"const long long ec{k}")
        elif e["kind"] == "V":
            P += [f"const {ct}* __restrict__ e{k}", f"const long long es{k}"]
        else:
            P.append(f"const {ct}* __restrict__ e{k}")
    # one-element slots sit at static offsets (16 B apart, first in the LDS block): a wide graph has
    # hundreds of them and the kernel-argument block is 4 KB; vector slots get run-time offsets
    for k in range(len(slots)):
        if not slots[k].get("scalar"):
            P += [f"const long long off{k}"]
    for j, st in enumerate(steps):
        if st["op"] == "finish":
            P += [f"const long long rows{j}", f"const long long M{j}", f"const double alpha{j}", f"const double beta{j}"]
        elif st["op"] == "rsum":
            P += [f"const long long rows{j}"]
        else:
            P += [f"const long long n{j}"]
    for k, o in enumerate(spec["outs"]):
        P += [f"{CTYPE[slots[o]['dtype']]}* __restrict__ dst{k}", f"const long long len{k}"]
    P += ["const int* status_src", "int* status_dst", "int* done_dst", "int* join_src"]
    if shrink:
        P += [f"const pthip_dev::TailTasksT<{TAIL_SHRINK_MAX_TASKS}> tasks_", "int* ticket_"]
    bodies = [st["body"] for st in steps if st["op"] == "ew"]
    if sizes is not None:
        return _tail_preload_source(name, spec, sizes, P, bodies, shrink)
    L = [reduce_header(), tail_header() if shrink else "", prelude_for(*bodies)]
    L.append(f'extern "C" __global__ __launch_bounds__({TAIL_BLOCK}) void {name}({", ".join(P)}) {{')
    L.append("  extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];")
    L.append("  __shared__ double red_[8];")
    L.append("  const int tid = threadIdx.x;")
    _tail_prologue(L, shrink)
    for k, e in enumerate(ext):
        if e["kind"] == "C":
            ct = CTYPE[e["dtype"]]
            L.append(f"  {ct} c{k}; {{ const long long b = ec{k}; __builtin_memcpy(&c{k}, &b, sizeof({ct})); }}")
    n_sc = 0
    for k, s in enumerate(slots):
        ct = CTYPE[s["dtype"]]
        if s.get("scalar"):
            L.append(f"  {ct}* const l{k} = ({ct}*)(lds_ + {16 * n_sc});")
            n_sc += 1
        else:
            L.append(f"  {ct}* const l{k} = ({ct}*)(lds_ + off{k});")

    def operand(ref, mode, i="i"):
        kind, k = ref
        if kind == "e":
            e = ext[k]
            if e["kind"] == "C":
                return f"c{k}"
            if e["kind"] == "V":
                return f"e{k}[{i} * es{k}]" if mode == "V" else f"e{k}[0]"
            return f"e{k}[0]"
        return f"l{k}[{i}]" if mode == "V" else f"l{k}[0]"

    for j, st in enumerate(steps):
        if st["op"] == "finish":
            ct = CTYPE[st["dtype"]]
            src = f"e{st['src'][1]}"
            L.append(f"  // step {j}: second stage + epilogue of a split Gemv / scatter-add")
            L.append(f"  for (long long i = tid; i < M{j}; i += {TAIL_BLOCK}) {{")
            L.append(f"    {ct} a0 = 0, a1 = 0;")
            L.append(f"    long long s = 0;")
            L.append(f"    for (; s + 1 < rows{j}; s += 2) {{ a0 += {src}[s * M{j} + i]; a1 += {src}[(s + 1) * M{j} + i]; }}")
            L.append(f"    if (s < rows{j}) a0 += {src}[s * M{j} + i];")
            L.append(f"    {ct} r = ({ct})alpha{j} * (a0 + a1);")
            if st.get("y") is not None:
                L.append(f"    if (beta{j} != 0.0) r += ({ct})beta{j} * ({ct}){operand(st['y'], st['ymode'])};")
            L.append(f"    l{st['out']}[i] = r;")
            L.append("  }")
            L.append("  __syncthreads();")
        elif st["op"] == "rsum":
            act, oct_ = CTYPE[st["acc_dtype"]], CTYPE[st["dtype"]]
            op = REDUCE_OPS[st["red"]]
            src = f"e{st['src'][1]}"
            L.append(f"  // step {j}: deferred second stage of a fused Elemwise+reduce kernel")
            L.append("  {")
            L.append(f"    {act} a = pthip_dev::{op}::identity<{act}>();")
            L.append(f"    for (long long p = tid; p < rows{j}; p += {TAIL_BLOCK}) a = pthip_dev::{op}::apply(a, ({act}){src}[p]);")
            L.append(f"    a = pthip_dev::block_reduce<pthip_dev::{op}, {act}, {TAIL_BLOCK}, true>(a, ({act}*)red_);")
            L.append(f"    if (tid == 0) l{st['out']}[0] = ({oct_})a;")
            L.append("  }")
            L.append("  __syncthreads();")
        elif st["op"] == "scatter":
            ct = CTYPE[st["dtype"]]
            L.append(f"  // step {j}: a chain of one-element IncSubtensor updates (widefuse.collect_scalar_updates)")
            L.append(f"  for (long long i = tid; i < n{j}; i += {TAIL_BLOCK}) l{st['out']}[i] = ({ct}){operand(st['base'], st['bmode'])};")
            L.append("  __syncthreads();")
            L.append("  if (tid == 0) {")
            for k, is_set, y in zip(st["indices"], st["set"], st["ys"]):
                L.append(f"    l{st['out']}[{k}] {'=' if is_set else '+='} ({ct}){operand(y, 'S')};")
            L.append("  }")
            L.append("  __syncthreads();")
        else:
            body, modes, red = st["body"], st["modes"], st["reduce"]
            L.append(f"  // step {j}: Elemwise over n{j} elements, operand modes {modes}")
            L.append("  {")
            for q, r in enumerate(red):
                if r is not None:
                    act = CTYPE[r[1]]
                    L.append(f"    {act} acc{q} = pthip_dev::{REDUCE_OPS[r[0]]}::identity<{act}>();")
            L.append(f"    for (long long i = tid; i < n{j}; i += {TAIL_BLOCK}) {{")
            in_names = [operand(ref, m) for ref, m in zip(st["ins"], modes)]
            out_names = []
            for q, dt in enumerate(body["out_dtypes"]):
                L.append(f"      {CTYPE[dt]} o{q};")
                out_names.append(f"o{q}")
            L.append(emit_body(body, in_names, out_names, indent="      "))
            for q, r in enumerate(red):
                if r is None:
                    L.append(f"      l{st['outs'][q]}[i] = o{q};")
                else:
                    L.append(f"      acc{q} = pthip_dev::{REDUCE_OPS[r[0]]}::apply(acc{q}, ({CTYPE[r[1]]})o{q});")
            L.append("    }")
            for q, r in enumerate(red):
                if r is not None:
                    act = CTYPE[r[1]]
                    L.append(f"    acc{q} = pthip_dev::block_reduce<pthip_dev::{REDUCE_OPS[r[0]]}, {act}, {TAIL_BLOCK}, true>(acc{q}, ({act}*)red_);")
                    L.append(f"    if (tid == 0) l{st['outs'][q]}[0] = ({CTYPE[r[2]]})acc{q};")
            L.append("  }")
            L.append("  __syncthreads();")
    _tail_epilogue(L, spec)
    return "\n".join(L)


TAIL_PRELOAD_MAX_REGS = 160  # 8-byte values a thread may hold in flight in the preloading form
TAIL_WAVE_FOLD_INTERLEAVED = os.environ.get("PTHIP_TAIL_FOLD_INTERLEAVED", "1") != "0"
TAIL_WAVE_Q = int(os.environ.get("PTHIP_TAIL_WAVE_Q", 4))  # a wave folds deferred reductions of up to 64 * this many partials
TAIL_SCALAR_STEPS_BY_WAVE = os.environ.get("PTHIP_TAIL_SCALAR_BY_WAVE", "1") != "0"


def tail_preload_sizes(spec: dict, ext_len, step_n):
    """Static size classes for ``tail_chain_source(..., sizes=)`` or ``None`` when the operands
    are too long to sit in registers.  ``ext_len[k]``: elements of external ``k`` ("V"), rows of
    a partial array; ``step_n[j]``: ``(rows, M)`` of a finish step, ``rows`` of an rsum step,
    ``n`` of an elementwise step."""
    ext, steps = spec["ext"], spec["steps"]
    cl = lambda n: max(1, (int(n) + TAIL_BLOCK - 1) // TAIL_BLOCK)
    eu = [cl(ext_len[k]) if e["kind"] == "V" else 0 for k, e in enumerate(ext)]
    su, regs = [], sum(eu) + sum(1 for e in ext if e["kind"] in ("S", "V"))
    for st, n in zip(steps, step_n):
        if st["op"] == "finish":
            rows, M = n
            R = (int(rows) + 3) // 4 * 4
            su.append((R, cl(M)))
            regs += R * cl(M)
        elif st["op"] == "rsum":
            su.append(cl(n))
            regs += cl(n) if n > 64 * TAIL_WAVE_Q else 0  # (few partials: values in ONE wave's lanes, see "wave")
        else:
            su.append(cl(n))
    if regs > TAIL_PRELOAD_MAX_REGS or any((u[0] > 64 or u[1] > 4) if isinstance(u, tuple) else u > 16 for u in su) or any(u > 16 for u in eu):
        return None
    # deferred reductions over <= 64 * TAIL_WAVE_Q partials are folded by single waves, four at a time (a lane adds its
    # up to TAIL_WAVE_Q values in index order first)
    wave = [st["op"] == "rsum" and int(n) <= 64 * TAIL_WAVE_Q for st, n in zip(steps, step_n)]
    wave_q = [max(1, (int(n) + 63) // 64) if w else 0 for w, n in zip(wave, step_n)]
    return {"ext_u": eu, "step_u": su, "wave": wave, "wave_q": wave_q}


def _tail_preload_source(name: str, spec: dict, sizes: dict, P, bodies, shrink=None) -> str:
    """The chain with every *global* operand requested up front (one memory latency for the whole
    kernel instead of one per step — a single workgroup cannot hide it with occupancy): partial
    slabs and partial arrays are summed in registers as they arrive, vectors stay in registers;
    the steps then run out of registers and LDS.  Loop trip counts are static (``sizes``)."""
    ext, slots, steps = spec["ext"], spec["slots"], spec["steps"]
    eu, su = sizes["ext_u"], sizes["step_u"]
    L = [reduce_header(), tail_header() if shrink else "", prelude_for(*bodies)]
    L.append(f'extern "C" __global__ __launch_bounds__({TAIL_BLOCK}) void {name}({", ".join(P)}) {{')
    L.append("  extern __shared__ __attribute__((aligned(16))) unsigned char lds_[];")
    L.append("  __shared__ double red_[8];")
    L.append("  const int tid = threadIdx.x;")
    _tail_prologue(L, shrink)  # (before the operand requests below: the shrunk slabs are among them)
    for k, e in enumerate(ext):
        ct = CTYPE[e["dtype"]]
        if e["kind"] == "C":
            L.append(f"  {ct} c{k}; {{ const long long b = ec{k}; __builtin_memcpy(&c{k}, &b, sizeof({ct})); }}")
    n_sc = 0
    for k, s in enumerate(slots):
        ct = CTYPE[s["dtype"]]
        if s.get("scalar"):
            L.append(f"  {ct}* const l{k} = ({ct}*)(lds_ + {16 * n_sc});")
            n_sc += 1
        else:
            L.append(f"  {ct}* const l{k} = ({ct}*)(lds_ + off{k});")
    # ---- phase 0: every global operand in flight ------------------------------------------
    used_len = {}  # V external -> name of its length (first elementwise step reading it as a vector)
    for j, st in enumerate(steps):
        if st["op"] == "ew":
            for ref, m in zip(st["ins"], st["modes"]):
                if ref[0] == "e" and ext[ref[1]]["kind"] == "V" and m == "V":
                    used_len.setdefault(ref[1], f"n{j}")
        elif st["op"] == "finish" and st.get("y") is not None and st["y"][0] == "e" and st["ymode"] == "V":
            used_len.setdefault(st["y"][1], f"M{j}")
        elif st["op"] == "scatter" and st["base"][0] == "e" and st["bmode"] == "V" and ext[st["base"][1]]["kind"] == "V":
            used_len.setdefault(st["base"][1], f"n{j}")
    for k, e in enumerate(ext):
        ct = CTYPE[e["dtype"]]
        if e["kind"] == "S":
            L.append(f"  const {ct} s{k} = e{k}[0];")
        elif e["kind"] == "V":
            L.append(f"  const {ct} s{k} = e{k}[0];")
            if k in used_len:
                for u in range(eu[k]):
                    # clamped, unconditional (a predicated load becomes an exec-masked branch with its own wait)
                    L.append(f"  const {ct} v{k}_{u} = e{k}[(tid + {u * TAIL_BLOCK} < {used_len[k]} ? (long long)(tid + {u * TAIL_BLOCK}) : {used_len[k]} - 1) * es{k}];")
    wave = sizes.get("wave") or [False] * len(steps)
    for j, st in enumerate(steps):
        if st["op"] == "finish":
            ct = CTYPE[st["dtype"]]
            src = f"e{st['src'][1]}"
            R, U = su[j]
            for u in range(U):
                for r in range(R):
                    L.append(f"  {ct} f{j}_{u}_{r} = {src}[({r} < rows{j} ? {r} : rows{j} - 1) * M{j} + (tid + {u * TAIL_BLOCK} < M{j} ? tid + {u * TAIL_BLOCK} : M{j} - 1)];")
        elif st["op"] == "rsum" and not wave[j]:
            act = CTYPE[st["acc_dtype"]]
            op = REDUCE_OPS[st["red"]]
            src = f"e{st['src'][1]}"
            for u in range(su[j]):
                L.append(f"  {act} p{j}_{u} = ({act}){src}[tid + {u * TAIL_BLOCK} < rows{j} ? tid + {u * TAIL_BLOCK} : rows{j} - 1];")
    # deferred second stages over <= 64 partials: wave w folds every fourth of them with shuffles —
    # no LDS scratch, no barrier per reduction (a wide graph hands over ~3 per likelihood term)
    wsteps = [j for j, st in enumerate(steps) if st["op"] == "rsum" and wave[j]]
    if wsteps:
        L.append("  {")
        L.append("    const int wv_ = tid >> 6, ln_ = tid & 63;")
        for w in range(TAIL_BLOCK // 64):
            mine = wsteps[w :: TAIL_BLOCK // 64]
            if not mine:
                continue
            L.append(f"    if (wv_ == {w}) {{")
            wq = sizes.get("wave_q") or [1] * len(steps)
            for j in mine:
                st = steps[j]
                act = CTYPE[st["acc_dtype"]]
                for q in range(wq[j]):
                    L.append(f"      {act} w{j}_{q} = ({act})e{st['src'][1]}[ln_ + {64 * q} < rows{j} ? ln_ + {64 * q} : rows{j} - 1];")
            L.append("      __builtin_amdgcn_sched_barrier(0);")
            for j in mine:
                st = steps[j]
                act = CTYPE[st["acc_dtype"]]
                op = REDUCE_OPS[st["red"]]
                for q in range(wq[j]):
                    L.append(f"      if (ln_ + {64 * q} >= rows{j}) w{j}_{q} = pthip_dev::{op}::identity<{act}>();")
                L.append(f"      {act} w{j} = w{j}_0;")
                for q in range(1, wq[j]):
                    L.append(f"      w{j} = pthip_dev::{op}::apply(w{j}, w{j}_{q});")
            if TAIL_WAVE_FOLD_INTERLEAVED:
                # step-major: every butterfly level runs over ALL of this wave's reductions before the next level —
                # their cross-lane exchanges (two ds_bpermute per double, ~100 cycles each) are in flight together.
                # Value by value (round 5) a wide graph's ~36 reductions per wave were 36 x 6 dependent exchanges:
                # most of the 25 + 39 us of north_star's two tail launches (profiles/r7_wide200_*).  Same butterfly
                # per value: the same bits.
                L.append("#pragma unroll")
                L.append("      for (int off_ = 32; off_ > 0; off_ >>= 1) {")
                for j in mine:
                    L.append(f"        const auto x{j}_ = pthip_dev::shfl_xor_any(w{j}, off_);")
                for j in mine:
                    L.append(f"        w{j} = pthip_dev::{REDUCE_OPS[steps[j]['red']]}::apply(w{j}, x{j}_);")
                L.append("      }")
            else:
                for j in mine:
                    L.append(f"      w{j} = pthip_dev::wave_reduce<pthip_dev::{REDUCE_OPS[steps[j]['red']]}>(w{j});")
            for j in mine:
                st = steps[j]
                L.append(f"      if (ln_ == 0) l{st['out']}[0] = ({CTYPE[st['dtype']]})w{j};")
            L.append("    }")
        L.append("  }")
    L.append("  __builtin_amdgcn_sched_barrier(0);")
    for j, st in enumerate(steps):
        if st["op"] == "finish":
            R, U = su[j]
            for u in range(U):
                for r in range(R):
                    L.append(f"  if ({r} >= rows{j}) f{j}_{u}_{r} = 0;")
        elif st["op"] == "rsum" and not wave[j]:
            act = CTYPE[st["acc_dtype"]]
            for u in range(su[j]):
                L.append(f"  if (tid + {u * TAIL_BLOCK} >= rows{j}) p{j}_{u} = pthip_dev::{REDUCE_OPS[st['red']]}::identity<{act}>();")

    def operand(ref, mode, u):
        kind, k = ref
        if kind == "e":
            e = ext[k]
            if e["kind"] == "C":
                return f"c{k}"
            if e["kind"] == "V" and mode == "V":
                return f"v{k}_{u}"
            return f"s{k}"
        return f"l{k}[tid + {u * TAIL_BLOCK}]" if mode == "V" else f"l{k}[0]"

    # phases: a step reads LDS slots written in earlier phases only, so the steps of one phase need
    # no barrier between them (slots are written once); one __syncthreads() per phase instead of one
    # per step — the scalar bookkeeping of a wide graph is dozens of independent one-element steps
    def lds_reads(st):
        refs = []
        if st["op"] == "finish" and st.get("y") is not None:
            refs.append(st["y"])
        elif st["op"] == "ew":
            refs += list(st["ins"])
        elif st["op"] == "scatter":
            refs += [st["base"], *st["ys"]]
        return [r[1] for r in refs if r[0] == "l"]

    def lds_writes(st):
        return list(st["outs"]) if st["op"] == "ew" else [st["out"]]

    writer, phase = {}, []
    for j, st in enumerate(steps):
        ph = 0
        for k in lds_reads(st):
            if k in writer:
                ph = max(ph, phase[writer[k]] + 1)
        if st["op"] == "rsum" and wave[j]:
            ph = 0
        phase.append(ph)
        for k in lds_writes(st):
            writer[k] = j
    # the wave-folded reductions were emitted above: everything that reads them is in phase >= 1
    if wsteps:
        L.append("  __syncthreads();")
    def scalar_only(st):
        """an Elemwise step whose operands are all one-element values and that reduces nothing: n == 1 by construction"""
        return st["op"] == "ew" and all(m in "SC" for m in st["modes"]) and all(r is None for r in st["reduce"])

    for ph in range(max(phase, default=-1) + 1):
        emitted = False
        n_scalar = 0
        for j, st in enumerate(steps):
            if phase[j] != ph or (st["op"] == "rsum" and wave[j]):
                continue
            emitted = True
            if TAIL_SCALAR_STEPS_BY_WAVE and scalar_only(st):
                # The steps of a phase are independent of each other, and a one-element step is one lane's work: the
                # first lane of wave (k mod 4) takes the k-th of them, so four run side by side instead of thread 0
                # running all of them in a row (a wide graph: ~30 scalar Composites with an exp each per phase).
                body = st["body"]
                w_ = n_scalar % (TAIL_BLOCK // 64)
                n_scalar += 1
                L.append(f"  // step {j}: one-element Elemwise (operand modes {st['modes']}), on wave {w_}")
                L.append(f"  if (tid == {64 * w_}) {{")
                in_names = [operand(ref, m, 0) for ref, m in zip(st["ins"], st["modes"])]
                out_names = []
                for q, dt in enumerate(body["out_dtypes"]):
                    L.append(f"    {CTYPE[dt]} o{q};")
                    out_names.append(f"o{q}")
                L.append(emit_body(body, in_names, out_names, indent="    "))
                for q in range(len(body["out_dtypes"])):
                    L.append(f"    l{st['outs'][q]}[0] = o{q};")
                L.append("  }")
                continue
            if st["op"] == "finish":
                ct = CTYPE[st["dtype"]]
                R, U = su[j]
                L.append(f"  // step {j}: second stage + epilogue of a split Gemv / scatter-add (rows even/odd, then the pair: the order of the looping form)")
                for u in range(U):
                    L.append(f"  if (tid + {u * TAIL_BLOCK} < M{j}) {{")
                    L.append(f"    {ct} a0 = 0, a1 = 0;")
                    for r in range(0, R, 2):
                        L.append(f"    a0 += f{j}_{u}_{r}; a1 += f{j}_{u}_{r + 1};")
                    L.append(f"    {ct} r = ({ct})alpha{j} * (a0 + a1);")
                    if st.get("y") is not None:
                        L.append(f"    if (beta{j} != 0.0) r += ({ct})beta{j} * ({ct}){operand(st['y'], st['ymode'], u)};")
                    L.append(f"    l{st['out']}[tid + {u * TAIL_BLOCK}] = r;")
                    L.append("  }")
            elif st["op"] == "rsum":
                act, oct_ = CTYPE[st["acc_dtype"]], CTYPE[st["dtype"]]
                op = REDUCE_OPS[st["red"]]
                L.append(f"  // step {j}: deferred second stage of a fused Elemwise+reduce kernel")
                L.append("  {")
                L.append(f"    {act} a = pthip_dev::{op}::identity<{act}>();")
                for u in range(su[j]):
                    L.append(f"    a = pthip_dev::{op}::apply(a, p{j}_{u});")
                L.append(f"    a = pthip_dev::block_reduce<pthip_dev::{op}, {act}, {TAIL_BLOCK}, true>(a, ({act}*)red_);")
                L.append(f"    if (tid == 0) l{st['out']}[0] = ({oct_})a;")
                L.append("  }")
            elif st["op"] == "scatter":
                ct = CTYPE[st["dtype"]]
                L.append(f"  // step {j}: a chain of one-element IncSubtensor updates (widefuse.collect_scalar_updates)")
                for u in range(su[j]):
                    L.append(f"  if (tid + {u * TAIL_BLOCK} < n{j}) l{st['out']}[tid + {u * TAIL_BLOCK}] = ({ct}){operand(st['base'], st['bmode'], u)};")
                L.append("  __syncthreads();")
                L.append("  if (tid == 0) {")
                for k, is_set, y in zip(st["indices"], st["set"], st["ys"]):
                    L.append(f"    l{st['out']}[{k}] {'=' if is_set else '+='} ({ct}){operand(y, 'S', 0)};")
                L.append("  }")
            else:
                body, modes, red = st["body"], st["modes"], st["reduce"]
                L.append(f"  // step {j}: Elemwise over n{j} elements, operand modes {modes}")
                L.append("  {")
                for q, r in enumerate(red):
                    if r is not None:
                        act = CTYPE[r[1]]
                        L.append(f"    {act} acc{q} = pthip_dev::{REDUCE_OPS[r[0]]}::identity<{act}>();")
                for u in range(su[j]):
                    L.append(f"    if (tid + {u * TAIL_BLOCK} < n{j}) {{")
                    in_names = [operand(ref, m, u) for ref, m in zip(st["ins"], modes)]
                    out_names = []
                    for q, dt in enumerate(body["out_dtypes"]):
                        L.append(f"      {CTYPE[dt]} o{q};")
                        out_names.append(f"o{q}")
                    L.append(emit_body(body, in_names, out_names, indent="      "))
                    for q, r in enumerate(red):
                        if r is None:
                            L.append(f"      l{st['outs'][q]}[tid + {u * TAIL_BLOCK}] = o{q};")
                        else:
                            L.append(f"      acc{q} = pthip_dev::{REDUCE_OPS[r[0]]}::apply(acc{q}, ({CTYPE[r[1]]})o{q});")
                    L.append("    }")
                for q, r in enumerate(red):
                    if r is not None:
                        act = CTYPE[r[1]]
                        L.append(f"    acc{q} = pthip_dev::block_reduce<pthip_dev::{REDUCE_OPS[r[0]]}, {act}, {TAIL_BLOCK}, true>(acc{q}, ({act}*)red_);")
                        L.append(f"    if (tid == 0) l{st['outs'][q]}[0] = ({CTYPE[r[2]]})acc{q};")
                L.append("  }")
        if emitted:
            L.append("  __syncthreads();")
    _tail_epilogue(L, spec)
    return "\n".join(L)
