// exp_device.h — the 24-instruction fp64 exp of the generated kernels (pytensor_amd/codegen.py PRELUDE: pt_exp /
// pt_exp_k, where its derivation and accuracy figures are: <= 1 ulp on 6e5 points in [-700, 700]) for the hand-written
// kernels of csrc/softmax.hip.  The device library's exp is ~34 instructions and, in unrolled code with many
// instances, every polynomial coefficient is materialised again per instance (two v_mov_b32 of a literal per Horner
// step): the log-sum-exp kernels then ran ~70-90 VALU instructions per element and were issue-bound next to a 22 us
// HBM floor (profiles/r7f_lse_kernels.md).  With the constants held in VGPRs across instances (ExpK, loaded once per
// kernel) each Horner step is one v_fma_f64.
// Reference semantics: Exp.c_code is libm's exp (pytensor/scalar/basic.py:3085-3118).  Overflow -> inf,
// underflow -> 0, NaN -> NaN, exp(-inf) = 0, exp(+inf) = inf.
#pragma once
#include <hip/hip_runtime.h>

namespace pthip_dev {

struct ExpK { double l2e, nh, nl, c[10], hi, lo; };

static __device__ __forceinline__ ExpK expk_load() {
  ExpK k = {0x1.71547652b82fep+0, -0x1.62e42fee00000p-1, -0x1.a39ef35793c76p-33,
            {0x1.af38a9b0ec855p-26, 0x1.289185613a3d6p-22, 0x1.71de0dae63bb3p-19, 0x1.a019b90d2ae7ap-16, 0x1.a01a01a7c41d5p-13,
             0x1.6c16c1788bd90p-10, 0x1.11111111109b3p-7, 0x1.5555555553d63p-5, 0x1.5555555555556p-3, 0x1.0000000000001p-1},
            0x1.62e42fefa39efp+9, -0x1.74910d52d3051p+9};
  asm volatile("" : "+v"(k.l2e), "+v"(k.nh), "+v"(k.nl), "+v"(k.hi), "+v"(k.lo));
#pragma unroll
  for (int i = 0; i < 10; i++) asm volatile("" : "+v"(k.c[i]));
  return k;
}

static __device__ __forceinline__ double exp_k(double x, const ExpK& k) {
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

// what a kernel templated on T carries: the constants for double, nothing for float (expf is 15 instructions of
// float arithmetic whose literals are inline operands)
template <class T> struct ExpCtx;
template <> struct ExpCtx<double> {
  ExpK k;
  __device__ __forceinline__ ExpCtx() : k(expk_load()) {}
  __device__ __forceinline__ double operator()(double x) const { return exp_k(x, k); }
};
template <> struct ExpCtx<float> {
  __device__ __forceinline__ ExpCtx() {}
  __device__ __forceinline__ float operator()(float x) const { return expf(x); }
};

}  // namespace pthip_dev
