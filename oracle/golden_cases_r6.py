"""Golden cases of round 6 (imported by ``make_golden.py``).  TEST INFRASTRUCTURE.

``c4_bigk_{K}_g{G}``   config #4's graph (ref_graphs.build_c4) with K in {1000, 2048, 4096} predictors and G in {128, 5000}
    groups: the instances of the one-pass ``gchain`` kernel that hold x in LDS and split the columns over 8 / 16 / 32
    chunks (``c8_g4``, ``c16_g2``, ``c32_g1``), which no fixture reached (``glm_fuzz_*`` stops at K = 257, G = 300).
    X (160 MB) is stored as its recipe — ``configs.c4_inputs(N, K, G)`` + SHA-256 — not as data.
``softmax_f32_offset``  column softmax / log-softmax of float32 operands shifted by 1e2 and 1e4, C-ordered and
    transposed (the statistic must stay in the accumulator dtype or the result is not shift-invariant).
``wide_200_gemm``       north_star's target with a real ``Gemm``: multi-response regression X(N,K) @ B(K,8) with a
    Cholesky prior on every column of B, next to the 48 likelihood terms (tensor/blas/gemm.py:76 via GemmOptimizer,
    tensor/rewriting/blas.py:437).
"""

from __future__ import annotations

import numpy as np
import pytensor.tensor as pt

import ref_graphs
from make_golden import case
from pytensor_amd import configs

_BIGK = {1000: 20000, 2048: 10000, 4096: 5000}  # K -> N (X is 160 MB in each)


def _make_bigk(K, G):
    kw = {"N": _BIGK[K], "K": K, "G": G, "seed": 60 + G % 7}

    def build():
        vals = configs.c4_inputs(**kw)
        params, outs = ref_graphs.build_c4(vals)
        return params, outs, vals

    return kw, build


for _K in _BIGK:
    for _G in (128, 5000):
        _kw, _b = _make_bigk(_K, _G)
        case(f"c4_bigk_{_K}_g{_G}", rtol=1e-10, generated={"fn": "c4_inputs", "kwargs": _kw, "names": ["X", "y", "gidx", "Sigma"]})(_b)


@case("softmax_f32_offset", rtol=1e-5)
def softmax_f32_offset():
    from pytensor.tensor.special import log_softmax, softmax

    rng = np.random.default_rng(601)
    A, B = pt.fmatrix("A"), pt.fmatrix("B")
    T = pt.ftensor3("T")
    outs = [softmax(A, axis=0), log_softmax(A, axis=0), softmax(B.T, axis=1), softmax(B, axis=0), log_softmax(B, axis=0),
            softmax(T, axis=1), softmax(T.transpose(2, 0, 1), axis=-1), softmax(A, axis=1), softmax(B, axis=None)]
    return [A, B, T], outs, {"A": (rng.normal(size=(257, 33)) + 100.0).astype("float32"), "B": (rng.normal(size=(130, 17)) + 1.0e4).astype("float32"),
                            "T": (rng.normal(size=(5, 65, 9)) * 3 - 1.0e4).astype("float32")}


@case("wide_200_gemm", rtol=1e-10)
def wide_200_gemm():
    from ref_graphs import build_wide200_gemm

    vals = configs.wide200_gemm_inputs(N=257, K=16, G=8, R=8)
    ins, outs = build_wide200_gemm(vals)
    return ins, outs, vals


@case("c4_gemm_multiresponse", rtol=1e-10)
def c4_gemm_multiresponse():
    # the same model without the 48 terms, odd sizes: R = 5 response columns, K = 37, N = 1031, 11 groups
    from ref_graphs import build_wide200_gemm

    vals = configs.wide200_gemm_inputs(N=1031, K=37, G=11, R=5, T=0, seed=3)
    ins, outs = build_wide200_gemm(vals, T=0)
    return ins, outs, vals


@case("r6_refsuite_fixes", rtol=1e-13)
def r6_refsuite_fixes():
    """What the reference's own test modules found at the end of round 6 (tests/test_gpu_refsuite_math.py), pinned as vectors:
    integer powers (the device library's pow(19, 3) = 6858.999... truncated to 6858), the gradient of ``prod`` with zeros in
    the input (``ProdWithoutZeros``, tensor/math.py:3786-3825), ``argmax`` of float16, ``bincount`` of 1- and 2-byte
    integers (an inc_subtensor of ones), a batched tridiagonal solve."""
    import pytensor
    from pytensor.tensor.slinalg import solve

    rng = np.random.default_rng(611)
    x = pt.dmatrix("x")
    xi, yi = pt.lvector("xi"), pt.lvector("yi")
    xf, yf = pt.dvector("xf"), pt.dvector("yf")
    h = pt.matrix("h", dtype="float16")
    b8, b16 = pt.vector("b8", dtype="int8"), pt.vector("b16", dtype="uint16")
    A, B = pt.dtensor3("A"), pt.dtensor3("B")
    xv = rng.normal(size=(5, 6))
    xv[0, 2] = 0.0
    xv[1, [1, 4]] = 0.0
    xv[3, :] = 0.0
    gi, gj = np.meshgrid(np.arange(-12, 13), np.arange(0, 9))
    xfv = np.array([3.0, 19.0, 11.0, 13.0, 8.0, 2.0, 10.0, 1.5, 0.3, 7.0, -3.0, 2.0])
    yfv = np.array([1.0, 3.0, 5.0, 6.0, 7.0, -3.0, -2.0, 2.0, 3.0, 0.5, 3.0, 0.0])
    Av = rng.normal(size=(3, 7, 7)) + 4.0 * np.eye(7)
    outs = [pt.prod(x, axis=1), pytensor.grad(pt.prod(x, axis=1).sum(), x), pytensor.grad(pt.prod(x), x), pt.pow(xi, yi), pt.pow(xf, yf),
            pt.cast(pt.pow(xf, yf), "int64"), pt.argmax(h, axis=1), pt.argmax(h, axis=None), pt.bincount(b8), pt.bincount(b16, minlength=20),
            solve(A, B, assume_a="tridiagonal")]
    vals = {"x": xv, "xi": gi.ravel().astype(np.int64), "yi": gj.ravel().astype(np.int64), "xf": xfv, "yf": yfv,
            "h": rng.normal(size=(9, 33)).astype(np.float16), "b8": rng.integers(0, 12, 300).astype(np.int8),
            "b16": rng.integers(0, 15, 500).astype(np.uint16), "A": Av, "B": rng.normal(size=(3, 7, 4))}
    return [x, xi, yi, xf, yf, h, b8, b16, A, B], outs, vals


@case("r6_device_math", rtol=1.2e-15)
def r6_device_math():
    """The generated kernels' own fp64 log / log1p (codegen.PRELUDE pt_log / pt_log1p, round 6) and what calls them (softplus,
    the shared sigmoid / softplus pair, log1mexp) against the reference's libm values, at a tolerance of ~5 ulp: arguments of
    every magnitude and both signs, the neighbourhoods of the reduction's switch points, subnormals."""
    rng = np.random.default_rng(612)
    n = 1500
    r = rng.random
    pos = np.concatenate([r(n) * 4.0, np.exp((r(n) - 0.5) * 1400.0), 1.0 + (r(n) - 0.5) * 0.6, np.ldexp(r(n) + 0.5, rng.integers(-1074, 1023, n)),
                          1.41421356 + (r(n) - 0.5) * 1e-6, 0.70710678 + (r(n) - 0.5) * 1e-6])
    pos = pos[(pos > 0) & np.isfinite(pos)]
    l1 = np.concatenate([r(n), -r(n) * 0.999999, np.exp((r(n) - 0.5) * 80.0), -np.exp(-r(n) * 40.0), (r(n) - 0.5) * 1.2,
                         0.41421356 + (r(n) - 0.5) * 1e-6, -0.29289321 + (r(n) - 0.5) * 1e-6, np.exp(-r(n) * 700.0)])
    sp = np.concatenate([rng.standard_normal(n) * 8.0, rng.uniform(-700.0, 700.0, n), rng.standard_normal(n) * 1e-3])
    neg = -np.exp((r(n) - 0.5) * 20.0)
    p, q, s, m = pt.dvector("p"), pt.dvector("q"), pt.dvector("s"), pt.dvector("m")
    outs = [pt.log(p), pt.log1p(q), pt.softplus(s), pt.sigmoid(s) * 1.0, pt.sigmoid(s) + pt.softplus(s), pt.log1mexp(m), pt.log(p) * pt.log1p(p)]
    return [p, q, s, m], outs, {"p": pos, "q": l1, "s": sp, "m": neg}
