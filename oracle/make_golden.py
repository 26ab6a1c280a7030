"""Generate ``tests/golden/*`` from the reference itself.  TEST INFRASTRUCTURE.

For every case this script (run in the build container, where
``/root/reference`` exists):

1. builds the graph with the reference's public API;
2. compiles it with ``mode="HIP"`` — i.e. through ``pytensor_amd.linker.HipLinker``,
   the product's own boundary — and stores the lowered IR (``<case>.json``);
3. evaluates the same graph with the reference C linker (``mode="CVM"``) and
   NumPy linker (``Mode("py", optimizer=None)`` as the backend tests do,
   tests/link/pytorch/test_basic.py:41-87) and stores inputs + both outputs in
   ``<case>.npz``;
4. checks ``oracle/np_graph.py`` on the IR against the C-linker outputs
   (the "pinning" of the oracle).

Usage:  python oracle/make_golden.py [case ...]
"""

from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_ref  # noqa: E402

make_ref.activate()

import numpy as np  # noqa: E402
import pytensor  # noqa: E402
import pytensor.tensor as pt  # noqa: E402
from pytensor.compile.mode import Mode  # noqa: E402
from pytensor.tensor.linalg import cho_solve, cholesky, solve_triangular  # noqa: E402

import pytensor_amd  # noqa: E402
from pytensor_amd import configs  # noqa: E402

pytensor_amd.register()
import np_graph  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = {}


PY_RTOL = {}  # case -> tolerance of the reference's own NumPy backend against its C backend
PY_LAZY = set()  # cases whose NumPy-backend run needs the lazy VM (IfElse): PerformLinker runs every node
PY_OPT = {}  # cases whose unoptimised NumPy-backend run fails inside the reference: rewrite set to use instead
GENERATED = {}  # case -> {"fn", "kwargs", "names"}: inputs stored as their recipe (tests/util._generated_inputs)


def case(name, rtol=None, py_rtol=None, lazy=False, py_optimizer=None, generated=None):
    """``py_rtol``: where the reference's two backends disagree beyond 1e-10 (Psi: AS 103 with
    10-digit constants in C, scipy.special.psi in Python) the C linker's values are the golden
    ones and the NumPy linker's are only sanity-checked at ``py_rtol``."""

    def deco(f):
        CASES[name] = (f, rtol)
        if py_rtol is not None:
            PY_RTOL[name] = py_rtol
        if lazy:
            PY_LAZY.add(name)
        if py_optimizer is not None:
            PY_OPT[name] = py_optimizer
        if generated is not None:
            GENERATED[name] = generated
        return f

    return deco


def _in(name, value):
    value = np.asarray(value)
    v = pt.tensor(name, dtype=str(value.dtype), shape=(None,) * value.ndim)
    return v


# ---------------------------------------------------------------------------
# BASELINE.json configs (small sizes; the IR is shape-agnostic in N)
# ---------------------------------------------------------------------------


@case("c1_gauss")
def c1():
    vals = configs.c1_inputs(N=1000)
    x = pt.dvector("x")
    mu = pt.dscalar("mu")
    y = pt.exp(-0.5 * (x - mu) ** 2).sum()
    return [x, mu], [y, pytensor.grad(y, x)], vals


def _c2_cheap(x, y):
    acc = x
    for k in range(4):
        u = acc * (0.9 + 0.01 * k) + y
        v = abs(u) - (0.3 + 0.1 * k) * x
        acc = pt.switch(v > 0, v * v, u - v) * 0.5 + acc * 0.25
    return acc + pt.exp(-x * x)


def _c2_transc(x, y):
    t = x
    s = pt.zeros_like(x)
    for k in range(10):
        t = pt.tanh(t + y * (0.1 * (k + 1)))
        s = s + pt.exp(-t * t) * (1.0 / (k + 1))
    return s


@case("c2_cheap")
def c2_cheap():
    vals = configs.c2_inputs(N=5000)
    x, y = pt.dvector("x"), pt.dvector("y")
    return [x, y], [_c2_cheap(x, y).sum()], vals


@case("c2_transc")
def c2_transc():
    vals = configs.c2_inputs(N=5000)
    x, y = pt.dvector("x"), pt.dvector("y")
    return [x, y], [_c2_transc(x, y).sum()], vals


@case("c3_dot22")
def c3_dot22():
    v = configs.c3_inputs(M=96, B=2, Bn=8)
    A, B = pt.dmatrix("A"), pt.dmatrix("B")
    return [A, B], [A @ B], {"A": v["A"], "B": v["B"]}


@case("c3_gemv")
def c3_gemv():
    v = configs.c3_inputs(M=96, B=2, Bn=8)
    A, x = pt.dmatrix("A"), pt.dvector("v")
    return [A, x], [A @ x], {"A": v["A"], "v": v["v"]}


@case("c3_bdot")
def c3_bdot():
    v = configs.c3_inputs(M=8, B=6, Bn=32)
    X, Y = pt.ftensor3("X3"), pt.ftensor3("Y3")
    return [X, Y], [pt.matmul(X, Y)], {"X3": v["X3"], "Y3": v["Y3"]}


from ref_graphs import build_c4  # noqa: E402


@case("c4_hier")
def c4():
    vals = configs.c4_inputs(N=3000)
    ins, outs = build_c4(vals)
    return ins, outs, vals


@case("c4_hier_small")
def c4_small():
    vals = configs.c4_inputs(N=257, K=16, G=8)
    ins, outs = build_c4(vals)
    return ins, outs, vals


def build_c5():
    xs = pt.ftensor3("xs")
    h0 = pt.fmatrix("h0")
    Ws = [pt.fmatrix(n) for n in ["Wz", "Wr", "Wh", "Uz", "Ur", "Uh"]]
    bs = [pt.fvector(n) for n in ["bz", "br", "bh"]]

    def step(x, h, Wz, Wr, Wh, Uz, Ur, Uh, bz, br, bh):
        zz = pt.sigmoid(x @ Wz + h @ Uz + bz)
        rr = pt.sigmoid(x @ Wr + h @ Ur + br)
        hh = pt.tanh(x @ Wh + (rr * h) @ Uh + bh)
        return (1 - zz) * h + zz * hh

    hs = pytensor.scan(step, sequences=[xs], outputs_info=[h0], non_sequences=Ws + bs, return_updates=False)
    return [xs, h0, *Ws, *bs], [hs[-1].sum(), hs[-1]]


@case("c5_gru", rtol=1e-5)
def c5():
    vals = configs.c5_inputs(T=7, B=4, H=32)
    ins, outs = build_c5()
    return ins, outs, vals


# ---------------------------------------------------------------------------
# op-level cases (shapes follow the reference's own tests)
# ---------------------------------------------------------------------------


@case("elemwise_bcast")
def elemwise_bcast():
    # tests/tensor/test_elemwise.py:239-443 TestBroadcast shape matrix
    rng = np.random.default_rng(10)
    a = pt.tensor("a", dtype="float64", shape=(None, None))
    b = pt.tensor("b", dtype="float64", shape=(1, None))
    c = pt.tensor("c", dtype="float64", shape=(None, 1))
    d = pt.dscalar("d")
    out1 = a * b + c - d
    out2 = pt.exp(a) / (1 + b**2) + pt.log1p(abs(c))
    vals = {"a": rng.normal(size=(7, 5)), "b": rng.normal(size=(1, 5)), "c": rng.normal(size=(7, 1)), "d": np.asarray(0.7)}
    return [a, b, c, d], [out1, out2], vals


@case("elemwise_f32_mixed", rtol=1e-5)
def elemwise_f32():
    rng = np.random.default_rng(11)
    a = pt.fmatrix("a")
    i = pt.lvector("i")
    out1 = pt.tanh(a) * i + pt.sigmoid(a)
    out2 = pt.softplus(a) - pt.sqrt(abs(a))
    out3 = pt.cast(a > 0, "int8") + pt.cast(i, "int8")
    vals = {"a": rng.normal(size=(6, 9)).astype("float32") * 3, "i": rng.integers(-3, 3, size=9)}
    return [a, i], [out1, out2, out3], vals


@case("elemwise_ints")
def elemwise_ints():
    rng = np.random.default_rng(12)
    i = pt.lvector("i")
    j = pt.lvector("j")
    out = [i // j, i % j, pt.maximum(i, j) - pt.minimum(i, j), pt.switch(pt.eq(i, j), i, -j), pt.bitwise_and(i, j) ^ 5, abs(i) * pt.sign(j)]
    jj = rng.integers(1, 9, size=33) * rng.choice([-1, 1], size=33)
    vals = {"i": rng.integers(-50, 50, size=33), "j": jj}
    return [i, j], out, vals


@case("careduce_axes")
def careduce_axes():
    # tests/tensor/test_elemwise.py:444-731 TestCAReduce / tests/benchmarks/test_careduce.py
    rng = np.random.default_rng(13)
    x = pt.dtensor3("x")
    outs = [
        x.sum(),
        x.sum(axis=0),
        x.sum(axis=1),
        x.sum(axis=2),
        x.sum(axis=(0, 1)),
        x.sum(axis=(1, 2)),
        x.sum(axis=(0, 2)),
        x.prod(axis=2),
        x.max(axis=1),
        x.min(axis=(0, 2)),
        x.max(),
    ]
    return [x], outs, {"x": rng.normal(size=(5, 17, 9))}


@case("careduce_f32_acc", rtol=1e-6)
def careduce_f32():
    # accumulator upcast f32 -> f64 (pytensor/tensor/elemwise.py:1383-1417)
    rng = np.random.default_rng(14)
    x = pt.fmatrix("x")
    b = pt.tensor("b", dtype="bool", shape=(None, None))
    return [x, b], [x.sum(), x.sum(axis=0), x.sum(axis=1), x.mean(axis=1), b.all(axis=0), b.any(axis=1), b.sum()], {
        "x": (rng.normal(size=(300, 41)) * 100).astype("float32"),
        "b": rng.random(size=(6, 7)) > 0.3,
    }


@case("softmax_family")
def softmax_family():
    rng = np.random.default_rng(15)
    x = pt.dmatrix("x")
    from pytensor.tensor.special import log_softmax, softmax

    return [x], [softmax(x, axis=-1), log_softmax(x, axis=-1), pt.logsumexp(x, axis=0), softmax(x, axis=0)], {
        "x": rng.normal(size=(9, 13)) * 4
    }


@case("cholesky_solve")
def cholesky_solve():
    # tests/tensor/linalg/test_decomposition/test_cholesky.py:28-125
    rng = np.random.default_rng(16)
    A = pt.dmatrix("A")
    b = pt.dvector("b")
    Bm = pt.dmatrix("Bm")
    L = cholesky(A)
    U = cholesky(A, lower=False)
    return [A, b, Bm], [
        L,
        U,
        solve_triangular(L, b, lower=True),
        solve_triangular(U, Bm, lower=False),
        solve_triangular(L.T, b, lower=False),
        cho_solve((L, True), b),
        cho_solve((U, False), Bm),
    ], {
        "A": (lambda M: M @ M.T + 12 * np.eye(12))(rng.normal(size=(12, 12))),
        "b": rng.normal(size=12),
        "Bm": rng.normal(size=(12, 5)),
    }


@case("cholesky_indefinite")
def cholesky_indefinite():
    # NaN on failure: cholesky.py:78-80; test_cholesky.py:57-70
    A = pt.dmatrix("A")
    return [A], [cholesky(A)], {"A": np.array([[1.0, 0.2], [0.2, -2.0]])}


@case("cholesky_batched")
def cholesky_batched():
    rng = np.random.default_rng(17)
    A = pt.dtensor3("A")
    b = pt.dmatrix("b")
    L = cholesky(A)
    M = rng.normal(size=(4, 6, 6))
    return [A, b], [L, solve_triangular(L, b, lower=True, b_ndim=1)], {
        "A": M @ M.transpose(0, 2, 1) + 6 * np.eye(6),
        "b": rng.normal(size=(4, 6)),
    }


@case("indexing")
def indexing():
    rng = np.random.default_rng(18)
    x = pt.dmatrix("x")
    v = pt.dvector("v")
    idx = pt.lvector("idx")
    k = pt.lscalar("k")
    outs = [
        x[1:5],
        x[::2, 1],
        x[k],
        x[:, -3:],
        x[idx],
        v[idx],
        pt.set_subtensor(x[2:4], 7.0),
        pt.inc_subtensor(x[:, 1], v[: x.shape[0]]),
        pt.inc_subtensor(v[idx], 1.5),
        pt.set_subtensor(v[idx[:3]], pt.stack([v[0], v[1], v[2]])),
        pt.concatenate([x, x[::-1]], axis=0),
        x.reshape((-1,)),
        x.T.reshape((x.shape[1] * 2, x.shape[0] // 2)),
        pt.diag(x[:6, :6]),
        pt.alloc(v[0], 3, 4),
    ]
    return [x, v, idx, k], outs, {
        "x": rng.normal(size=(8, 6)),
        "v": rng.normal(size=11),
        "idx": np.array([0, 3, 3, 7, 2, 3]),
        "k": np.asarray(5),
    }


@case("gemm_variants")
def gemm_variants():
    # tests/tensor/test_blas.py:82-404 TestGemm (transposes, alpha/beta)
    rng = np.random.default_rng(19)
    A, B, C = pt.dmatrix("A"), pt.dmatrix("B"), pt.dmatrix("C")
    x, y = pt.dvector("x"), pt.dvector("y")
    outs = [
        0.4 * C + 0.8 * pt.dot(A, B),
        C - pt.dot(B.T, A.T).T * 2.0,
        pt.dot(A.T, A),
        pt.dot(A, A.T),
        y * 0.5 + 1.5 * pt.dot(A, x),
        x + pt.dot(A.T, y),
        pt.outer(y, x) + A,
        pt.dot(x, x),
    ]
    return [A, B, C, x, y], outs, {
        "A": rng.normal(size=(21, 13)),
        "B": rng.normal(size=(13, 17)),
        "C": rng.normal(size=(21, 17)),
        "x": rng.normal(size=13),
        "y": rng.normal(size=21),
    }


@case("scan_cumsum_taps")
def scan_taps():
    # tests/link/jax/test_scan.py patterns: sit-sot, mit-sot (fibonacci-like), nit-sot
    rng = np.random.default_rng(20)
    xs = pt.dmatrix("xs")
    s0 = pt.dvector("s0")
    f0 = pt.dmatrix("f0")
    w = pt.dscalar("w")

    def step(x, s, fm2, fm1, w):
        s_new = s * w + x
        f_new = fm1 + 0.5 * fm2
        return s_new, f_new, pt.tanh(s_new).sum()

    (ss, fs, ns) = pytensor.scan(
        step,
        sequences=[xs],
        outputs_info=[s0, dict(initial=f0, taps=[-2, -1]), None],
        non_sequences=[w],
        return_updates=False,
    )
    return [xs, s0, f0, w], [ss, fs, ns, ss[-1]], {
        "xs": rng.normal(size=(9, 4)),
        "s0": rng.normal(size=4),
        "f0": rng.normal(size=(2, 4)),
        "w": np.asarray(0.9),
    }


@case("edge_empty")
def edge_empty():
    # empty dims: tests/tensor/test_elemwise.py:446-462 (TestCAReduce), test_cholesky.py empty case
    x = pt.dmatrix("x")
    v = pt.dvector("v")
    A = pt.dmatrix("A")
    outs = [
        pt.exp(x) + 1.0,
        x.sum(),
        x.sum(axis=0),
        x.sum(axis=1),
        x.prod(),
        pt.exp(v).sum(),
        pt.dot(A, v),
        pt.dot(x.T, x),
        cholesky(pt.dot(x.T, x)[:0, :0]),
    ]
    return [x, v, A], outs, {"x": np.zeros((0, 5)), "v": np.zeros((0,)), "A": np.zeros((4, 0))}


@case("edge_nan_reduce")
def edge_nan_reduce():
    # NaN handling of max/min reductions: tests/tensor/test_elemwise.py:611,677
    rng = np.random.default_rng(30)
    x = pt.dmatrix("x")
    xv = rng.normal(size=(6, 70))
    xv[1, 3] = np.nan
    xv[4, 69] = np.inf
    xv[5, 0] = -np.inf
    return [x], [x.max(axis=1), x.min(axis=0), x.max(), x.min(), x.sum(axis=1), pt.isnan(x).any(axis=1), pt.maximum(x, 0.5), pt.switch(pt.isnan(x), 0.0, x).sum()], {"x": xv}


@case("edge_noncontig")
def edge_noncontig():
    # non-contiguous operands: tests/tensor/test_blas.py:344 (TestGemm non-contiguous),
    # TestBlasStrides 1943-2327, test_elemwise.py TestBroadcast with transposes
    rng = np.random.default_rng(31)
    A = pt.dmatrix("A")
    B = pt.dmatrix("B")
    v = pt.dvector("v")
    outs = [
        pt.exp(A.T) + B[::2, ::-1][: A.shape[1], : A.shape[0]],
        A[::2].sum(axis=0),
        A[:, ::3].sum(axis=1),
        A.T[1:].sum(),
        pt.dot(A[::2], B[:, ::2].T[: A.shape[1]]),
        pt.dot(A.T, A[:, ::-1]),
        pt.dot(A[::-1], v[::2]),
        pt.dot(A.T[:, ::2], v[1::4][: (A.shape[0] + 1) // 2]),
        (A[1:, 1:] * B[:-1, :-1][: A.shape[0] - 1, : A.shape[1] - 1]).max(axis=0),
    ]
    return [A, B, v], outs, {"A": rng.normal(size=(10, 14)), "B": rng.normal(size=(28, 30)), "v": rng.normal(size=28)}


@case("edge_odd_sizes")
def edge_odd_sizes():
    # sizes that are not multiples of the 16-byte vector width / wave / tile sizes
    rng = np.random.default_rng(32)
    x = pt.dvector("x")
    y = pt.fvector("y")
    A = pt.dmatrix("A")
    F = pt.fmatrix("F")
    outs = [
        pt.tanh(x[1:]) * x[:-1],
        (x * 2.0).sum(),
        pt.exp(y)[1:].sum(),
        y[3:] + y[:-3],
        pt.dot(A, A.T),
        pt.dot(F.T, F),
        pt.dot(A, x[: A.shape[1]]),
        pt.dot(F.T, y[: F.shape[0]]),
        x[:1] ** 2,
    ]
    return [x, y, A, F], outs, {
        "x": rng.normal(size=131),
        "y": rng.normal(size=67).astype("float32"),
        "A": rng.normal(size=(33, 17)),
        "F": rng.normal(size=(19, 35)).astype("float32"),
    }


@case("careduce_int_bool")
def careduce_int_bool():
    rng = np.random.default_rng(33)
    i = pt.lmatrix("i")
    s = pt.tensor("s", dtype="int8", shape=(None, None))
    b = pt.tensor("b", dtype="bool", shape=(None, None))
    return [i, s, b], [i.sum(axis=0), i.prod(axis=1), i.max(), i.min(axis=0), s.sum(), s.max(axis=1), b.all(), b.any(axis=0), b.sum(axis=1)], {
        "i": rng.integers(-9, 9, size=(7, 11)),
        "s": rng.integers(-100, 100, size=(5, 9)).astype("int8"),
        "b": rng.random(size=(4, 6)) > 0.5,
    }


@case("scalar_math_f64", rtol=1e-11)
def scalar_math_f64():
    # every unary float ScalarOp of pytensor/scalar/basic.py + math.py that the hot path can
    # meet inside a Composite, one output per op so that nothing cancels or is rewritten away
    rng = np.random.default_rng(30)
    x = pt.dvector("x")  # (-3, 3)
    u = pt.dvector("u")  # (-0.95, 0.95)
    p = pt.dvector("p")  # (0.1, 4)
    q = pt.dvector("q")  # (1.05, 6)
    outs = [
        pt.sin(x), pt.cos(x), pt.tan(x * 0.4), pt.sinh(x), pt.cosh(x), pt.arcsinh(x), pt.arctan(x),
        pt.arcsin(u), pt.arccos(u), pt.arctanh(u), pt.arccosh(q),
        pt.exp2(x), pt.expm1(x * 1e-3), pt.expm1(x), pt.log2(p), pt.log10(p), pt.log1p(u), pt.log1mexp(-p),
        pt.deg2rad(x), pt.rad2deg(x), pt.reciprocal(q), pt.sqr(x), pt.sqrt(p), pt.sigmoid(x * 10), pt.softplus(x * 12),
        pt.ceil(x * 3), pt.floor(x * 3), pt.trunc(x * 3), pt.round(x * 3, mode="half_to_even"),
        pt.round(x * 3, mode="half_away_from_zero"), pt.sign(x), abs(x), -x,
    ]
    n = 257
    xs = rng.uniform(-3, 3, size=n)
    xs[:6] = [0.5, -0.5, 1.5, -1.5, 2.5 / 3, -2.5 / 3]  # ties for the two rounding modes (x*3)
    vals = {"x": xs, "u": rng.uniform(-0.95, 0.95, size=n), "p": rng.uniform(0.1, 4, size=n), "q": rng.uniform(1.05, 6, size=n)}
    return [x, u, p, q], outs, vals


@case("scalar_special_f64", rtol=1e-10, py_rtol=1e-8)
def scalar_special_f64():
    # scalar/math.py: Erf 61, Erfc 110, Erfcx 156, Erfinv 209, Erfcinv 246, Gamma 292, GammaLn 340, Psi 370
    rng = np.random.default_rng(31)
    x = pt.dvector("x")  # (-3, 3)
    u = pt.dvector("u")  # (-0.95, 0.95)
    p = pt.dvector("p")  # (0.2, 8)
    outs = [pt.erf(x), pt.erfc(x), pt.erfcx(x), pt.erfinv(u), pt.erfcinv(u + 1.0), pt.gamma(p), pt.gammaln(p), pt.psi(p),
            pt.gammaln(p * 40), pt.psi(p * 40), pt.erfc(x * 4), pt.erfcx(x * 9)]
    n = 193
    vals = {"x": rng.uniform(-3, 3, size=n), "u": rng.uniform(-0.95, 0.95, size=n), "p": rng.uniform(0.2, 8, size=n)}
    return [x, u, p], outs, vals


@case("scalar_binary_logic")
def scalar_binary_logic():
    rng = np.random.default_rng(32)
    x = pt.dvector("x")
    y = pt.dvector("y")
    p = pt.dvector("p")
    i = pt.lvector("i")
    j = pt.lvector("j")
    b = pt.bvector("b")
    outs = [
        pt.pow(p, y), pt.pow(x, 3), pt.arctan2(x, y), pt.maximum(x, y), pt.minimum(x, y), pt.clip(x, -0.5, y * 0 + 0.75),
        pt.lt(x, y), pt.le(x, y), pt.gt(x, y), pt.ge(x, y), pt.eq(i, j), pt.neq(i, j),
        pt.or_(i, j), pt.and_(i, j), pt.xor(i, j), pt.invert(i), pt.invert(b),
        pt.second(x, y), pt.switch(pt.lt(x, y), x, y), pt.isnan(x / (y - y)), pt.isinf(1.0 / (i - i + 0.0 * x)),
        pt.mod(x, y), pt.true_div(i, j + 100), pt.maximum(i, j), pt.minimum(i, j), pt.pow(i % 5, 3),
    ]
    n = 129
    vals = {
        "x": rng.normal(size=n), "y": rng.normal(size=n), "p": rng.uniform(0.1, 3, size=n),
        "i": rng.integers(-60, 60, size=n), "j": rng.integers(-60, 60, size=n), "b": rng.integers(-100, 100, size=n).astype("int8"),
    }
    return [x, y, p, i, j, b], outs, vals


@case("scalar_math_f32", rtol=2e-5)
def scalar_math_f32():
    rng = np.random.default_rng(33)
    x = pt.fvector("x")
    u = pt.fvector("u")
    p = pt.fvector("p")
    outs = [
        pt.sin(x), pt.cos(x), pt.tan(x * 0.4), pt.sinh(x), pt.cosh(x), pt.arcsinh(x), pt.arctan(x), pt.arcsin(u), pt.arccos(u),
        pt.arctanh(u), pt.exp2(x), pt.expm1(x), pt.log2(p), pt.log10(p), pt.log1p(u), pt.erf(x), pt.erfc(x), pt.gammaln(p),
        pt.exp(x), pt.log(p), pt.tanh(x), pt.sigmoid(x * 5), pt.softplus(x * 6), pt.sqrt(p), pt.reciprocal(p), pt.pow(p, x),
        pt.ceil(x * 3), pt.floor(x * 3), pt.trunc(x * 3), pt.round(x * 3), pt.maximum(x, u), pt.arctan2(x, u),
    ]
    n = 200
    vals = {"x": rng.uniform(-3, 3, size=n).astype("float32"), "u": rng.uniform(-0.95, 0.95, size=n).astype("float32"),
            "p": rng.uniform(0.2, 6, size=n).astype("float32")}
    return [x, u, p], outs, vals


# ---------------------------------------------------------------------------


def _tolerance(dtypes, rtol):
    if rtol is not None:
        return rtol
    if any(np.dtype(d) == np.float32 for d in dtypes):
        return 1e-5
    return 1e-12


def generate(name):
    f, rtol = CASES[name]
    ins, outs, vals = f()
    fn_hip = pytensor.function(ins, outs, mode="HIP", on_unused_input="ignore")
    graph = fn_hip.maker.linker.last_ir
    host_nodes = [n.params["name"] for n in graph.nodes if n.op == "HostPerform"]
    if host_nodes:
        raise RuntimeError(f"{name}: ops without a hip lowering: {host_nodes}")
    # input values in fgraph.inputs order (explicit + shared)
    fg_inputs = fn_hip.maker.fgraph.inputs
    names = []
    in_vals = []
    for v, cont in zip(fg_inputs, fn_hip.input_storage):
        nm = v.name
        names.append(nm)
        if nm in vals:
            in_vals.append(np.asarray(vals[nm], dtype=v.type.dtype))
        elif isinstance(cont.storage[0], np.random.Generator):
            in_vals.append(cont.storage[0])  # a shared RNG: stored as its Philox key + counter words
        else:
            in_vals.append(np.asarray(cont.storage[0]))
    explicit = [np.asarray(vals[v.name], dtype=v.type.dtype) for v in ins]

    fn_c = pytensor.function(ins, outs, mode="CVM", on_unused_input="ignore")
    out_c = [np.asarray(o) for o in fn_c(*explicit)]
    if name in PY_LAZY:
        from pytensor.link.vm import VMLinker

        py_mode = Mode(VMLinker(use_cloop=False, c_thunks=False), optimizer=None)  # Python thunks, lazy
    else:
        py_mode = Mode("py", optimizer=PY_OPT.get(name))
    fn_py = pytensor.function(ins, outs, mode=py_mode, on_unused_input="ignore")
    out_py = [np.asarray(o) for o in fn_py(*explicit)]

    out_or = np_graph.run_graph(graph, in_vals)
    tol = _tolerance([o.dtype for o in out_c], rtol)
    for k, (a, b, c) in enumerate(zip(out_or, out_c, out_py)):
        a = np.asarray(a)
        assert a.shape == b.shape, (name, k, a.shape, b.shape)
        assert a.dtype == b.dtype, (name, k, a.dtype, b.dtype)
        if b.dtype.kind in "biu":
            np.testing.assert_array_equal(a, b, err_msg=f"{name} out{k} oracle vs C linker")
        else:
            np.testing.assert_allclose(a, b, rtol=tol, atol=tol * 1e-3, equal_nan=True, err_msg=f"{name} out{k} oracle vs C linker")
            ptol = PY_RTOL.get(name, max(tol, 1e-10))
            np.testing.assert_allclose(c, b, rtol=ptol, atol=max(tol, ptol * 1e-2), equal_nan=True, err_msg=f"{name} out{k} py vs C linker")

    os.makedirs(GOLDEN, exist_ok=True)
    d = graph.to_dict()
    d["input_names"] = names
    d["rtol"] = tol
    if name in PY_RTOL:
        d["py_rtol"] = PY_RTOL[name]
    skip = set()
    if name in GENERATED:
        import hashlib

        spec = GENERATED[name]
        regen = getattr(configs, spec["fn"])(**spec["kwargs"])
        sha = {}
        for k, nm in enumerate(names):
            if nm in spec["names"]:
                a = np.ascontiguousarray(regen[nm])
                np.testing.assert_array_equal(a, in_vals[k])
                sha[nm] = hashlib.sha256(a.tobytes()).hexdigest()
                skip.add(k)
        d["generated_inputs"] = {"fn": spec["fn"], "kwargs": spec["kwargs"], "sha256": sha}
    with open(os.path.join(GOLDEN, f"{name}.json"), "w") as fh:
        json.dump(d, fh, separators=(",", ":"))
    import philox_ref

    def storable(v):
        if isinstance(v, np.random.Generator):
            key, ctr = philox_ref.generator_state(v)
            return np.array([*key, *[(ctr >> (64 * j)) & philox_ref.MASK for j in range(4)]], dtype=np.uint64)
        return v

    arrays = {f"in{k}": storable(v) for k, v in enumerate(in_vals) if k not in skip}
    arrays.update({f"cvm{k}": v for k, v in enumerate(out_c)})
    arrays.update({f"py{k}": v for k, v in enumerate(out_py)})
    np.savez_compressed(os.path.join(GOLDEN, f"{name}.npz"), **arrays)
    print(f"{name:24s} ok  [{graph.summary()}]")


if __name__ == "__main__":
    sys.modules.setdefault("make_golden", sys.modules["__main__"])  # one CASES registry
    import golden_cases_fuzz  # noqa: F401
    import golden_cases_more  # noqa: F401  (registers its cases)
    import golden_cases_r2  # noqa: F401
    import golden_cases_r2b  # noqa: F401
    import golden_cases_r5  # noqa: F401
    import golden_cases_layout_fuzz  # noqa: F401
    import golden_cases_glm_fuzz  # noqa: F401
    import golden_cases_r6  # noqa: F401

    names = sys.argv[1:] or list(CASES)
    for n in names:
        generate(n)
