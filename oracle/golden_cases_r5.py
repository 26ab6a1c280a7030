"""Golden cases of round 5 (imported by ``make_golden.py``).  TEST INFRASTRUCTURE.

What they pin: the layouts the tiled N-d ``Elemwise`` kernel and the N-d ``CAReduce`` kernels were
written for — the reference's own benchmark graphs (tests/benchmarks/test_elemwise.py:7-28,
test_careduce.py:7-61, test_logsumexp.py:9-37) at sizes that are not multiples of any tile, vector or
wave width, plus the operand classes of a broadcasting loop (row / column broadcast, transposed,
strided, reversed, small inner dimension, mixed dtypes, > 2 collapsed dimensions).  The same IRs are
run at the benchmark sizes by ``tools/bench_hotpath.py``.
"""

from __future__ import annotations

import numpy as np
import pytensor.tensor as pt

from make_golden import case


@case("ew_rowcol_bcast")
def ew_rowcol_bcast():
    # Elemwise with static row / column broadcasts (tensor/elemwise.py:755-823; C loop
    # elemwise_cgen.py:212-465): A * r[None, :] + c[:, None]
    rng = np.random.default_rng(501)
    A, r, c = pt.dmatrix("A"), pt.dvector("r"), pt.dvector("c")
    F, rf = pt.fmatrix("F"), pt.fvector("rf")
    outs = [A * r[None, :] + c[:, None], pt.tanh(F * rf[None, :]) * rf[None, :], pt.cast(F, "float64") * r[None, :] - c[:, None] ** 2]
    return [A, r, c, F, rf], outs, {"A": rng.normal(size=(37, 53)), "r": rng.normal(size=53), "c": rng.normal(size=37),
                                    "F": rng.normal(size=(37, 53)).astype("float32"), "rf": rng.normal(size=53).astype("float32")}


@case("ew_rowbcast_f32", rtol=1e-5)
def ew_rowbcast_f32():
    rng = np.random.default_rng(509)
    F, rf = pt.fmatrix("F"), pt.fvector("rf")
    return [F, rf], [pt.tanh(F * rf[None, :]) * rf[None, :]], {"F": rng.normal(size=(37, 54)).astype("float32"), "rf": rng.normal(size=54).astype("float32")}


@case("ew_transposed")
def ew_transposed():
    # one operand arrives transposed (DimShuffle view, elemwise.py:41): A + B.T, and batched forms
    rng = np.random.default_rng(502)
    A, B = pt.dmatrix("A"), pt.dmatrix("B")
    X, Y = pt.dtensor3("X"), pt.dtensor3("Y")
    outs = [A + B.T, pt.exp(-abs(A)) * B.T - A, X + Y.transpose(0, 2, 1), X * Y.transpose(0, 2, 1) + X, A.T * 2.0 + B]
    return [A, B, X, Y], outs, {"A": rng.normal(size=(70, 45)), "B": rng.normal(size=(45, 70)),
                                "X": rng.normal(size=(3, 33, 65)), "Y": rng.normal(size=(3, 65, 33))}


@case("ew_a_plus_bt")
def ew_a_plus_bt():
    rng = np.random.default_rng(510)
    A, B = pt.dmatrix("A"), pt.dmatrix("B")
    return [A, B], [A + B.T], {"A": rng.normal(size=(70, 45)), "B": rng.normal(size=(45, 70))}


@case("ew_small_inner")
def ew_small_inner():
    rng = np.random.default_rng(511)
    S, s = pt.dmatrix("S"), pt.dvector("s")
    return [S, s], [S + s[None, :]], {"S": rng.normal(size=(211, 10)), "s": rng.normal(size=10)}


@case("ew_simple_bcast")
def ew_simple_bcast():
    # the reference's own elemwise benchmark graph (tests/benchmarks/test_elemwise.py:7-28)
    rng = np.random.default_rng(42)
    x, y = pt.matrix("y", dtype="float64"), pt.vector("z", dtype="float64")
    return [x, y], [pt.exp(2 * x * y + y)], {"y": rng.normal(size=(20, 50)), "z": rng.normal(size=50)}


@case("ew_nd_layouts")
def ew_nd_layouts():
    # strided / reversed / small-inner-dimension / 4-d outer-product broadcasts / mixed dtypes / a
    # reduction fused behind a broadcasting loop / two outputs of different dtypes
    rng = np.random.default_rng(503)
    x, y = pt.dvector("x"), pt.dvector("y")
    S, s3 = pt.dmatrix("S"), pt.dvector("s3")
    a, b = pt.dmatrix("a"), pt.dmatrix("b")
    F, d = pt.fmatrix("F"), pt.dvector("d")
    T = pt.dtensor3("T")
    outs = [
        x[::2] + y[: (x.shape[0] + 1) // 2],
        x[::-1] * 2.0 - x,
        S + s3[None, :],
        pt.exp(S) * S[:, ::-1],
        a[:, None, :, None] * b[None, :, None, :],
        F * d[None, :],
        (S * s3[None, :]).sum(),
        pt.gt(S, s3[None, :]),
        pt.cast(S * 3.0, "int32") + pt.cast(s3[None, :], "int32"),
        T * T[:, :, ::-1],
        T[:, ::2, :] + T[:, 1::2, :][:, : (T.shape[1] + 1) // 2, :],
        T.transpose(1, 0, 2) - 1.0,
    ]
    return [x, y, S, s3, a, b, F, d, T], outs, {
        "x": rng.normal(size=1001), "y": rng.normal(size=600), "S": rng.normal(size=(211, 3)), "s3": rng.normal(size=3),
        "a": rng.normal(size=(5, 7)), "b": rng.normal(size=(4, 9)), "F": rng.normal(size=(19, 23)).astype("float32"), "d": rng.normal(size=23),
        "T": rng.normal(size=(6, 10, 21)),
    }


def _careduce_layout_outputs(x, x2):
    # tests/benchmarks/test_careduce.py:7-35: c_contiguous / transposed (2,0,1) / strided ([::2] then (2,0,1))
    views = [x, x.transpose(2, 0, 1), x2[::2].transpose(2, 0, 1)]
    outs = []
    for v in views:
        for axis in (0, 1, 2, (0, 1), (0, 2), (1, 2), None):
            outs.append(v.sum(axis=axis))
    return outs


@case("careduce_layouts")
def careduce_layouts():
    rng = np.random.default_rng(504)
    x, x2 = pt.dtensor3("x"), pt.dtensor3("x2")
    return [x, x2], _careduce_layout_outputs(x, x2), {"x": rng.uniform(size=(5, 6, 7)), "x2": rng.uniform(size=(10, 6, 7))}


@case("careduce_layouts_big")
def careduce_layouts_big():
    # the same 21 reductions at sizes that cross the vector / wave / split thresholds
    rng = np.random.default_rng(505)
    x, x2 = pt.dtensor3("x"), pt.dtensor3("x2")
    return [x, x2], _careduce_layout_outputs(x, x2), {"x": rng.uniform(size=(33, 70, 129)), "x2": rng.uniform(size=(66, 70, 129))}


@case("careduce_ops_layouts", rtol=1e-6)
def careduce_ops_layouts():
    # other scalar ops / dtypes over the same layouts: Max/Min propagate NaN, Prod, f32 with f64 accumulator,
    # integer and bool inputs (elemwise.py:1383-1417, math.py:468-475,3438-3587)
    rng = np.random.default_rng(506)
    x, f = pt.dtensor3("x"), pt.ftensor3("f")
    i, b = pt.tensor("i", dtype="int32", shape=(None, None, None)), pt.tensor("b", dtype="bool", shape=(None, None, None))
    xt, ft, it, bt = (v.transpose(2, 0, 1) for v in (x, f, i, b))
    outs = [xt.max(axis=0), xt.max(axis=1), xt.min(axis=(0, 2)), x.max(axis=(0, 1)), (x * 0.5 + 1.0).prod(axis=1), (xt * 0.5 + 1.0).prod(axis=2),
            ft.sum(axis=0), ft.sum(axis=1), ft.sum(axis=2), f.sum(axis=(0, 2)), f.mean(axis=0), ft.max(axis=1),
            it.sum(axis=1), i.sum(axis=0), i.max(axis=2), bt.all(axis=1), bt.any(axis=0), b.sum(axis=(0, 1))]
    xv = rng.normal(size=(9, 20, 35))
    xv[3, 4, 5] = np.nan
    return [x, f, i, b], outs, {"x": xv, "f": (rng.normal(size=(9, 20, 35)) * 10).astype("float32"),
                                "i": rng.integers(-1000, 1000, size=(9, 20, 35)).astype("int32"), "b": rng.random(size=(9, 20, 35)) > 0.02}


def _logsumexp(X, axis):
    # the reference's benchmark graph, verbatim in structure (tests/benchmarks/test_logsumexp.py:9-13)
    X_max = pt.max(X, axis=axis, keepdims=True)
    X_max = pt.switch(pt.isinf(X_max), 0, X_max)
    return pt.log(pt.sum(pt.exp(X - X_max), axis=axis, keepdims=True)) + X_max


@case("logsumexp_bench")
def logsumexp_bench():
    rng = np.random.default_rng(23920)
    X = pt.matrix("X", dtype="float64")
    return [X], [_logsumexp(X, 0), _logsumexp(X, 1)], {"X": rng.normal(size=(37, 91))}


@case("logsumexp_axis0")
def logsumexp_axis0():
    rng = np.random.default_rng(23921)
    X = pt.matrix("X", dtype="float64")
    return [X], [_logsumexp(X, 0)], {"X": rng.normal(size=(37, 91))}


@case("logsumexp_axis1")
def logsumexp_axis1():
    rng = np.random.default_rng(23922)
    X = pt.matrix("X", dtype="float64")
    return [X], [_logsumexp(X, 1)], {"X": rng.normal(size=(37, 91))}


@case("logsumexp_degenerate")
def logsumexp_degenerate():
    """Shapes the one-pass log-sum-exp tile does not take (dispatch/elemwise.py _logsumexp_axis_by_axis): a reduced
    axis of extent 1 (found by layout_fuzz_f64_8), seven dimensions whose kept / reduced roles alternate (nothing
    merges), an empty kept dimension."""
    rng = np.random.default_rng(77123)
    A = pt.tensor("A", dtype="float64", shape=(None, None, None))
    B = pt.tensor("B", dtype="float64", shape=(None,) * 7)
    E = pt.matrix("E", dtype="float64")
    outs = [_logsumexp(A, 1), pt.logsumexp(A + 4.0, axis=(0, 1)), pt.logsumexp(B, axis=(1, 3, 5)), pt.logsumexp(B, axis=(0, 2, 4, 6)), _logsumexp(E, 1)]
    return [A, B, E], outs, {"A": rng.normal(size=(3, 1, 4)), "B": rng.normal(size=(2, 3, 2, 2, 3, 2, 2)), "E": np.zeros((0, 5))}


@case("softmax_bench")
def softmax_bench():
    from pytensor.tensor.special import log_softmax, softmax

    rng = np.random.default_rng(507)
    X = pt.dmatrix("X")
    return [X], [softmax(X, axis=1), log_softmax(X, axis=1), softmax(X, axis=0)], {"X": rng.normal(size=(37, 91)) * 3}


@case("elemwise_axis_reduce")
def elemwise_axis_reduce():
    # an Elemwise whose only consumer is a CAReduce over SOME axes (row / column sums of a fused
    # expression): sum_j exp(a_ij - m_i), column sums of squares, a max over a product
    rng = np.random.default_rng(508)
    A, m, r = pt.dmatrix("A"), pt.dvector("m"), pt.dvector("r")
    T = pt.dtensor3("T")
    outs = [pt.exp(A - m[:, None]).sum(axis=1), (A * A).sum(axis=0), (A * r[None, :]).max(axis=1), pt.log1p(abs(T)).sum(axis=(0, 2)),
            (T * T).sum(axis=1), pt.exp(T.transpose(2, 0, 1)).sum(axis=0)]
    return [A, m, r, T], outs, {"A": rng.normal(size=(41, 67)), "m": rng.normal(size=41), "r": rng.normal(size=67), "T": rng.normal(size=(7, 11, 13))}


@case("wide_200")
def wide_200():
    # north_star's literal target graph at a small N (IR is shape-agnostic): config #4 + 48 likelihood terms
    from pytensor_amd import configs
    from ref_graphs import build_wide200

    vals = configs.wide200_inputs(N=257, K=16, G=8)
    ins, outs = build_wide200(vals)
    return ins, outs, vals


@case("linalg_contention")
def linalg_contention():
    # the three persistent / cooperative kernels in one graph (task-graph Cholesky, vector triangular solve, LU panel):
    # tests/test_gpu_two_procs.py runs it at n = 2048 / 4096 / 1024 from two processes on ONE device
    from pytensor.tensor.linalg import cholesky, solve_triangular
    from pytensor.tensor.nlinalg import det

    rng = np.random.default_rng(512)
    S, Tm, M = pt.dmatrix("S"), pt.dmatrix("Tm"), pt.dmatrix("M")
    b = pt.dvector("b")
    n = 9
    A = rng.normal(size=(n, n + 3))
    return [S, Tm, b, M], [cholesky(S), solve_triangular(Tm, b, lower=True), det(M)], {
        "S": A @ A.T / n + np.eye(n), "Tm": np.tril(rng.normal(size=(n, n))) + 3 * np.eye(n), "b": rng.normal(size=n), "M": rng.normal(size=(n, n)) + 2 * np.eye(n)}


@case("c4_hier_f32", rtol=2e-4)
def c4_hier_f32():
    # config #4's model as a floatX=float32 user builds it: Gemv / Elemwise / scatter in float32 (the one-pass kernel's
    # float32 instance, dispatch/fused.py); the reference's float32 sums over 3000 terms set the tolerance
    from pytensor_amd import configs
    from ref_graphs import build_c4

    v = configs.c4_inputs(N=3000, K=32, G=16)
    vals = {k: (np.asarray(a, dtype="float32") if np.asarray(a).dtype.kind == "f" else a) for k, a in v.items()}
    ins, outs = build_c4(vals, dtype="float32")
    return ins, outs, vals
