"""GPU: the dense decompositions of csrc/decomp.hip against the calls the reference makes
(QR.perform: LAPACK geqrf/orgqr, SVD.perform: np.linalg.svd, MatrixPinv: np.linalg.pinv, Lstsq:
np.linalg.lstsq, the tridiagonal pair: gttrf/gttrs) on shapes the golden cases do not reach:
degenerate sizes, rank deficiency, reflectors longer than the LDS stage, batches through Blockwise."""
import numpy as np
import pytest
import scipy.linalg

import np_graph
from pytensor_amd.ir import Graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def one_node(op, params, in_specs, out_specs, consts=()):
    g = Graph(name=f"one_{op}")
    ins = [g.new_var(dt, (None,) * nd) for dt, nd in in_specs]
    cs = [g.new_var(str(c.dtype), c.shape, const=c) for c in consts]
    outs = [g.new_var(dt, (None,) * nd) for dt, nd in out_specs]
    g.add_node(op, params, [*ins, *cs], outs)
    g.inputs, g.outputs = ins, outs
    return g


def run(g, *vals):
    from pytensor_amd.executor import HipExecutable

    got = HipExecutable(g)(*vals)
    want = np_graph.run_graph(g, list(vals))
    return got, want


QR_OUT = {"full": 2, "economic": 2, "r": 1, "raw": 3}


@pytest.mark.parametrize("shape", [(1, 1), (1, 5), (5, 1), (64, 64), (130, 67), (67, 130), (300, 40), (5000, 3)])
@pytest.mark.parametrize("mode", ["full", "economic", "r", "raw"])
def test_qr_matches_lapack(hip, shape, mode):
    if mode == "full" and shape[0] > 1000:
        pytest.skip("m x m Q of a 5000-row matrix: nothing the other modes do not cover")
    rng = np.random.default_rng(sum(shape))
    x = rng.normal(size=shape)
    specs = [("float64", 1 if (mode == "raw" and k == 1) else 2) for k in range(QR_OUT[mode])]
    got, want = run(one_node("QR", {"mode": mode}, [("float64", 2)], specs), x)
    scale = np.abs(x).max()
    for k, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and a.dtype == b.dtype
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13 * scale * max(shape), err_msg=f"{mode} out{k}")
    if mode in ("full", "economic"):
        Q, R = got
        np.testing.assert_allclose(Q @ R, x, atol=1e-13 * scale * max(shape))
        np.testing.assert_allclose(Q.T @ Q, np.eye(Q.shape[1]), atol=1e-13 * max(shape))


def test_qr_zero_columns_and_float32(hip):
    x = np.random.default_rng(0).normal(size=(9, 6))
    x[:, 2] = 0.0
    x[3:, 4] = 0.0  # nothing below the diagonal in column 4 after elimination? (H = I only if exactly zero)
    got, want = run(one_node("QR", {"mode": "economic"}, [("float64", 2)], [("float64", 2)] * 2), x)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-14)
    z = np.zeros((4, 3))
    got, want = run(one_node("QR", {"mode": "raw"}, [("float64", 2)], [("float64", 2), ("float64", 1), ("float64", 2)]), z)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)  # tau = 0, H = I
    xf = np.random.default_rng(1).normal(size=(40, 17)).astype("float32")
    got, want = run(one_node("QR", {"mode": "full"}, [("float32", 2)], [("float32", 2)] * 2), xf)
    for a, b in zip(got, want):
        assert a.dtype == np.float32
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=3e-6)


def _check_svd(x, U, s, Vt, full):
    m, n = x.shape
    k = min(m, n)
    s_ref = np.linalg.svd(x, compute_uv=False)
    tol = 1e-12 if x.dtype == np.float64 else 2e-5
    np.testing.assert_allclose(s, s_ref, rtol=tol, atol=tol * (s_ref[0] if k else 0.0) * 1e-3 + (1e-300))
    assert U.shape == ((m, m) if full else (m, k)) and Vt.shape == ((n, n) if full else (k, n))
    np.testing.assert_allclose((U[:, :k] * s) @ Vt[:k], x, atol=50 * tol * max(np.abs(x).max(), 1e-300))
    np.testing.assert_allclose(U.T @ U, np.eye(U.shape[1]), atol=50 * tol)
    np.testing.assert_allclose(Vt @ Vt.T, np.eye(Vt.shape[0]), atol=50 * tol)


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (7, 1), (2, 2), (33, 33), (100, 60), (60, 100), (257, 31)])
@pytest.mark.parametrize("full", [False, True])
def test_svd_properties(hip, shape, full):
    from pytensor_amd.executor import HipExecutable

    x = np.random.default_rng(sum(shape) + full).normal(size=shape)
    g = one_node("SVD", {"full_matrices": full, "compute_uv": True}, [("float64", 2)], [("float64", 2), ("float64", 1), ("float64", 2)])
    U, s, Vt = HipExecutable(g)(x)
    _check_svd(x, U, s, Vt, full)
    g = one_node("SVD", {"full_matrices": full, "compute_uv": False}, [("float64", 2)], [("float64", 1)])
    (s2,) = HipExecutable(g)(x)
    np.testing.assert_allclose(s2, s, rtol=1e-13)


def test_svd_rank_deficient_graded_and_float32(hip):
    from pytensor_amd.executor import HipExecutable

    rng = np.random.default_rng(5)
    g = one_node("SVD", {"full_matrices": False, "compute_uv": True}, [("float64", 2)], [("float64", 2), ("float64", 1), ("float64", 2)])
    # rank 3 of 8: the null directions are completed orthonormally
    x = rng.normal(size=(12, 3)) @ rng.normal(size=(3, 8))
    U, s, Vt = HipExecutable(g)(x)
    assert (s[3:] < 1e-13 * s[0]).all()
    np.testing.assert_allclose(s[:3], np.linalg.svd(x, compute_uv=False)[:3], rtol=1e-12)
    np.testing.assert_allclose(U.T @ U, np.eye(8), atol=1e-12)
    np.testing.assert_allclose(Vt @ Vt.T, np.eye(8), atol=1e-12)
    np.testing.assert_allclose((U * s) @ Vt, x, atol=1e-12)
    # the zero matrix: s = 0, orthonormal factors
    U, s, Vt = HipExecutable(g)(np.zeros((5, 4)))
    assert np.array_equal(s, np.zeros(4))
    np.testing.assert_allclose(U.T @ U, np.eye(4), atol=1e-14)
    np.testing.assert_allclose(Vt @ Vt.T, np.eye(4), atol=1e-14)
    # singular values over 12 orders of magnitude: one-sided Jacobi keeps relative accuracy
    Q1, _ = np.linalg.qr(rng.normal(size=(20, 20)))
    Q2, _ = np.linalg.qr(rng.normal(size=(20, 20)))
    sv = np.logspace(0, -12, 20)
    U, s, Vt = HipExecutable(g)((Q1 * sv) @ Q2)
    np.testing.assert_allclose(s, sv, rtol=1e-3, atol=1e-15)  # (the matrix product itself perturbs by eps * s_max)
    np.testing.assert_allclose(s[:8], sv[:8], rtol=1e-8)
    xf = rng.normal(size=(30, 18)).astype("float32")
    gf = one_node("SVD", {"full_matrices": True, "compute_uv": True}, [("float32", 2)], [("float32", 2), ("float32", 1), ("float32", 2)])
    U, s, Vt = HipExecutable(gf)(xf)
    assert U.dtype == s.dtype == Vt.dtype == np.float32
    _check_svd(xf, U, s, Vt, True)


def test_pinv_and_lstsq(hip):
    rng = np.random.default_rng(6)
    gp = one_node("MatrixPinv", {"hermitian": False}, [("float64", 2)], [("float64", 2)])
    for shape in [(1, 1), (9, 4), (4, 9), (30, 30)]:
        x = rng.normal(size=shape)
        (got,), (want,) = run(gp, x)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-13 * np.abs(want).max())
    x = rng.normal(size=(7, 2)) @ rng.normal(size=(2, 5))  # rank 2
    (got,), (want,) = run(gp, x)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
    outs = [("float64", 2), ("float64", 1), ("int32", 0), ("float64", 1)]
    for a_shape, b_shape, rcond in [((12, 5), (12, 3), -1.0), ((5, 12), (5, 3), -1.0), ((12, 5), (12,), 1e-2), ((6, 6), (6, 2), -1.0)]:
        a, b = rng.normal(size=a_shape), rng.normal(size=b_shape)
        g = one_node("Lstsq", {}, [("float64", 2), ("float64", len(b_shape))], [(outs[0][0], len(b_shape)), *outs[1:]], consts=[np.asarray(rcond)])
        got, want = run(g, a, b)
        for k, (u, v) in enumerate(zip(got, want)):
            assert u.shape == v.shape and u.dtype == v.dtype, (a_shape, k, u.shape, v.shape, u.dtype, v.dtype)
            np.testing.assert_allclose(u, v, rtol=1e-10, atol=1e-13, err_msg=f"{a_shape} out{k}")
    # rank-deficient: no residuals, rank < n
    a = rng.normal(size=(10, 2)) @ rng.normal(size=(2, 4))
    g = one_node("Lstsq", {}, [("float64", 2), ("float64", 2)], outs, consts=[np.asarray(1e-10)])
    got, want = run(g, a, rng.normal(size=(10, 2)))
    assert int(got[2]) == int(want[2]) == 2 and got[1].shape == want[1].shape == (0,)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("n", [3, 4, 64, 1000])  # (scipy's gttrf wrapper, which the reference calls, needs n >= 3)
def test_tridiagonal_factor_and_solves_are_lapack_bit_for_bit(hip, n):
    rng = np.random.default_rng(n)
    dl, du = rng.normal(size=n - 1), rng.normal(size=n - 1)
    d = rng.normal(size=n) * np.where(rng.random(n) < 0.4, 0.05, 2.0)
    gf = one_node("LUFactorTridiagonal", {}, [("float64", 1)] * 3, [("float64", 1)] * 4 + [("int32", 1)])
    got, want = run(gf, dl, d, du)
    for k, (a, b) in enumerate(zip(got, want)):
        assert a.dtype == b.dtype and np.array_equal(a, b), k
    for b_ndim, trans in [(1, False), (2, True), (2, False), (1, True)]:
        b = rng.normal(size=(n,) if b_ndim == 1 else (n, 4))
        gs = one_node("SolveLUFactorTridiagonal", {"b_ndim": b_ndim, "transposed": trans},
                      [("float64", 1)] * 4 + [("int32", 1), ("float64", b_ndim)], [("float64", b_ndim)])
        (x,), (xw,) = run(gs, *want, b)
        assert np.array_equal(x, xw), (b_ndim, trans)
    if True:
        A = np.diag(d) + np.diag(dl, -1) + np.diag(du, 1) + np.triu(rng.normal(size=(n, n)), 2)  # (entries off the band are ignored)
        b = rng.normal(size=(n, 2))
        g = one_node("Solve", {"assume_a": "tridiagonal", "lower": False, "b_ndim": 2}, [("float64", 2), ("float64", 2)], [("float64", 2)])
        from pytensor_amd.executor import HipExecutable

        (x,) = HipExecutable(g)(A, b)
        gttrf, gttrs = scipy.linalg.get_lapack_funcs(("gttrf", "gttrs"), dtype="float64")
        f = gttrf(dl, d, du)
        assert np.array_equal(x, gttrs(*f[:5], b)[0])


def test_blockwise_batches_and_small_ops(hip):
    from pytensor_amd.executor import HipExecutable

    rng = np.random.default_rng(8)
    x = rng.normal(size=(3, 2, 9, 5))
    g = one_node("Blockwise", {"core_op": "QR", "core_params": {"mode": "economic"}, "signature": "(m,n)->(m,k),(k,n)"},
                 [("float64", 4)], [("float64", 4)] * 2)
    Q, R = HipExecutable(g)(x)
    assert Q.shape == (3, 2, 9, 5) and R.shape == (3, 2, 5, 5)
    np.testing.assert_allclose(Q @ R, x, atol=1e-13)
    g = one_node("Blockwise", {"core_op": "SVD", "core_params": {"full_matrices": False, "compute_uv": False}, "signature": "(m,n)->(k)"},
                 [("float64", 4)], [("float64", 3)])
    (s,) = HipExecutable(g)(x)
    np.testing.assert_allclose(s, np.linalg.svd(x, compute_uv=False), rtol=1e-12)
    # the generalised symmetric problem over a batch (Blockwise loops the items)
    Ab = rng.normal(size=(3, 7, 7))
    Ab = Ab + Ab.transpose(0, 2, 1)
    Qb = rng.normal(size=(3, 7, 7))
    Bb = Qb @ Qb.transpose(0, 2, 1) / 7 + np.eye(7)
    g = one_node("Blockwise", {"core_op": "Eigh", "core_params": {"lower": True}, "signature": "(m,m),(m,m)->(m),(m,m)"},
                 [("float64", 3), ("float64", 3)], [("float64", 2), ("float64", 3)])
    w, v = HipExecutable(g)(Ab, Bb)
    import scipy.linalg

    for k in range(3):
        np.testing.assert_allclose(w[k], scipy.linalg.eigh(Ab[k], Bb[k], eigvals_only=True), rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(Ab[k] @ v[k], Bb[k] @ v[k] * w[k], atol=1e-11)
    # Eigvalsh / TensorInv / TensorSolve / BlockDiagonal against the oracle restatement
    M = rng.normal(size=(14, 14))
    for g, vals in [
        (one_node("Eigvalsh", {"lower": True}, [("float64", 2)], [("float64", 1)]), [M + M.T]),
        (one_node("TensorInv", {"ind": 1}, [("float64", 3)], [("float64", 3)]), [rng.normal(size=(6, 2, 3))]),
        (one_node("TensorSolve", {"axes": [0]}, [("float64", 3), ("float64", 1)], [("float64", 2)]), [rng.normal(size=(2, 6, 3)), rng.normal(size=6)]),
        (one_node("BlockDiagonal", {"dtype": "float64"}, [("float64", 2), ("int64", 2), ("float32", 2)], [("float64", 2)]),
         [rng.normal(size=(2, 3)), np.arange(4).reshape(4, 1), np.zeros((0, 2), dtype="float32")]),
    ]:
        got, want = run(g, *vals)
        for a, b in zip(got, want):
            assert a.shape == b.shape and a.dtype == b.dtype
            np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)
