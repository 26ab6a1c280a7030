"""GPU: triangular solves beyond one CU's LDS (n > 141 fp64 / 200 fp32) through the C-ABI ``pthip_trsm``
— the persistent row-block kernel for a few right-hand sides (csrc/linalg.hip ``trsv_dag_kernel``) and
the blocked solve with MFMA GEMM updates for many (``trsm_blocked``).

Reference: ``SolveTriangular.perform`` (pytensor/tensor/linalg/solvers/triangular.py:32-71: LAPACK
``trtrs``; ``trans`` / ``lower`` / ``unit_diagonal``).  Bound: substitution is backward stable row-wise,
``|T x - b| <= c n eps |T||x|`` (Higham, Accuracy and Stability, Thm 8.5) — asserted entry-wise with c
stated; the solution itself is compared with LAPACK's at ``C n eps cond(T)``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _tri(n, dtype, seed, lower, unit=False):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n)) / np.sqrt(n)
    A = np.tril(A) if lower else np.triu(A)
    A[np.diag_indices(n)] = 1.0 + rng.uniform(0.5, 1.5, n)
    full = A + (np.triu(rng.normal(size=(n, n)), 1) if lower else np.tril(rng.normal(size=(n, n)), -1)) * 1e3  # junk in the other triangle
    return A.astype(dtype), full.astype(dtype)


def _solve(hip, Tfull, b, lower, trans=False, unit=False):
    from pytensor_amd.device import DeviceArray

    n = Tfull.shape[-1]
    batch = Tfull.shape[0] if Tfull.ndim == 3 else 1
    nrhs = 1 if b.ndim == Tfull.ndim - 1 else b.shape[-1]
    dT = DeviceArray.from_host(np.ascontiguousarray(Tfull))
    db = DeviceArray.from_host(np.ascontiguousarray(b))
    out = DeviceArray.empty(b.shape, b.dtype)
    hip.check(hip.lib().pthip_trsm(hip.np_dtype_code(b.dtype), int(lower), int(trans), int(unit), batch, n, nrhs,
                                   dT.ptr, n * n, n, 1, db.ptr, n * nrhs, out.ptr))
    return out.to_host()


@pytest.mark.parametrize("dtype,n", [("float64", 142), ("float64", 300), ("float64", 1000), ("float64", 2048), ("float32", 201), ("float32", 777)])
@pytest.mark.parametrize("lower", [True, False])
@pytest.mark.parametrize("trans", [False, True])
@pytest.mark.parametrize("nrhs", [None, 3, 7, 40, 300])
def test_large_triangular_solve_matches_lapack(hip, dtype, n, lower, trans, nrhs):
    import scipy.linalg

    if n >= 1000 and nrhs in (7, 40):
        pytest.skip("covered at the smaller sizes")
    Tm, full = _tri(n, dtype, n, lower)
    rng = np.random.default_rng(n + 1)
    b = rng.normal(size=(n,) if nrhs is None else (n, nrhs)).astype(dtype)
    got = _solve(hip, full, b, lower, trans)
    want = scipy.linalg.solve_triangular(Tm, b, lower=lower, trans=int(trans))
    eps = np.finfo(dtype).eps
    op = Tm.T if trans else Tm
    op64, g64 = op.astype("float64"), got.astype("float64")
    resid = np.abs(op64 @ g64 - b)
    bound = 8.0 * n * eps * (np.abs(op64) @ np.abs(g64)) + 1e-300
    assert (resid <= bound).all(), float(np.max(resid / bound))
    cond = np.linalg.cond(op64)
    assert np.max(np.abs(got - want)) <= 8.0 * n * eps * cond * np.max(np.abs(want))
    np.testing.assert_array_equal(got, _solve(hip, full, b, lower, trans))  # deterministic


@pytest.mark.parametrize("nrhs", [None, 50])
def test_large_unit_diagonal_and_batch(hip, nrhs):
    import scipy.linalg

    n = 260
    Ts, fulls = zip(*[_tri(n, "float64", 5 + k, True) for k in range(3)])
    rng = np.random.default_rng(9)
    b = rng.normal(size=(3, n) if nrhs is None else (3, n, nrhs))
    got = _solve(hip, np.stack(fulls), b, True, unit=True)
    for k in range(3):
        want = scipy.linalg.solve_triangular(Ts[k], b[k], lower=True, unit_diagonal=True)
        np.testing.assert_allclose(got[k], want, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("nrhs", [None, 64])
@pytest.mark.parametrize("bad", [0, 150, 299])
def test_zero_pivot_poisons_the_result(hip, nrhs, bad):
    n = 300
    _, full = _tri(n, "float64", 3, True)
    full[bad, bad] = 0.0
    b = np.ones((n,) if nrhs is None else (n, nrhs))
    assert np.isnan(_solve(hip, full, b, True)).all()


def test_gp_solves_n2048_through_the_graph_dispatch(hip):
    """``L^-1 y`` then ``L^-T z`` with a vector, the two solves of a GP marginal likelihood, on a Cholesky
    factor of the size the blocked kernels exist for — against LAPACK at rtol 1e-10."""
    import scipy.linalg

    n = 2048
    rng = np.random.default_rng(0)
    A = rng.normal(size=(n, n + 8))
    S = A @ A.T / n + np.eye(n)
    L = np.linalg.cholesky(S)
    y = rng.normal(size=n)
    z = _solve(hip, L, y, True)
    w = _solve(hip, L, z, True, trans=True)
    np.testing.assert_allclose(z, scipy.linalg.solve_triangular(L, y, lower=True), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(w, scipy.linalg.cho_solve((L, True), y), rtol=1e-9, atol=1e-12)
