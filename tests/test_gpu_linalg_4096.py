"""GPU: parity of the persistent linalg kernels at the sizes the performance figures are quoted on —
n = 4096 for ``pthip_potrf`` (8 rounds of the task graph on 256 workgroups, where n = 2048 has 2), the
vector and the square triangular solve, ``pthip_getrf`` (pivots exact), and n = 8192 for the vector solve.

Reference: ``Cholesky.perform`` (pytensor/tensor/linalg/decomposition/cholesky.py:48-83),
``SolveTriangular.perform`` (solvers/triangular.py:32-71), ``Solve`` / ``LUFactor`` (solvers/general.py:17,
decomposition/lu.py:206) — LAPACK potrf / trtrs / getrf, which take any n.  Bounds are the ones of
tests/test_gpu_chol_blocked.py / test_gpu_trsm_blocked.py / test_gpu_lu_blocked.py (Higham, Accuracy and
Stability, Thms 10.3, 8.5, 9.3): entry-wise residuals ``c n eps |L||L^T|`` etc., and the distance to
LAPACK's answer at ``C n eps cond``.  The condition numbers are in the infinity norm from an explicit
inverse (LAPACK trtri / potri class work, seconds at these sizes) — an SVD of an 8192^2 matrix is not."""
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


def _cond_inf_tri(T, lower):
    import scipy.linalg

    Ti = scipy.linalg.solve_triangular(T, np.eye(T.shape[0]), lower=lower)
    return np.abs(T).sum(1).max() * np.abs(Ti).sum(1).max()


@pytest.mark.parametrize("lower", [True, False])
def test_cholesky_4096_matches_lapack(hip, lower):
    import scipy.linalg

    from pytensor_amd.device import DeviceArray

    n = 4096
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n + 5))
    S = A @ A.T / n + np.eye(n)
    dS, dL = DeviceArray.from_host(S), DeviceArray.empty(S.shape, S.dtype)
    call = lambda: hip.check(hip.lib().pthip_potrf(hip.np_dtype_code(S.dtype), int(lower), 1, n, dS.ptr, dL.ptr))
    call()
    got = dL.to_host()
    want = scipy.linalg.cholesky(S, lower=lower)
    eps = np.finfo("float64").eps
    other = np.triu(got, 1) if lower else np.tril(got, -1)
    assert not other.any()  # potrf clean=True
    L, Lw = (got, want) if lower else (got.T, want.T)
    w = np.linalg.eigvalsh(S)
    cond = w[-1] / w[0]
    C = 4.0
    assert np.max(np.abs(L - Lw)) <= C * n * eps * cond * np.max(np.abs(Lw)), (np.max(np.abs(L - Lw)), cond)
    resid = np.abs(L @ L.T - S)
    bound = C * n * eps * (np.abs(L) @ np.abs(L).T)
    assert (resid <= bound).all(), float(np.max(resid / bound))
    call()
    np.testing.assert_array_equal(got, dL.to_host())  # deterministic: 2080 tasks, any schedule, same bits


def test_cholesky_4096_failure_in_a_late_round_is_all_nan(hip):
    from pytensor_amd.device import DeviceArray

    n = 4096
    rng = np.random.default_rng(1)
    A = rng.normal(size=(n, n + 5))
    S = A @ A.T / n + np.eye(n)
    S[3900, 3900] = -1.0
    dS, dL = DeviceArray.from_host(S), DeviceArray.empty(S.shape, S.dtype)
    hip.check(hip.lib().pthip_potrf(hip.np_dtype_code(S.dtype), 1, 1, n, dS.ptr, dL.ptr))
    assert np.isnan(dL.to_host()).all()


def _tri(n, seed, lower):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n)) / np.sqrt(n)
    A = np.tril(A) if lower else np.triu(A)
    A[np.diag_indices(n)] = 1.0 + rng.uniform(0.5, 1.5, n)
    junk = (np.triu(rng.normal(size=(n, n)), 1) if lower else np.tril(rng.normal(size=(n, n)), -1)) * 1e3
    return A, A + junk


def _solve(hip, full, b, lower, trans):
    from pytensor_amd.device import DeviceArray

    n = full.shape[-1]
    nrhs = 1 if b.ndim == 1 else b.shape[-1]
    dT, db = DeviceArray.from_host(full), DeviceArray.from_host(np.ascontiguousarray(b))
    out = DeviceArray.empty(b.shape, b.dtype)
    hip.check(hip.lib().pthip_trsm(hip.np_dtype_code(b.dtype), int(lower), int(trans), 0, 1, n, nrhs,
                                   dT.ptr, n * n, n, 1, db.ptr, n * nrhs, out.ptr))
    return out.to_host()


@pytest.mark.parametrize("n,nrhs", [(4096, None), (8192, None), (4096, 4096)])
@pytest.mark.parametrize("lower,trans", [(True, False), (False, False), (True, True)])
def test_triangular_solve_4096_8192_matches_lapack(hip, n, nrhs, lower, trans):
    import scipy.linalg

    if n == 8192 and trans:
        pytest.skip("the transposed walk is covered at n = 4096")
    Tm, full = _tri(n, n + int(lower), lower)
    rng = np.random.default_rng(n + 1)
    b = rng.normal(size=(n,) if nrhs is None else (n, nrhs))
    got = _solve(hip, full, b, lower, trans)
    want = scipy.linalg.solve_triangular(Tm, b, lower=lower, trans=int(trans))
    eps = np.finfo("float64").eps
    op = Tm.T if trans else Tm
    resid = np.abs(op @ got - b)
    bound = 8.0 * n * eps * (np.abs(op) @ np.abs(got)) + 1e-300
    assert (resid <= bound).all(), float(np.max(resid / bound))
    cond = _cond_inf_tri(op, lower != trans)
    assert np.max(np.abs(got - want)) <= 8.0 * n * eps * cond * np.max(np.abs(want)), (np.max(np.abs(got - want)), cond)
    np.testing.assert_array_equal(got, _solve(hip, full, b, lower, trans))  # deterministic


def test_lu_4096_same_pivots_as_lapack(hip):
    import scipy.linalg

    from pytensor_amd.device import DeviceArray

    n = 4096
    A = np.random.default_rng(n).normal(size=(n, n))
    dA, LU = DeviceArray.from_host(A), DeviceArray.empty(A.shape, A.dtype)
    perm, sign, logabs = DeviceArray.empty((1, n), "int64"), DeviceArray.empty((1,), A.dtype), DeviceArray.empty((1,), A.dtype)
    hip.check(hip.lib().pthip_getrf(hip.np_dtype_code(A.dtype), 1, n, dA.ptr, LU.ptr, perm.ptr, sign.ptr, logabs.ptr, 0))
    lu, p = LU.to_host(), perm.to_host()[0]
    lu_ref, piv = scipy.linalg.lu_factor(A)
    pref = np.arange(n)
    for k, q in enumerate(piv):
        pref[[k, q]] = pref[[q, k]]
    np.testing.assert_array_equal(p, pref)  # the same rows, in the same order, as LAPACK
    eps = np.finfo("float64").eps
    L = np.tril(lu, -1) + np.eye(n)
    U = np.triu(lu)
    resid = np.abs(L @ U - A[p])
    bound = 4.0 * n * eps * (np.abs(L) @ np.abs(U))
    assert (resid <= bound).all(), float(np.max(resid / bound))
    s_ref, la_ref = np.linalg.slogdet(A)
    assert sign.to_host()[0] == s_ref
    assert abs(logabs.to_host()[0] - la_ref) <= 1e-9 * abs(la_ref)
