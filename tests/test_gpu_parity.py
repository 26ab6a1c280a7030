"""GPU: the HIP path (through the C-ABI) against the reference's golden vectors and the oracle."""
import numpy as np
import pytest

import np_graph
from util import assert_parity, golden_cases, load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


@pytest.mark.parametrize("name", golden_cases())
def test_golden_case(hip, name):
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case(name)
    exe = HipExecutable(g)
    out = exe(*ins)
    assert len(out) == len(cvm)
    for k, (a, b) in enumerate(zip(out, cvm)):
        assert isinstance(a, np.ndarray)
        # element-wise at north_star's rtol, no atol; exceptions per output in tests/tolerances.json
        assert_parity(a, b, None, f"{name} out{k} (hip vs reference C linker)", case=name, k=k, py=py[k])
    # second call: cached kernels, pooled buffers — same answer, bit for bit
    out2 = exe(*ins)
    for a, b in zip(out, out2):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", golden_cases())
def test_unfused_equals_fused(hip, name):
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case(name)
    a = HipExecutable(g, fuse=True)(*ins)
    b = HipExecutable(g, fuse=False)(*ins)
    for k, (x, y) in enumerate(zip(a, b)):
        assert_parity(x, y, None, f"{name} out{k} fused vs unfused", case=name, k=k, py=py[k], slack=2.0)


@pytest.mark.parametrize("dtype,n", [("float64", 1), ("float64", 7), ("float64", 64), ("float64", 128), ("float64", 141),
                                     ("float32", 33), ("float32", 200)])
def test_potrf_trsv_fused_kernel(hip, dtype, n):
    """The fused Cholesky + forward-substitution launch against LAPACK potrf/trtrs."""
    import scipy.linalg

    from pytensor_amd.device import DeviceArray

    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n + 3))
    S = (A @ A.T / n + np.eye(n)).astype(dtype)
    b = rng.normal(size=n).astype(dtype)
    dS, db = DeviceArray.from_host(S), DeviceArray.from_host(b)
    L, x = DeviceArray.empty((n, n), dtype), DeviceArray.empty((n,), dtype)
    lib = hip.lib()
    hip.check(lib.pthip_potrf_trsv(hip.np_dtype_code(dtype), 1, n, dS.ptr, db.ptr, L.ptr, x.ptr))
    Lr = scipy.linalg.cholesky(S, lower=True)
    xr = scipy.linalg.solve_triangular(Lr, b, lower=True)
    rtol = 1e-11 if dtype == "float64" else 2e-4
    np.testing.assert_allclose(L.to_host(), Lr, rtol=rtol, atol=rtol)
    np.testing.assert_allclose(x.to_host(), xr, rtol=rtol, atol=rtol)
    # not positive definite: the factor and the solve are NaN (reference on_error="nan", cholesky.py:76-83)
    S[n // 2, n // 2] = -1.0
    dS = DeviceArray.from_host(S)
    hip.check(lib.pthip_potrf_trsv(hip.np_dtype_code(dtype), 1, n, dS.ptr, db.ptr, L.ptr, x.ptr))
    assert np.isnan(L.to_host()).all() and np.isnan(x.to_host()).all()


@pytest.mark.parametrize("dtype,n,batch,kind", [("float64", 161, 1, "random"), ("float64", 200, 2, "random"), ("float64", 256, 1, "plusminus"),
                                                ("float64", 300, 1, "lowrank"), ("float64", 512, 1, "lowrank"), ("float64", 513, 1, "random"), ("float64", 1024, 1, "random"),
                                                ("float32", 320, 1, "random")])
def test_eigh_block_jacobi_beyond_one_cu(hip, dtype, n, batch, kind):
    """pthip_eigh for matrices beyond one CU's LDS (csrc/eigh.hip eigh_block_jacobi: 32-column blocks, batched
    64 x 64 subproblems, MFMA GEMM updates; any n) against LAPACK and the defining properties.  ``plusminus``:
    eigenvalues in +/- pairs (a one-sided method on the unshifted matrix cannot separate them); ``lowrank``: an
    exactly singular matrix (the null space still gets orthonormal vectors); n not a multiple of 64 exercises the
    decoupled padding; ``lower=False`` reads only the upper triangle.  Reference: Eigh.perform,
    pytensor/tensor/linalg/decomposition/eigen.py:177-195 (scipy.linalg.eigh: any n)."""
    from pytensor_amd.device import DeviceArray

    rng = np.random.default_rng(500 + n)
    if kind == "plusminus":
        Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
        lam = np.concatenate([np.arange(1, n // 2 + 1), -np.arange(1, n - n // 2 + 1)]).astype("float64")
        S = ((Q * lam) @ Q.T)[None]
    elif kind == "lowrank":
        B = rng.normal(size=(n, 40))
        S = (B @ B.T)[None]
    else:
        M = rng.normal(size=(batch, n, n))
        S = (M + M.transpose(0, 2, 1)) / 2
    S = ((S + S.transpose(0, 2, 1)) / 2).astype(dtype)
    lower = n % 2 == 0
    junk = rng.normal(size=(n, n)).astype(dtype) * 1e3
    keep = np.tril(S) + np.triu(junk, 1) if lower else np.triu(S) + np.tril(junk, -1)
    dS = DeviceArray.from_host(np.ascontiguousarray(keep))
    w, v = DeviceArray.empty((batch, n), dtype), DeviceArray.empty((batch, n, n), dtype)
    import ctypes

    hip.check(hip.lib().pthip_eigh(hip.np_dtype_code(dtype), batch, n, int(lower), dS.ptr, w.ptr, v.ptr))
    w, v = w.to_host(), v.to_host()
    st = ctypes.c_int(-1)
    hip.check(hip.lib().pthip_check_status(ctypes.byref(st)))
    assert st.value == 0, f"device status word {st.value} (bit 3: the iteration did not converge)"
    tol = 1e-12 if dtype == "float64" else 5e-6
    for b in range(batch):
        S64 = S[b].astype("float64")
        wr = np.linalg.eigvalsh(S64)
        scale = max(1.0, float(np.abs(wr).max()))
        np.testing.assert_allclose(w[b], wr, rtol=0, atol=tol * scale * n)
        assert (np.diff(w[b]) >= 0).all()
        V = v[b].astype("float64")
        np.testing.assert_allclose(V.T @ V, np.eye(n), atol=tol * n)
        np.testing.assert_allclose(S64 @ V, V * w[b][None, :], atol=tol * scale * n)


@pytest.mark.parametrize("dtype,n,batch", [("float64", 1, 1), ("float64", 2, 3), ("float64", 5, 2), ("float64", 64, 2), ("float64", 96, 1),
                                           ("float64", 100, 1), ("float64", 141, 2), ("float32", 33, 2), ("float32", 150, 1)])
def test_eigh_jacobi_kernel(hip, dtype, n, batch):
    """pthip_eigh against LAPACK (eigenvalues) and the defining properties (A V = V diag(w),
    V^T V = I, ascending order) — the sizes walk the three storage layouts (A and V in LDS; V in
    the global scratch; both), odd n exercises the padded round-robin."""
    from pytensor_amd.device import DeviceArray

    rng = np.random.default_rng(100 + n)
    M = rng.normal(size=(batch, n, n))
    S = ((M + M.transpose(0, 2, 1)) / 2).astype(dtype)
    if n > 3:
        S[0, 1, 1] = S[0, 2, 2] = 0.75  # a repeated diagonal entry, an exactly zero coupling
        S[0, 1, 2] = S[0, 2, 1] = 0.0
    junk = np.triu(rng.normal(size=(n, n)), 1).astype(dtype)  # the upper triangle must not be read
    dS = DeviceArray.from_host(np.ascontiguousarray(np.tril(S) + junk))
    w, v = DeviceArray.empty((batch, n), dtype), DeviceArray.empty((batch, n, n), dtype)
    import ctypes

    hip.check(hip.lib().pthip_eigh(hip.np_dtype_code(dtype), batch, n, 1, dS.ptr, w.ptr, v.ptr))
    w, v = w.to_host(), v.to_host()
    st = ctypes.c_int(-1)
    hip.check(hip.lib().pthip_check_status(ctypes.byref(st)))
    assert st.value == 0, f"device status word {st.value}"
    tol = 1e-12 if dtype == "float64" else 5e-6
    for b in range(batch):
        wr = np.linalg.eigvalsh(S[b].astype("float64"))
        scale = max(1.0, float(np.abs(wr).max()))
        np.testing.assert_allclose(w[b], wr, rtol=0, atol=tol * scale * n)
        assert (np.diff(w[b]) >= 0).all()
        V = v[b].astype("float64")
        np.testing.assert_allclose(V.T @ V, np.eye(n), atol=tol * n)
        np.testing.assert_allclose(S[b].astype("float64") @ V, V * w[b][None, :], atol=tol * scale * n)
