"""GPU: the round-6 log-sum-exp / column-softmax kernels (csrc/softmax.hip) through the C-ABI against SciPy — what the
reference's ``perform`` of Softmax / LogSoftmax is (pytensor/tensor/special.py:26-120: scipy.special.softmax /
log_softmax) and what its logsumexp graph evaluates to (math.py logsumexp; tests/benchmarks/test_logsumexp.py).
fp64 rtol 1e-12 on softmax values, 1e-12 relative + 4 eps absolute on log-domain values (a log-sum-exp near 0 is a
difference of two O(1) numbers); fp32 1e-5."""
import numpy as np
import pytest
import scipy.special as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _tol(dt, log_domain, scale=1.0):
    if dt == "float64":
        return 1e-12, (4 * 2.2e-16 * max(1.0, scale) if log_domain else 0.0)
    return 1e-5, (4 * 1.2e-7 * max(1.0, scale) if log_domain else 0.0)


def _lse_rows(hip, x):
    from pytensor_amd.device import DeviceArray

    rows, cols = x.shape
    dx, out = DeviceArray.from_host(x), DeviceArray.empty((rows,), x.dtype)
    hip.check(hip.lib().pthip_logsumexp_rows(hip.np_dtype_code(x.dtype), rows, cols, dx.ptr, out.ptr))
    return out.to_host()


def _cols(hip, x, what):
    from pytensor_amd.device import DeviceArray

    b, R, Cc = x.shape
    lib, code = hip.lib(), hip.np_dtype_code(x.dtype)
    n = int(lib.pthip_colstat_workspace(code, b, R, Cc))
    ws = DeviceArray.empty((n,), "uint8")
    dx = DeviceArray.from_host(x)
    if what == "lse":
        out = DeviceArray.empty((b, Cc), x.dtype)
        hip.check(lib.pthip_logsumexp_cols(code, b, R, Cc, dx.ptr, out.ptr, ws.ptr, n))
    else:
        out = DeviceArray.empty(x.shape, x.dtype)
        hip.check(lib.pthip_softmax_cols(code, int(what == "log_softmax"), b, R, Cc, dx.ptr, out.ptr, ws.ptr, n))
    return out.to_host()


@pytest.mark.parametrize("dt", ["float64", "float32"])
@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 2), (1000, 10), (257, 16), (300, 17), (513, 31), (70000, 32), (5, 33), (129, 64), (1000, 100), (77, 256),
                                       (64, 1000), (33, 2048), (9, 2047), (1500, 4096), (300, 5000), (40, 10001), (7, 70000)])
def test_logsumexp_rows(hip, dt, rows, cols):
    rng = np.random.default_rng(rows * 131 + cols)
    x = (rng.normal(size=(rows, cols)) * 3 + rng.normal() * 50).astype(dt)
    got = _lse_rows(hip, x)
    want = sp.logsumexp(x.astype("float64"), axis=1)
    rtol, atol = _tol(dt, True, float(np.max(np.abs(x))))
    np.testing.assert_allclose(got, want.astype(dt), rtol=rtol, atol=atol)
    assert got.dtype == np.dtype(dt)


@pytest.mark.parametrize("dt", ["float64", "float32"])
@pytest.mark.parametrize("b,R,C", [(1, 2, 2), (1, 1000, 10), (1, 4099, 5), (1, 100003, 10), (3, 257, 7), (2, 64, 256), (1, 300, 257), (1, 513, 512), (1, 1031, 2050),
                                   (1, 8192, 128), (5, 33, 130), (1, 2, 100000), (1, 70000, 1), (2, 9, 3)])
def test_column_statistics_and_softmax(hip, dt, b, R, C):
    rng = np.random.default_rng(b * 7 + R * 13 + C)
    x = (rng.normal(size=(b, R, C)) * 3 + rng.normal(size=(b, 1, C)) * 30).astype(dt)
    x64 = x.astype("float64")
    rtol, atol = _tol(dt, True, float(np.max(np.abs(x))))
    np.testing.assert_allclose(_cols(hip, x, "lse"), sp.logsumexp(x64, axis=1).astype(dt), rtol=rtol, atol=atol)
    np.testing.assert_allclose(_cols(hip, x, "log_softmax"), sp.log_softmax(x64, axis=1).astype(dt), rtol=rtol, atol=atol)
    sm = _cols(hip, x, "softmax")
    r2 = 1e-12 if dt == "float64" else 1e-5
    np.testing.assert_allclose(sm, sp.softmax(x64, axis=1).astype(dt), rtol=r2, atol=0.0 if dt == "float64" else 1e-37)
    np.testing.assert_allclose(sm.astype("float64").sum(axis=1), 1.0, rtol=0, atol=(1e-13 if dt == "float64" else 2e-6) * max(1.0, np.sqrt(R)))


def test_shift_invariance_fp64(hip):
    """softmax(x + c) == softmax(x) for a large c: the normalisation subtracts the MAX (exact), not the log-sum-exp"""
    rng = np.random.default_rng(5)
    x = rng.normal(size=(1, 999, 12))
    a = _cols(hip, x, "softmax")
    b = _cols(hip, x + 1.0e6, "softmax")  # (x + 1e6 rounds x: compare with SciPy on the same rounded values)
    np.testing.assert_allclose(b, sp.softmax(x + 1.0e6, axis=1), rtol=1e-12)
    np.testing.assert_allclose(a, sp.softmax(x, axis=1), rtol=1e-12)


def test_non_finite_entries_follow_the_reference(hip):
    """the stabilised graph's semantics (shift = isinf(max) ? 0 : max): an all -inf slice -> -inf; a +inf term -> +inf;
    NaN propagates — per column / per row, the neighbours untouched"""
    x = np.random.default_rng(6).normal(size=(1, 600, 6))
    x[0, :, 1] = -np.inf
    x[0, 17, 2] = np.inf
    x[0, 500, 3] = np.nan
    x[0, 3, 4] = -np.inf
    with np.errstate(all="ignore"):
        want = sp.logsumexp(x, axis=1)
    got = _cols(hip, x, "lse")
    assert got[0, 1] == -np.inf and got[0, 2] == np.inf and np.isnan(got[0, 3])
    np.testing.assert_allclose(got[0, [0, 4, 5]], want[0, [0, 4, 5]], rtol=1e-12, atol=1e-15)
    with np.errstate(all="ignore"):
        sm, smw = _cols(hip, x, "softmax"), sp.softmax(x, axis=1)
    np.testing.assert_allclose(sm[0][:, [0, 4, 5]], smw[0][:, [0, 4, 5]], rtol=1e-12)
    assert np.isnan(sm[0][:, 1]).all() and np.isnan(sm[0][:, 3]).all()  # 0/0 and NaN, as exp(x - 0) / 0 in the reference
    rows = np.random.default_rng(7).normal(size=(300, 9))
    rows[5] = -np.inf
    rows[6, 2] = np.inf
    rows[7, 8] = np.nan
    got = _lse_rows(hip, rows)
    assert got[5] == -np.inf and got[6] == np.inf and np.isnan(got[7])
    keep = [k for k in range(300) if k not in (5, 6, 7)]
    np.testing.assert_allclose(got[keep], sp.logsumexp(rows[keep], axis=1), rtol=1e-12, atol=1e-15)
    wide = np.random.default_rng(8).normal(size=(40, 512))
    wide[3] = -np.inf
    wide[4, 100] = np.inf
    wide[5, 511] = np.nan
    got = _lse_rows(hip, wide)
    assert got[3] == -np.inf and got[4] == np.inf and np.isnan(got[5])


def test_results_are_reproducible_run_to_run(hip):
    x = np.random.default_rng(9).normal(size=(1, 50000, 10)) * 4
    a = _cols(hip, x, "softmax")
    l1 = _cols(hip, x, "lse")
    for _ in range(3):
        np.testing.assert_array_equal(_cols(hip, x, "softmax"), a)
        np.testing.assert_array_equal(_cols(hip, x, "lse"), l1)
