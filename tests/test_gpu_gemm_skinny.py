"""GPU: tall-and-skinny products (csrc/gemm_skinny.hip) through the C-ABI ``pthip_gemm`` — the shapes of a multi-response
regression's two ``Gemm`` nodes (oracle/ref_graphs.build_wide200_gemm; reference ops pytensor/tensor/blas/gemm.py:76):
forward ``beta*C + alpha * X @ B`` with ~1e4..1e5 rows and <= 16 columns, backward ``beta*C + alpha * X.T @ W`` with X.T a
transposed view — against a float64 NumPy product with the dot-product bound ``|err| <= rtol*|want| + c*eps*(|A| @ |B|)``
(c = 8), and against the MFMA kernels on the same operands (``PTHIP_GEMM_SKINNY`` cannot be switched inside a process: the
comparison is with NumPy only; the executor-level golden ``c4_gemm_multiresponse`` / ``wide_200_gemm`` cover both paths'
agreement with the reference C linker)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
C_SUM = 8.0


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _dev_view(hip, a):
    """upload the BASE of a view and return (DeviceArray of the base, pointer of the view's first element, element strides)"""
    from pytensor_amd.device import DeviceArray

    base = a
    while isinstance(base.base, np.ndarray):
        base = base.base
    d = DeviceArray.from_host(np.ascontiguousarray(base)) if base.flags.c_contiguous else None
    assert d is not None
    off = a.__array_interface__["data"][0] - base.__array_interface__["data"][0]
    return d, d.ptr + off, tuple(s // a.itemsize for s in a.strides)


def _gemm(hip, A, B, alpha=1.0, beta=0.0, Cm=None):
    from pytensor_amd.device import DeviceArray

    M, K = A.shape
    N = B.shape[1]
    dA, pA, sA = _dev_view(hip, A)
    dB, pB, sB = _dev_view(hip, B)
    out = DeviceArray.empty((M, N), A.dtype)
    if Cm is not None:
        dC, pC, sC = _dev_view(hip, Cm)
        sC0 = 0 if Cm.shape[0] == 1 and M != 1 else sC[0]
        sC1 = 0 if Cm.shape[1] == 1 and N != 1 else sC[1]
    else:
        pC, sC0, sC1 = None, 0, 0
    hip.check(hip.lib().pthip_gemm(hip.np_dtype_code(A.dtype), 1, M, N, K, float(alpha), pA, 0, sA[0], sA[1], pB, 0, sB[0], sB[1], float(beta), pC, 0, sC0, sC1,
                                   out.ptr))
    return out.to_host()


def _check(got, A, B, alpha, beta, Cm, dt):
    eps = float(np.finfo(dt).eps)
    A64, B64 = A.astype("float64"), B.astype("float64")
    want = alpha * (A64 @ B64) + (beta * Cm.astype("float64") if Cm is not None else 0.0)
    bound = (1e-12 if dt == "float64" else 1e-5) * np.abs(want) + C_SUM * eps * (abs(alpha) * (np.abs(A64) @ np.abs(B64)) + (abs(beta) * np.abs(Cm) if Cm is not None else 0.0))
    assert got.dtype == np.dtype(dt) and got.shape == want.shape
    worst = float(np.max(np.abs(got.astype("float64") - want) / np.maximum(bound, 1e-300)))
    assert worst <= 1.0, f"|err| / bound = {worst}"


@pytest.mark.parametrize("dt", ["float64", "float32"])
@pytest.mark.parametrize("M,K,N,layout", [(10000, 128, 8, "c"), (8192, 1, 1, "c"), (20011, 37, 5, "wide"), (9000, 300, 16, "c"), (8200, 4096, 3, "c"),
                                          (70001, 16, 8, "c"), (12345, 129, 12, "wide_even"), (8192, 128, 1, "c")])
def test_forward_long_rows_times_narrow_matrix(hip, dt, M, K, N, layout):
    rng = np.random.default_rng(M + 3 * K + N)
    if layout == "c":
        A = rng.normal(size=(M, K)).astype(dt)
    else:  # a column block of a wider array: row pitch > K (odd pitch: the 8-byte / 4-byte load path)
        extra = 3 if layout == "wide" else 4
        A = rng.normal(size=(M, K + extra)).astype(dt)[:, 1 if layout == "wide" else 0:][:, :K]
    B = rng.normal(size=(K, N)).astype(dt)
    _check(_gemm(hip, A, B), A, B, 1.0, 0.0, None, dt)
    Cm = rng.normal(size=(M, N)).astype(dt)
    _check(_gemm(hip, A, B, alpha=-0.7, beta=1.3, Cm=Cm), A, B, -0.7, 1.3, Cm, dt)
    row = rng.normal(size=(1, N)).astype(dt)  # a broadcast row as C
    _check(_gemm(hip, A, B, alpha=2.0, beta=1.0, Cm=row), A, B, 2.0, 1.0, np.broadcast_to(row, (M, N)), dt)
    Bt = np.ascontiguousarray(B.T).T  # B stored transposed (column-major view): any strides are accepted for the short operand
    _check(_gemm(hip, A, Bt), A, B, 1.0, 0.0, None, dt)


@pytest.mark.parametrize("dt", ["float64", "float32"])
@pytest.mark.parametrize("rows,M,N,layout", [(20000, 128, 8, "c"), (8192, 1, 1, "c"), (30011, 37, 5, "wide"), (9001, 300, 16, "c"), (10000, 512, 8, "c"),
                                             (100003, 16, 2, "c"), (8199, 129, 9, "wide_even")])
def test_backward_transposed_long_operand(hip, dt, rows, M, N, layout):
    rng = np.random.default_rng(rows + 3 * M + N)
    if layout == "c":
        X = rng.normal(size=(rows, M)).astype(dt)
    else:
        extra = 3 if layout == "wide" else 4
        X = rng.normal(size=(rows, M + extra)).astype(dt)[:, 1 if layout == "wide" else 0:][:, :M]
    W = rng.normal(size=(rows, N)).astype(dt)
    A = X.T  # the transposed VIEW the reference's gradient graph hands to Gemm
    _check(_gemm(hip, A, W), A, W, 1.0, 0.0, None, dt)
    Cm = rng.normal(size=(M, N)).astype(dt)
    _check(_gemm(hip, A, W, alpha=0.5, beta=-2.0, Cm=Cm), A, W, 0.5, -2.0, Cm, dt)
    # run to run: the slabs are added in workgroup order
    a, b = _gemm(hip, A, W), _gemm(hip, A, W)
    np.testing.assert_array_equal(a, b)


def test_shapes_outside_the_skinny_kernels_still_take_the_mfma_path(hip):
    rng = np.random.default_rng(3)
    for M, K, N in ((8192, 64, 17), (4000, 128, 8), (128, 4000, 8)):
        A, B = rng.normal(size=(M, K)), rng.normal(size=(K, N))
        _check(_gemm(hip, A, B), A, B, 1.0, 0.0, None, "float64")
