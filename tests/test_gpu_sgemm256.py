"""GPU: the 256 x 256-tile fp32 GEMM (csrc/gemm.hip ``sgemm256_kernel``: row-major A and B, M and N
multiples of 256, K a multiple of 16) through the C-ABI ``pthip_gemm`` — one tile, many tiles per
workgroup (the persistent loop and its next-tile prefetch), a batch, K down to one step, the
``beta*C`` epilogue with a broadcast row, against NumPy with the dot-product bound
``|err| <= c·eps·(|A||B|)`` (c stated), and against the 128 x 128 kernel on a shape both serve
(a transposed-B call of the same product) within the same bound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = float(np.finfo(np.float32).eps)
C_SUM = 8.0


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _gemm(hip, A, B, alpha=1.0, beta=0.0, Cm=None, b_transposed_storage=False):
    from pytensor_amd.device import DeviceArray

    batch = A.shape[0] if A.ndim == 3 else 1
    M, K = A.shape[-2:]
    N = B.shape[-1]
    dA = DeviceArray.from_host(np.ascontiguousarray(A))
    if b_transposed_storage:  # B stored as (N, K): K-contiguous -> the 128 x 128 kernel's TN instance
        Bt = np.ascontiguousarray(np.swapaxes(B, -1, -2))
        dB = DeviceArray.from_host(Bt)
        sB0, sB1, sBb = 1, K, (N * K if A.ndim == 3 else 0)
    else:
        dB = DeviceArray.from_host(np.ascontiguousarray(B))
        sB0, sB1, sBb = N, 1, (K * N if A.ndim == 3 else 0)
    out = DeviceArray.empty((*A.shape[:-2], M, N), "float32")
    if Cm is not None:
        dC = DeviceArray.from_host(np.ascontiguousarray(Cm))
        sC0 = 0 if Cm.shape[-2] == 1 else Cm.shape[-1]
        sC1 = 0 if Cm.shape[-1] == 1 else 1
        cptr = dC.ptr
    else:
        sC0 = sC1 = 0
        cptr = None
    hip.check(hip.lib().pthip_gemm(hip.np_dtype_code("float32"), batch, M, N, K, float(alpha), dA.ptr, M * K if A.ndim == 3 else 0, K, 1,
                                   dB.ptr, sBb, sB0, sB1, float(beta), cptr, 0, sC0, sC1, out.ptr))
    return out.to_host()


@pytest.mark.parametrize("shape", [(256, 256, 16), (256, 256, 256), (512, 768, 80), (1024, 256, 1024), (2048, 2304, 48)])
def test_sgemm256_matches_numpy(hip, shape):
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    A, B = rng.normal(size=(M, K)).astype("float32"), rng.normal(size=(K, N)).astype("float32")
    got = _gemm(hip, A, B)
    want = A.astype("float64") @ B.astype("float64")
    bound = C_SUM * EPS * (np.abs(A).astype("float64") @ np.abs(B).astype("float64")) + 1e-30
    assert np.max(np.abs(got - want) / bound) <= 1.0
    other = _gemm(hip, A, B, b_transposed_storage=True)  # same product through the 128 x 128 kernel
    assert np.max(np.abs(other - want) / bound) <= 1.0
    np.testing.assert_array_equal(got, _gemm(hip, A, B))  # deterministic


def test_sgemm256_batched_with_epilogue(hip):
    rng = np.random.default_rng(5)
    A, B = rng.normal(size=(7, 256, 64)).astype("float32"), rng.normal(size=(7, 64, 512)).astype("float32")
    got = _gemm(hip, A, B, alpha=-0.5)
    want = -0.5 * np.matmul(A.astype("float64"), B.astype("float64"))
    bound = C_SUM * EPS * 0.5 * np.matmul(np.abs(A).astype("float64"), np.abs(B).astype("float64"))
    assert np.max(np.abs(got - want) / bound) <= 1.0
    A2, B2 = A[0], B[0]
    row = rng.normal(size=(1, 512)).astype("float32")  # a bias row, broadcast along M (Gemm's z broadcast, gemm.py:194-198)
    got = _gemm(hip, A2, B2, alpha=2.0, beta=3.0, Cm=row)
    want = 2.0 * (A2.astype("float64") @ B2.astype("float64")) + 3.0 * row
    bound = C_SUM * EPS * (2.0 * (np.abs(A2).astype("float64") @ np.abs(B2).astype("float64")) + 3.0 * np.abs(row))
    assert np.max(np.abs(got - want) / bound) <= 1.0


def _gemm_oriented(hip, A, B, a_t, b_t, env256=None):
    """A (.., M, K) @ B (.., K, N) with A stored transposed (K x M, M-contiguous) when ``a_t`` and B stored
    transposed (N x K, K-contiguous) when ``b_t`` — the four stride patterns the reference maps to BLAS transpose
    flags (pytensor/tensor/blas/c_code/codegen.py:159-250)."""
    from pytensor_amd.device import DeviceArray

    batch = A.shape[0] if A.ndim == 3 else 1
    M, K = A.shape[-2:]
    N = B.shape[-1]
    b3 = A.ndim == 3
    if a_t:
        dA = DeviceArray.from_host(np.ascontiguousarray(np.swapaxes(A, -1, -2)))
        sA0, sA1 = 1, M
    else:
        dA = DeviceArray.from_host(np.ascontiguousarray(A))
        sA0, sA1 = K, 1
    if b_t:
        dB = DeviceArray.from_host(np.ascontiguousarray(np.swapaxes(B, -1, -2)))
        sB0, sB1 = 1, K
    else:
        dB = DeviceArray.from_host(np.ascontiguousarray(B))
        sB0, sB1 = N, 1
    out = DeviceArray.empty((*A.shape[:-2], M, N), "float32")
    hip.check(hip.lib().pthip_gemm(hip.np_dtype_code("float32"), batch, M, N, K, 1.0, dA.ptr, M * K if b3 else 0, sA0, sA1,
                                   dB.ptr, K * N if b3 else 0, sB0, sB1, 0.0, None, 0, 0, 0, out.ptr))
    return out.to_host()


@pytest.mark.parametrize("a_t,b_t", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("shape", [(256, 256, 16), (512, 768, 80), (1024, 256, 1024), (3, 256, 256, 256)])
def test_sgemm256_every_operand_orientation(hip, shape, a_t, b_t):
    """Round 4: the 256 x 256-tile kernel serves NT / TN / TT as well (round 3: row-major x row-major only, the
    other three fell back to the 128-tile kernel).  Same dot-product bound; bit-identical to itself on a repeat."""
    *bs, M, N, K = shape
    rng = np.random.default_rng(M + N + K + 7 * a_t + 13 * b_t)
    A, B = rng.normal(size=(*bs, M, K)).astype("float32"), rng.normal(size=(*bs, K, N)).astype("float32")
    got = _gemm_oriented(hip, A, B, a_t, b_t)
    want = np.matmul(A.astype("float64"), B.astype("float64"))
    bound = C_SUM * EPS * np.matmul(np.abs(A).astype("float64"), np.abs(B).astype("float64")) + 1e-30
    assert np.max(np.abs(got - want) / bound) <= 1.0
    np.testing.assert_array_equal(got, _gemm_oriented(hip, A, B, a_t, b_t))
