"""GPU: boundary dtypes beyond round 1 (pytensor/tensor/type.py:40-57) through the C-ABI —
what the golden fixtures cannot pin because the reference's own two linkers disagree."""
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


def _reduce(hip, op, x, acc, out, axis_len_first=True):
    from pytensor_amd.device import DeviceArray
    from pytensor_amd.dispatch.elemwise import device_reduce

    class _Env:
        lib = hip.lib()

    d = DeviceArray.from_host(x.reshape(-1))
    r = device_reduce(_Env, op, d, 1, x.size, 1, 0, 1, 0, acc, out, ())
    return r.to_host()


@pytest.mark.parametrize("dtype", ["uint8", "uint16", "uint32", "uint64", "int8", "int16", "int32", "int64"])
def test_min_max_follow_numpy_not_the_c_seed(hip, dtype):
    """Minimum over unsigned inputs: the reference's C code seeds the accumulator with the literal
    1 (elemwise.py:1609-1611) and returns min(1, true minimum); its NumPy linker and this
    backend return the true minimum.  64-bit maxima above 2^53 are exact here (the C linker
    rounds them through double, scalar/basic.py Maximum.c_code)."""
    rng = np.random.default_rng(5)
    info = np.iinfo(dtype)
    x = rng.integers(max(info.min, -(2**62)) // 2 + 7, info.max // 2, size=10_001).astype(dtype)
    x[x < 7] = 7 if info.min == 0 else x[x < 7]
    if dtype in ("uint64", "int64"):
        x[17] = np.asarray(2**62 + 12345, dtype=dtype)  # not representable in double
    assert _reduce(hip, "Minimum", x, dtype, dtype) == x.min()
    assert _reduce(hip, "Maximum", x, dtype, dtype) == x.max()


@pytest.mark.parametrize("dtype,acc", [("uint8", "uint64"), ("uint16", "uint64"), ("uint32", "uint64"), ("uint64", "uint64"),
                                       ("int8", "int64"), ("int16", "int64"), ("bool", "int64")])
def test_sum_prod_wide_accumulator_narrow_store(hip, dtype, acc):
    rng = np.random.default_rng(6)
    x = rng.integers(0, 2 if dtype == "bool" else 120, size=70_001).astype(dtype)
    for out in (acc, dtype) if dtype != "bool" else (acc,):
        want = np.add.reduce(x, dtype=acc).astype(out)
        got = _reduce(hip, "Add", x, acc, out)
        assert got.dtype == np.dtype(out) and got == want, (got, want)
    x = rng.integers(1, 4, size=45).astype(dtype) if dtype != "bool" else np.ones(45, dtype=bool)
    assert _reduce(hip, "Mul", x, acc, acc) == np.multiply.reduce(x, dtype=acc)


def test_float16_is_rounded_per_op_like_numpy(hip):
    """+,-,*,/ and sqrt on float16 are bit-identical to NumPy (IEEE half operations); Sum
    accumulates in float32 (elemwise.py:1383-1417) and rounds once at the end."""
    from pytensor_amd.executor import HipExecutable
    from util import load_case

    g, ins, cvm, py, meta = load_case("float16_storage")
    out = HipExecutable(g)(*ins)
    exact = [0, 1, 2, 6, 11, 12, 13, 14, 15]  # arithmetic / casts / comparisons / views: no libm involved
    for k in exact:
        np.testing.assert_array_equal(out[k], cvm[k], err_msg=f"float16_storage out{k}")
    for k, (a, b) in enumerate(zip(out, cvm)):
        assert a.dtype == b.dtype
        if b.dtype.kind == "f":
            np.testing.assert_allclose(a.astype("float64"), b.astype("float64"), rtol=1e-3, err_msg=f"out{k}")
