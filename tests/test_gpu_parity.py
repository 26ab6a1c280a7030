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
        assert_parity(a, b, meta["rtol"], f"{name} out{k} (hip vs reference C linker)")
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
        assert_parity(x, y, meta["rtol"], f"{name} out{k} fused vs unfused")
