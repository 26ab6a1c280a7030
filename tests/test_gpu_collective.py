"""GPU: the explicit all-reduce node through the HIP executor (north_star: "RCCL over xGMI only
for the rare explicit all-reduce Op").  One GPU is what the test box has: world size 1 (identity)
two ranks sharing GPU 0 (host staging over gloo: RCCL refuses two ranks on one device), and the
RCCL leg itself with a communicator of one rank — the 8-GPU path is that same ``pthip_all_reduce``."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import np_graph
from pytensor_amd.ir import Graph

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


def _graph(op):
    g = Graph(name="allreduce_one")
    v = g.new_var("float32", (None,), name="x")
    r = g.new_var("float32", (None,))
    o = g.new_var("float32", (None,))
    g.add_node("AllReduce", {"op": op}, [v], [r])
    g.add_node("Elemwise", {"scalar": {"in_dtypes": ["float32", "float32"], "out_dtypes": ["float32"],
                                        "body": [{"op": "Mul", "in": [["i", 0], ["i", 1]], "dtype": "float32"}], "outs": [["t", 0]]}}, [r, v], [o])
    g.inputs, g.outputs = [v], [r, o]
    return g


@pytest.mark.parametrize("op", ["sum", "max"])
def test_single_rank_is_identity_and_stays_eager(hip, op):
    from pytensor_amd.executor import HipExecutable

    g = _graph(op)
    x = np.random.default_rng(3).normal(size=1000).astype("float32")
    exe = HipExecutable(g, auto_freeze=True)
    for _ in range(3):
        got = exe(x)
    want = np_graph.run_graph(g, [x])
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    assert exe._auto_plan is None and exe.has_collective
    with pytest.raises(NotImplementedError):
        exe.freeze(x)


def test_two_ranks_on_one_gpu_gloo(hip, tmp_path):
    env = dict(os.environ, DIST_OUT=str(tmp_path), OMP_NUM_THREADS="1", PTHIP_COMM="gloo")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", "29583",
        os.path.join(ROOT, "tests", "_allreduce_gpu_worker.py"),
    ]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"gpu_ar{k}.json")) for k in range(2))
    assert a["world"] == b["world"] == 2
    # each rank reduced its own shard on the device ...
    np.testing.assert_allclose(a["local"], a["x_colsum"], rtol=1e-12)
    np.testing.assert_allclose(b["local"], b["x_colsum"], rtol=1e-12)
    # ... the collective added them, bit-identically on both ranks, and the replicated tail agrees
    assert a["reduced"] == b["reduced"] and a["tanh"] == b["tanh"]
    np.testing.assert_array_equal(np.array(a["reduced"]), np.array(a["local"]) + np.array(b["local"]))
    np.testing.assert_allclose(a["tanh"], np.tanh(a["reduced"]), rtol=1e-12)


def test_rccl_single_rank_communicator_through_the_c_abi(hip):
    """The RCCL leg itself on the one GPU the box has: unique id -> communicator of one rank ->
    ``pthip_all_reduce`` on the executor's stream (identity for one rank) -> destroy.  Loads
    librccl.so through the C-ABI exactly as the 8-GPU path does."""
    import ctypes as C

    from pytensor_amd.device import DeviceArray

    lib = hip.lib()
    ident = (C.c_ubyte * 128)()
    hip.check(lib.pthip_comm_unique_id(ident))
    assert any(ident)
    hip.check(lib.pthip_comm_init(1, 0, ident))
    try:
        n, r = C.c_int(-1), C.c_int(-1)
        hip.check(lib.pthip_comm_size(C.byref(n), C.byref(r)))
        assert (n.value, r.value) == (1, 0)
        for dt, code in (("float64", hip.np_dtype_code("float64")), ("float32", hip.np_dtype_code("float32")), ("int64", hip.np_dtype_code("int64"))):
            x = (np.arange(5000) % 97 - 40).astype(dt)
            d = DeviceArray.empty(x.shape, dt)
            hip.check(lib.pthip_h2d(d.ptr, x.ctypes.data, x.nbytes))
            for op in range(4):
                hip.check(lib.pthip_all_reduce(code, op, d.size, d.ptr))
            np.testing.assert_array_equal(d.to_host(), x)
        assert lib.pthip_all_reduce(hip.np_dtype_code("int16"), 0, 4, d.ptr) != 0  # no RCCL type
        assert lib.pthip_comm_init(1, 0, ident) != 0  # one communicator per process
    finally:
        hip.check(lib.pthip_comm_destroy())
