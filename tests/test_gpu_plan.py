"""GPU: frozen hipGraph plans (single- and two-stream) reproduce the eager results bit for bit."""
import numpy as np
import pytest

from util import golden_cases, load_case

pytestmark = pytest.mark.gpu

# cases whose graphs read device data back to the host mid-graph cannot be frozen
FREEZABLE = [c for c in golden_cases() if c not in ()]


@pytest.fixture(scope="module")
def hip():
    from pytensor_amd import ffi

    if ffi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests must run on the MI355X box")
    ffi.init(0)
    return ffi


@pytest.mark.parametrize("name", FREEZABLE)
@pytest.mark.parametrize("multi", [False, True])
def test_frozen_plan_equals_eager(hip, name, multi):
    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case(name)
    exe = HipExecutable(g)
    want = exe(*ins)
    try:
        plan = exe.freeze(*ins, multi_stream=multi)
    except hip.HipError as e:
        if "data-dependent host read" in str(e):
            pytest.skip("graph reads device data on the host: not freezable")
        raise
    for rep in range(3):
        got = plan(*ins)
        for k, (a, b) in enumerate(zip(got, want)):
            np.testing.assert_array_equal(a, b, err_msg=f"{name} out{k} replay {rep}")
    plan.close()


def test_plan_new_parameters_same_signature(hip):
    """replays pick up new parameter values (the staging block is rewritten per call)"""
    import np_graph
    from pytensor_amd import configs
    from pytensor_amd.executor import HipExecutable
    from util import assert_parity

    g, ins, cvm, py, meta = load_case("c4_hier_small")
    names = meta["input_names"]
    resident = [k for k, n in enumerate(names) if n in configs.C4_DATA]
    exe = HipExecutable(g, resident=resident)
    exe(*ins)
    plan = exe.freeze(*ins)
    for chain in range(3):
        v = configs.c4_inputs(N=257, K=16, G=8, chain=chain)
        cur = [ins[k] if k in resident else np.asarray(v[n]) for k, n in enumerate(names)]
        got = plan(*cur)
        ref = np_graph.run_graph(g, cur)
        for k, (a, b) in enumerate(zip(got, ref)):
            assert_parity(a, b, 1e-12, f"chain {chain} out{k}")
    plan.close()


def test_bench_multi_rank_flow_on_one_gpu(hip, tmp_path):
    """The N>1 launch contract (torch.distributed.run, barrier, max-over-ranks) end to end,
    with two ranks sharing GPU 0 over gloo (RCCL refuses two ranks on one device)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PTHIP_DIST_BACKEND="gloo")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", "29581",
        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--rows", "20000",
    ]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"] is None


def test_coexists_with_torch_hip_runtime(hip):
    """bench.py's multi-rank path imports torch (which bundles its own libamdhip64):
    our library must keep working next to it."""
    import torch

    from pytensor_amd.executor import HipExecutable

    g, ins, cvm, py, meta = load_case("c1_gauss")
    a = HipExecutable(g)(*ins)
    if torch.cuda.is_available():
        t = torch.ones(1024, device="cuda").sum().item()
        assert t == 1024.0
    b = HipExecutable(g)(*ins)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
