"""CPU: the N>1 (replica) path with world_size 2 over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replicas_world2_gloo(tmp_path):
    env = dict(os.environ, DIST_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", "29577",
        os.path.join(ROOT, "tests", "_dist_worker.py"),
    ]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert outs[0]["chains"] == [0, 2, 4, 6] and outs[1]["chains"] == [1, 3, 5, 7]
    # whole-job accounting: every chain exactly once, time = max over ranks on both ranks
    assert outs[0]["total"] == outs[1]["total"] == 8
    assert outs[0]["max"] == outs[1]["max"] >= max(o["elapsed"] for o in outs) - 1e-9
    # chains differ (different parameter draws), and are deterministic functions of the chain id
    all_logps = {**outs[0]["logps"], **outs[1]["logps"]}
    assert len(set(round(v, 9) for v in all_logps.values())) == 8


def test_self_launch_without_torchrun():
    """``python script.py --gpus 2`` with no launcher and no WORLD_SIZE starts two ranks itself
    (``replicas.ensure_world`` — what ``bench.py --gpus N`` relies on when nobody wraps it in torchrun)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    worker = os.path.join(ROOT, "tests", "_selflaunch_worker.py")
    r = subprocess.run([sys.executable, worker, "--gpus", "2", "--tag", "abc"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "sum": 3.0, "max": 1.0, "tag": "abc", "master": "127.0.0.1", "local_rank": 0}
    # N = 1: no launcher, no process group
    r = subprocess.run([sys.executable, worker, "--gpus", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # a launcher world that contradicts --gpus is refused
    r = subprocess.run([sys.executable, worker, "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_calls_ensure_world_before_touching_the_device():
    src = open(os.path.join(ROOT, "bench.py")).read()
    i, j = src.index("replicas.ensure_world(args.gpus)"), src.index("ffi.lib()")
    assert 0 < i < j


def test_single_rank_defaults():
    from pytensor_amd import replicas

    info = replicas.RankInfo(0, 0, 1)
    assert replicas.chains_for_rank(3, info) == [0, 1, 2]
    assert replicas.init_process_group(info) is None
    assert replicas.max_over_ranks(None, 1.5) == 1.5


def test_all_reduce_world2_gloo(tmp_path):
    """The explicit all-reduce (pytensor_amd/comm.py + the ``AllReduce`` IR node) on two ranks."""
    import numpy as np

    env = dict(os.environ, DIST_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", "29578",
        os.path.join(ROOT, "tests", "_allreduce_worker.py"),
    ]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = (json.load(open(tmp_path / f"ar{k}.json")) for k in range(2))
    xa, xb = np.array(a["x"]), np.array(b["x"])
    want = {"sum": xa + xb, "prod": xa * xb, "max": np.maximum(xa, xb), "min": np.minimum(xa, xb)}
    for op, w in want.items():
        # bit-identical on both ranks, equal to NumPy's reduction of the stacked shards
        assert a[op] == b[op]
        np.testing.assert_array_equal(np.array(a[op]), w)
    assert a["int_sum"] == b["int_sum"] == [0, 3, 6, 9]
    assert a["bool_max"] == b["bool_max"] == [True, False, True]
    tot = xa.sum() + xb.sum()
    assert a["graph"] == b["graph"]
    np.testing.assert_allclose(a["graph"], [tot, np.exp(tot)], rtol=1e-13)


def test_all_reduce_single_process_is_identity():
    import numpy as np

    from pytensor_amd import comm

    x = np.arange(6.0).reshape(2, 3)
    for op in comm.OPS:
        y = comm.all_reduce_host(x, op)
        assert y is not x
        np.testing.assert_array_equal(y, x)
    assert comm.world_size() == 1
    import pytest

    with pytest.raises(ValueError):
        comm.all_reduce_host(x, "mean")
