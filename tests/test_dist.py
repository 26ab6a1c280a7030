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


def test_single_rank_defaults():
    from pytensor_amd import replicas

    info = replicas.RankInfo(0, 0, 1)
    assert replicas.chains_for_rank(3, info) == [0, 1, 2]
    assert replicas.init_process_group(info) is None
    assert replicas.max_over_ranks(None, 1.5) == 1.5
