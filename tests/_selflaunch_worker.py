"""Worker for tests/test_dist.py::test_self_launch: a script with a ``--gpus`` flag started WITHOUT a
launcher must start its own ranks (``replicas.ensure_world``), exactly as ``bench.py`` does."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pytensor_amd import replicas

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--tag", default="x")
args = ap.parse_args()
replicas.ensure_world(args.gpus)
info = replicas.rank_info()
dist = replicas.init_process_group(info, backend="gloo")
replicas.barrier(dist)
tot = replicas.sum_over_ranks(dist, info.rank + 1)
mx = replicas.max_over_ranks(dist, float(info.rank))
if info.rank == 0:
    print(json.dumps({"n_gpus": info.world, "sum": tot, "max": mx, "tag": args.tag,
                      "master": os.environ.get("MASTER_ADDR"), "local_rank": info.local_rank}))
if dist is not None:
    dist.destroy_process_group()
