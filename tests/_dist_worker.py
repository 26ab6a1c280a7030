"""Worker for tests/test_dist.py: world_size-2 gloo run of the replica-sharding logic."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np

import np_graph
from pytensor_amd import configs, replicas
from pytensor_amd.ir import Graph

info = replicas.rank_info()
dist = replicas.init_process_group(info, backend="gloo")
chains = replicas.chains_for_rank(8, info)
d = json.load(open(os.path.join(ROOT, "tests", "golden", "c4_hier_small.json")))
g = Graph.from_dict(d)
names = d["input_names"]
# every rank evaluates its own chains (different parameter draws over the same data) —
# on CPU through the oracle here; on the GPU box bench.py does the same through the HIP path
logps = {}
replicas.barrier(dist)
t0 = time.perf_counter()
for c in chains:
    vals = configs.c4_inputs(N=257, K=16, G=8, chain=c)
    logps[c] = float(np_graph.run_graph(g, [vals[n] for n in names])[0])
time.sleep(0.05 * (info.rank + 1))  # uneven ranks: the max must win
elapsed = time.perf_counter() - t0
mx = replicas.max_over_ranks(dist, elapsed)
tot = replicas.sum_over_ranks(dist, len(chains))
replicas.barrier(dist)
out = {"rank": info.rank, "world": info.world, "chains": chains, "logps": logps, "elapsed": elapsed, "max": mx, "total": tot}
with open(os.path.join(os.environ["DIST_OUT"], f"rank{info.rank}.json"), "w") as fh:
    json.dump(out, fh)
dist.destroy_process_group()
