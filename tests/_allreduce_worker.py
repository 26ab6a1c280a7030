"""Worker for tests/test_dist.py::test_all_reduce_world2_gloo: the explicit collective on two ranks."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np

import np_graph
from pytensor_amd import comm, replicas
from pytensor_amd.ir import Graph

info = replicas.rank_info()
dist = replicas.init_process_group(info, backend="gloo")
assert comm.world_size() == 2
rng = np.random.default_rng(100 + info.rank)  # a different shard per rank
x = rng.normal(size=(5, 3))
res = {}
for op in comm.OPS:
    res[op] = comm.all_reduce_host(x, op).tolist()
res["int_sum"] = comm.all_reduce_host(np.arange(4, dtype="int64") * (info.rank + 1), "sum").tolist()
res["bool_max"] = comm.all_reduce_host(np.array([info.rank == 0, False, info.rank == 1]), "max").tolist()
# through the IR: logp shards summed over ranks, then used by a replicated elementwise node
g = Graph(name="allreduce_unit")
v = g.new_var("float64", (None, None), name="x")
s = g.new_var("float64", ())
r = g.new_var("float64", ())
o = g.new_var("float64", ())
g.add_node("CAReduce", {"scalar_op": "Add", "axis": [0, 1], "acc_dtype": "float64", "dtype": "float64"}, [v], [s])
g.add_node("AllReduce", {"op": "sum"}, [s], [r])
g.add_node("Elemwise", {"scalar": {"in_dtypes": ["float64"], "out_dtypes": ["float64"],
                                    "body": [{"op": "Exp", "in": [["i", 0]], "dtype": "float64"}], "outs": [["t", 0]]}}, [r], [o])
g.inputs, g.outputs = [v], [r, o]
out = np_graph.run_graph(g, [x])
res["graph"] = [float(out[0]), float(out[1])]
res["x"] = x.tolist()
with open(os.path.join(os.environ["DIST_OUT"], f"ar{info.rank}.json"), "w") as fh:
    json.dump(res, fh)
dist.destroy_process_group()
