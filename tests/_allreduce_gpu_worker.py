"""Worker for tests/test_gpu_collective.py: two ranks sharing GPU 0 (gloo transport: RCCL refuses
two ranks on one device), each running a graph with an ``AllReduce`` node through the HIP executor."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from pytensor_amd import comm, ffi, replicas
from pytensor_amd.executor import HipExecutable
from pytensor_amd.ir import Graph

info = replicas.rank_info()
ffi.init(0)
dist = replicas.init_process_group(info, backend="gloo")
rng = np.random.default_rng(200 + info.rank)
x = rng.normal(size=(300, 7))
g = Graph(name="allreduce_gpu")
v = g.new_var("float64", (None, None), name="x")
s = g.new_var("float64", (None,))
r = g.new_var("float64", (None,))
o = g.new_var("float64", (None,))
g.add_node("CAReduce", {"scalar_op": "Add", "axis": [0], "acc_dtype": "float64", "dtype": "float64"}, [v], [s])
g.add_node("AllReduce", {"op": "sum"}, [s], [r])
g.add_node("Elemwise", {"scalar": {"in_dtypes": ["float64"], "out_dtypes": ["float64"],
                                    "body": [{"op": "Tanh", "in": [["i", 0]], "dtype": "float64"}], "outs": [["t", 0]]}}, [r], [o])
g.inputs, g.outputs = [v], [s, r, o]
exe = HipExecutable(g, auto_freeze=True)
outs = [exe(x) for _ in range(3)]  # stays eager: a collective is not captured
assert exe._auto_plan is None and exe.has_collective
res = {"x_colsum": x.sum(axis=0).tolist(), "local": outs[0][0].tolist(), "reduced": outs[2][1].tolist(), "tanh": outs[2][2].tolist(),
       "world": comm.world_size()}
with open(os.path.join(os.environ["DIST_OUT"], f"gpu_ar{info.rank}.json"), "w") as fh:
    json.dump(res, fh)
dist.destroy_process_group()
