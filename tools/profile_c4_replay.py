"""rocprofv3 helper: config #4 as plan replays only (no eager profiling pass afterwards), so that
the last dispatches of the trace are the steady-state replay timeline.
usage: rocprofv3 --kernel-trace -d DIR -o k -- python tools/profile_c4_replay.py [replays=40]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytensor_amd import configs, ffi
from pytensor_amd.executor import HipExecutable
from pytensor_amd.ir import Graph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ffi.init(0)
d = json.load(open(os.path.join(ROOT, "tests", "golden", "c4_hier.json")))
g = Graph.from_dict(d)
vals = configs.c4_inputs()
ins = [vals[k] for k in d["input_names"]]
exe = HipExecutable(g, resident=[k for k, nm in enumerate(d["input_names"]) if nm in configs.C4_DATA])
exe(*ins)
plan = exe.freeze(*ins)
for _ in range(10):
    plan(*ins)
t0 = time.perf_counter()
for _ in range(n):
    plan(*ins)
print({"replays": n, "ms_per_eval": (time.perf_counter() - t0) / n * 1e3})
