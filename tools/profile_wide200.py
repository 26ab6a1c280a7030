"""rocprofv3 helper: north_star's literal graph (tests/golden/wide_200.json: config #4 + 48 likelihood terms) as plan
replays only, so that the last dispatches of the trace are the steady-state replay timeline.
usage: rocprofv3 --kernel-trace -d DIR -o k -- python tools/profile_wide200.py [replays=20]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytensor_amd import configs, ffi
from pytensor_amd.executor import HipExecutable
from pytensor_amd.ir import Graph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
variant = sys.argv[2] if len(sys.argv) > 2 else "wide_200"  # or wide_200_gemm (the multi-response form with two Gemm nodes)
ffi.init(0)
d = json.load(open(os.path.join(ROOT, "tests", "golden", f"{variant}.json")))
g = Graph.from_dict(d)
vals = configs.wide200_gemm_inputs() if variant == "wide_200_gemm" else configs.wide200_inputs()
names = d["input_names"]
ins = [vals[k] for k in names]
params = set(configs.wide200_gemm_params() if variant == "wide_200_gemm" else configs.wide200_params())
exe = HipExecutable(g, resident=[k for k, nm in enumerate(names) if nm not in params])
t0 = time.perf_counter(); exe(*ins); t_first = time.perf_counter() - t0
plan = exe.freeze(*ins)
for _ in range(5):
    plan(*ins)
t0 = time.perf_counter()
for _ in range(n):
    plan(*ins)
ms = (time.perf_counter() - t0) / n * 1e3
from collections import Counter
print({"replays": n, "ms_per_eval": ms, "first_call_s": t_first, "nodes": len(exe.graph.nodes), "ops": dict(Counter(nd.op for nd in exe.graph.nodes).most_common(12)),
       "segments": [k for k, _ in plan._graphs] if hasattr(plan, "_graphs") else None})
