"""The "wide PyMC model" of SURVEY Appendix B / north_star ("~200 fused Elemwise"): 40 independent
likelihood terms (4 families) over N observations each, logp + gradients wrt 80 parameters.
IR lowered by HipLinker from the graph of tests/test_gpu_e2e.py::test_wide_model_many_fused_kernels
(tools/wide_model_ir.json).  usage: python tools/bench_wide.py [N=100000] [reps=50]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pytensor_amd import ffi
from pytensor_amd.executor import HipExecutable
from pytensor_amd.ir import Graph

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ffi.init(0)
d = json.load(open(os.path.join(ROOT, "tools", "wide_model_ir.json")))
g = Graph.from_dict(d)
rng = np.random.default_rng(15)
ins = [rng.normal(size=40) * 0.1, rng.normal(size=40) * 0.1] + [rng.normal(size=N) + 0.1 * k for k in range(40)]
exe = HipExecutable(g, resident=range(2, 42))
out = exe(*ins)
if N <= 200_000:
    import np_graph
    ref = np_graph.run_graph(g, ins)
    for a, b in zip(out, ref):
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-9 * N)
plan = exe.freeze(*ins)
for _ in range(5):
    plan(*ins)
t0 = time.perf_counter()
for _ in range(reps):
    plan(*ins)
ms = (time.perf_counter() - t0) / reps * 1e3
kinds = [k for k, _ in plan._graphs]
print(json.dumps({"N": N, "ms_per_eval": round(ms, 4), "nodes": len(exe.graph.nodes), "segments": kinds,
                  "algorithmic_MB": round(40 * N * 8 / 1e6, 1), "GBps": round(40 * N * 8 / ms / 1e6, 1)}))
