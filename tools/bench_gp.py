"""The GP marginal-likelihood graph of tests/golden (Cholesky(n), two triangular solves with a vector, the
log-determinant, gradients) at n points through the lowered graph, next to the NumPy/LAPACK oracle on the
host — the end-to-end effect of the large-matrix kernels (task-graph Cholesky, row-block solves).

usage: python tools/bench_gp.py [n ...]      (on the MI355X box)
"""
import json
import os
import sys
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "oracle"))
import np_graph  # noqa: E402
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.executor import HipExecutable  # noqa: E402
from pytensor_amd.ir import Graph  # noqa: E402


def main(sizes):
    ffi.init(0)
    d = json.load(open(os.path.join(root, "tests", "golden", "gp_marginal_likelihood.json")))
    g = Graph.from_dict(d)
    z = np.load(os.path.join(root, "tests", "golden", "gp_marginal_likelihood.npz"))
    small = [z[f"in{k}"] for k in range(len(g.inputs))]
    for n in sizes:
        rng = np.random.default_rng(n)
        ins = [rng.normal(size=(n, *a.shape[1:])).astype(a.dtype) if (a.ndim >= 1 and a.shape[0] == small[0].shape[0] and a.size > 1) else a for a in small]
        exe = HipExecutable(g)
        exe(*ins)
        exe(*ins)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            exe(*ins)
        ms = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        np_graph.run_graph(g, ins)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        if os.environ.get("PTHIP_GP_NODES"):
            prof = sorted(exe.profile_nodes(ins, reps=3), key=lambda t: -t[2])[:12]
            print("top nodes (device ms, handler brackets):", [(k, op, round(ms, 3)) for k, op, ms in prof], flush=True)
        print(json.dumps({"n": n, "ms_per_eval_hip (wall, eager: the graph's runtime asserts read the device)": round(ms, 3),
                          "ms_oracle_numpy_host": round(cpu_ms, 1), "host_cores": os.cpu_count()}), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [512, 2048, 4096])
