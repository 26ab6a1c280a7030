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


def gp_inputs(g, small, n):
    rng = np.random.default_rng(n)
    return [rng.normal(size=(n, *a.shape[1:])).astype(a.dtype) if (a.ndim >= 1 and a.shape[0] == small[0].shape[0] and a.size > 1) else a for a in small]


def kernel_matrix_cond(d, ins):
    """cond_2 of the kernel matrix the graph factorises (the input of its ``Cholesky`` node), from the oracle."""
    import copy

    dd = copy.deepcopy(d)
    chol = next(n for n in dd["nodes"] if n["op"] == "Cholesky")
    dd["outputs"] = [chol["inputs"][0]]
    K = np_graph.run_graph(Graph.from_dict(dd), ins)[0]
    w = np.linalg.eigvalsh(K)
    return float(w[-1] / w[0])


def measure(n, reps=10, parity=True, eager=True):
    """One size: wall per call of the frozen plan (and of the eager executor), device time per replay, the
    NumPy/LAPACK oracle's wall time, and the error of every output in units of eps * cond(K) * max|want|."""
    d = json.load(open(os.path.join(root, "tests", "golden", "gp_marginal_likelihood.json")))
    g = Graph.from_dict(d)
    z = np.load(os.path.join(root, "tests", "golden", "gp_marginal_likelihood.npz"))
    small = [z[f"in{k}"] for k in range(len(g.inputs))]
    ins = gp_inputs(g, small, n)
    exe = HipExecutable(g)
    got = exe(*ins)
    exe(*ins)
    res = {"n": n, "host_cores": os.cpu_count()}
    if eager:
        t0 = time.perf_counter()
        for _ in range(reps):
            exe(*ins)
        res["ms_eager_call"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    try:
        plan = exe.freeze(*ins)
    except Exception as e:  # noqa: BLE001 (report why the graph stays eager)
        res["freeze_error"] = f"{type(e).__name__}: {e}"[:300]
        plan = None
    if plan is not None:
        out = plan(*ins)
        for a, b in zip(out, got):
            np.testing.assert_array_equal(a, b)
        plan(*ins)
        t0 = time.perf_counter()
        for _ in range(reps):
            plan(*ins)
        res["ms_plan_call"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
        plan.close()
        sys.path.insert(0, os.path.join(root, "tools"))
        import bench_configs

        dplan = exe.freeze(*ins, fetch_outputs=False)
        res["ms_device"] = round(bench_configs.device_time_ms(dplan, reps), 4)
        dplan.close()
    if os.environ.get("PTHIP_GP_NODES"):
        prof = sorted(exe.profile_nodes(ins, reps=3), key=lambda t: -t[2])[:12]
        res["top_nodes_ms"] = [(k, op, round(ms, 3)) for k, op, ms in prof]
        res["top_kernels_ms"] = {k: round(v, 4) for k, v in sorted(exe.last_kernel_times.items(), key=lambda t: -t[1])[:10]}
        res["kernel_sum_ms"] = round(sum(exe.last_kernel_times.values()), 4)
    if parity:
        t0 = time.perf_counter()
        want = np_graph.run_graph(g, ins)
        res["ms_oracle_numpy_host"] = round((time.perf_counter() - t0) * 1e3, 1)
        cond = kernel_matrix_cond(d, ins)
        eps = np.finfo("float64").eps
        res["cond_K"] = cond
        res["err_over_eps_cond_scale"] = [float(np.max(np.abs(a - b)) / (eps * cond * max(1.0, float(np.max(np.abs(b)))))) for a, b in zip(got, want)]
        res["rel_err"] = [float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b))))) for a, b in zip(got, want)]
    return res


def main(sizes):
    ffi.init(0)
    for n in sizes:
        print(json.dumps(measure(n)), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [512, 2048, 4096])
