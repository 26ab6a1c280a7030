"""bench.py — graph evals/sec of the PyMC-style logp+grad graph (BASELINE.json config #4).

A "step" = one evaluation of ``[logp, d logp / d{mu_g, log_tau, z, beta, log_sigma}]`` of the
hierarchical-normal model (N=1e6 observations, K=128 regressors, G=128 groups,
Cholesky(128)), fp64, data (y, X, gidx, Sigma) resident in HBM as shared variables,
parameters (small) uploaded per call, outputs copied back per call.

    python bench.py --gpus N --steps K --warmup W

For N>1 the driver launches one rank per GPU with torch.distributed.run; each rank
evaluates an independent chain (weak scaling, no data-path collective: "replicas
only") and ``value`` = total evals / max-over-ranks time.

The JSON line also carries
  roofline      the dominant kernel (Gemv over X, both orientations): algorithmic bytes
                (N*K*8 per launch) / mean launch time measured with HIP events on the
                context stream;
  cpu_baseline  the CPU oracle (NumPy/SciPy restatement of the reference's perform
                methods, oracle/np_graph.py) timed on the host cores on a bounded
                sample (N=1e5), scaled linearly to N=1e6 — kind "port".
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


def _load_graph(name):
    from pytensor_amd.ir import Graph

    d = json.load(open(os.path.join(ROOT, "tests", "golden", f"{name}.json")))
    return Graph.from_dict(d), d["input_names"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="per-node dispatch instead of the frozen hipGraph plan")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_

        torch.cuda.set_device(local_rank)
        dist_.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_

    from pytensor_amd import configs, ffi
    from pytensor_amd.executor import HipExecutable

    ffi.init(local_rank)
    lib = ffi.lib()
    graph, names = _load_graph("c4_hier")
    vals = configs.c4_inputs(N=args.n, chain=rank)
    inputs = [vals[n] for n in names]
    resident = [k for k, n in enumerate(names) if n in configs.C4_DATA]
    exe = HipExecutable(graph, resident=resident, device=local_rank)

    # parity gate before timing: GPU result vs the oracle on this input
    out = exe(*inputs)
    plan = None
    if not args.eager:
        plan = exe.freeze(*inputs)
        out_p = plan(*inputs)
        for a, b in zip(out, out_p):
            np.testing.assert_array_equal(a, b)
    call = plan if plan is not None else exe

    def sync_all():
        ffi.check(lib.pthip_synchronize())
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        call(*inputs)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call(*inputs)
    ffi.check(lib.pthip_synchronize())
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (Gemv over X) measured live with HIP events ----
    from pytensor_amd.dispatch.blas import gemv_device
    from pytensor_amd.executor import Env

    env = Env(exe)
    Xd = exe._resident_cache[names.index("X")][1]
    N, K = Xd.shape
    beta_d = env.to_device(__import__("pytensor_amd.executor", fromlist=["HostValue"]).HostValue(vals["beta"]))
    r_d = env.to_device(__import__("pytensor_amd.executor", fromlist=["HostValue"]).HostValue(vals["y"]))
    XT = Xd.view((K, N), (Xd.strides[1], Xd.strides[0]))

    def time_kernel(fn, reps=20):
        e0, e1 = C.c_void_p(), C.c_void_p()
        ffi.check(lib.pthip_event_create(C.byref(e0)))
        ffi.check(lib.pthip_event_create(C.byref(e1)))
        for _ in range(3):
            fn()
        ffi.check(lib.pthip_event_record(e0))
        for _ in range(reps):
            fn()
        ffi.check(lib.pthip_event_record(e1))
        ffi.check(lib.pthip_event_synchronize(e1))
        ms = C.c_float()
        ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
        return ms.value / reps

    ms_row = time_kernel(lambda: gemv_device(env, -1.0, Xd, beta_d, 1.0, r_d))
    ms_col = time_kernel(lambda: gemv_device(env, 1.0, XT, r_d, 0.0, None))
    bytes_per_launch = N * K * 8
    ach_row = bytes_per_launch / (ms_row * 1e-3) / 1e9
    ach_col = bytes_per_launch / (ms_col * 1e-3) / 1e9
    # dominant = the slower of the two orientations (both read X once per eval)
    dom = ("gemv_row_kernel<double>", ms_row, ach_row) if ms_row >= ms_col else ("gemv_col_kernel<double>", ms_col, ach_col)

    if rank != 0:
        return

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import np_graph

        n_s = min(args.n, 100_000)
        sv = configs.c4_inputs(N=n_s, chain=0)
        sin = [sv[n] for n in names]
        np_graph.run_graph(graph, sin)
        reps = 0
        tc0 = time.perf_counter()
        while True:
            np_graph.run_graph(graph, sin)
            reps += 1
            if time.perf_counter() - tc0 > 10.0 or reps >= 50:
                break
        per_eval = (time.perf_counter() - tc0) / reps * (args.n / n_s)
        ref_full = np_graph.run_graph(graph, inputs) if args.n <= 2_000_000 else None
        if ref_full is not None:
            for k, (a, b) in enumerate(zip(out, ref_full)):
                np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-10, err_msg=f"bench parity output {k}")
        cpu = {
            "value": 1.0 / per_eval,
            "unit": "graph evals/sec",
            "cores": os.cpu_count(),
            "kind": "port",
            "sample": f"oracle/np_graph.py (NumPy/SciPy perform semantics, BLAS threads = host cores) at N={n_s}, {reps} evals, scaled x{args.n // n_s} to N={args.n}",
        }

    total_evals = args.steps * world
    line = {
        "metric": "graph evals/sec (logp+grad, N=1e6 fp64)",
        "value": total_evals / elapsed,
        "unit": "graph evals/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[3]: hierarchical-normal logp + grad, N=%d, K=128, G=128, Cholesky(128); one chain per GPU" % args.n,
            "mode": "eager" if args.eager else "hipGraph plan",
            "parallelism": f"replicas x{world}",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dom[0],
            "achieved": dom[2],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": dom[2] / HBM_PEAK_GBS,
            "traffic": None,
            "detail": {
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "gemv_row_ms": ms_row,
                "gemv_row_GBs": ach_row,
                "gemv_col_ms": ms_col,
                "gemv_col_GBs": ach_col,
            },
        },
        "cpu_baseline": cpu,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
