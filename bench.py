"""bench.py — graph evals/sec of the PyMC-style logp+grad graph (BASELINE.json configs[3]).

A "step" = one evaluation of ``[logp, d logp / d{mu_g, log_tau, z, beta, log_sigma}]`` of the
hierarchical-normal model (N=1e6 observations, K=128 regressors, G=128 groups,
Cholesky(128)), fp64.  Data (y, X, gidx, Sigma) are shared variables resident in HBM when
the timed region starts; the parameters (small) go in and the 6 outputs come out through
the boundary on every step (both inside the timed region).

    python bench.py --gpus N --steps K --warmup W

For N>1 the driver launches one rank per GPU with torch.distributed.run; started WITHOUT a launcher
(``python bench.py --gpus N``, no WORLD_SIZE in the environment) the script starts its own N ranks the
same way (``replicas.ensure_world``) and rank 0 prints the one JSON line.  Each rank
evaluates an independent chain (weak scaling, no data-path collective: "replicas only",
SURVEY.md §8e) and ``value`` = total evals / max-over-ranks time.

``value`` / ``ms_per_step`` time the drop-in API: ``pytensor.function(params, outs, mode="hip")`` ->
``Function.__call__`` with ``trust_input=True``, on every rank (needs the importable reference copy ``oracle/_ref``,
which travels with the snapshot; without it the executor-level loop is timed and ``config.value_is`` says so).
``value_executor_level`` / ``ms_per_step_executor_level`` time the same K steps on the compiled ``HipExecutable``
plan called directly — the object ``HipLinker.jit_compile`` returns.

The JSON line also carries
  roofline      the dominant kernel (gchain_*: one pass over X for X@beta and X.T@w): algorithmic
                bytes (N*K*8 + 2*N*8 per launch) / mean launch duration measured live with HIP
                events on the context stream; ``traffic`` = HBM bytes per launch from two
                ``rocprofv3 --pmc`` passes (FETCH_SIZE corrected x2, WRITE_SIZE) run as child processes
                after the timed region at N=1; if the profiler is unavailable the committed summary
                ``profiles/pmc_c4_current.json`` is used — ``traffic_source`` says which;
  cpu_baseline  the REFERENCE ITSELF on the host cores of this box: the same graph compiled by
                the reference's C linker (``mode="CVM"``, ``trust_input=True``) from the importable
                copy ``oracle/_ref`` (a built artefact that travels with the snapshot; test
                infrastructure), at the full N, 1 warm-up + median of 5 evals — kind
                "reference-cvm" (SURVEY §8d).  Its outputs gate the HIP outputs at
                rtol 1e-12 + 8*eps*sum|term| (oracle/bounds.py).  ``port`` inside it keeps the
                NumPy/SciPy oracle timing (kind "port") at the full N as a second number.  When
                ``oracle/_ref`` is absent the port is the baseline and says so;
  configs       BASELINE configs #1, #2 (cheap / transcendental), #3 (Dot22, Gemv, BatchedDot), #5
                at their stated sizes: device-event timed hipGraph replays, achieved GB/s or
                TFLOP/s and the fraction of the roofline that bounds each (tools/bench_configs.py).
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


def live_pmc_traffic(kernel_prefix):
    """(HBM bytes per launch of the dominant kernel, how it was obtained) from two ``rocprofv3 --pmc``
    passes over ``tools/profile_c4_replay.py`` run as child processes of this bench: FETCH_SIZE and
    WRITE_SIZE need separate runs (TCC counter slots), the unit is KiB and on gfx950 FETCH_SIZE reports
    half the bytes of a wide streaming read (MI355X_MICROARCH.md, HBM section).  (None, reason) when
    rocprofv3 is unavailable or a pass fails — the caller falls back to the committed summary."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if "rocprofiler-sdk-tool" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF_") for k in os.environ):
        return None, "this run is itself being profiled (no counter passes nested inside a trace)"
    per_kernel = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pthip_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "profile_c4_replay.py"), "12"]
        try:
            subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60, check=True)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                return None, f"rocprofv3 --pmc {counter}: no rocpd database written"
            rows = sqlite3.connect(dbs[0]).execute(
                "select name, avg(counter_value) from pmc_events where counter_name=? group by name", (counter,)).fetchall()
            per_kernel[counter] = dict(rows)
        except Exception as e:  # noqa: BLE001 (any failure of the profiler pass: report it, do not fail the bench)
            return None, f"rocprofv3 --pmc {counter}: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for name, fetch in per_kernel["FETCH_SIZE"].items():
        if kernel_prefix in name.lower() or name.startswith(kernel_prefix):
            total = fetch * 1024 * 2 + per_kernel["WRITE_SIZE"].get(name, 0.0) * 1024
            return total, "live: rocprofv3 --pmc FETCH_SIZE (x2, KiB) and WRITE_SIZE in two child passes of this run, after the timed region"
    return None, "dominant kernel not in the PMC passes"


def _time_launches(lib, fn, reps=20):
    from pytensor_amd import ffi

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
    lib.pthip_event_destroy(e0)
    lib.pthip_event_destroy(e1)
    return ms.value / reps


def cpu_baseline(graph, names, vals, inputs, out_hip, args):
    """The reference C linker (or, without ``oracle/_ref``, the NumPy port) on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bounds
    import np_graph

    cores = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info

        blas = "; ".join(f"{d.get('internal_api')} {d.get('version')} x{d.get('num_threads')}" for d in threadpool_info())
    except Exception:  # pragma: no cover
        blas = "unknown"

    # -- the NumPy/SciPy restatement at the full N (second number; also a parity gate) ----------
    np_graph.run_graph(graph, inputs)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ref_full = np_graph.run_graph(graph, inputs)
        ts.append(time.perf_counter() - t0)
    port = {"value": 1.0 / float(np.median(ts)), "unit": "graph evals/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/np_graph.py at N={args.n} (no scaling), 1 warm-up + median of 5; BLAS: {blas}; elementwise single-threaded"}
    used_port = bounds.check_c4(out_hip, ref_full, vals, what="bench parity vs oracle")

    import make_ref

    if not make_ref.importable():
        port["parity_err_over_bound"] = max(used_port)
        port["note"] = "oracle/_ref absent on this box: the reference C linker could not be timed"
        return port
    make_ref.activate()
    import pytensor
    from pytensor.compile.mode import Mode

    import ref_graphs

    have_cxx = bool(pytensor.config.cxx)
    mode = Mode(linker="cvm" if have_cxx else "py", optimizer="fast_run")
    params, outs = ref_graphs.build_c4(vals)
    t0 = time.perf_counter()
    f = pytensor.function(params, outs, mode=mode)
    t_compile = time.perf_counter() - t0
    f.trust_input = True
    pv = [np.asarray(vals[n]) for n in configs_params()]
    ref_out = f(*pv)  # warm-up
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ref_out = f(*pv)
        ts.append(time.perf_counter() - t0)
    used = bounds.check_c4(out_hip, ref_out, vals, what="bench parity vs the reference C linker")
    return {
        "value": 1.0 / float(np.median(ts)),
        "unit": "graph evals/sec",
        "cores": cores,
        "kind": "reference-cvm" if have_cxx else "reference-py (no g++ on this box)",
        "sample": (f"pytensor.function(mode=Mode('cvm','fast_run')) from oracle/_ref, trust_input=True, N={args.n} (full size), "
                   f"1 warm-up + median of 5 evals (min {min(ts) * 1e3:.0f} ms, max {max(ts) * 1e3:.0f} ms; compile {t_compile:.1f} s); "
                   f"nproc={cores}, OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}, BLAS: {blas}; "
                   "C Elemwise loops single-threaded (config.openmp=False)"),
        "ms_per_eval": float(np.median(ts)) * 1e3,
        "parity_err_over_bound": max(used),
        "port": port,
    }


def make_function(vals, out_hip):
    """``pytensor.function(params, outs, mode="hip")`` — the drop-in API itself: the graph built with the reference's own
    front end (``oracle/_ref``: an importable copy that travels with the snapshot), rewritten by the reference's
    rewriter under the HIP mode, linked by ``HipLinker`` (pytensor_amd/linker.py), called through
    ``Function.__call__`` (compile/executor.py:651-744) and the JIT thunk (link/basic.py:670-684) with
    ``trust_input=True`` — the leg the reference's C linker is timed on in ``cpu_baseline``.  ``None`` when the
    reference copy is absent (the executor-level loop is then the headline and the line says so)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_ref

    if not make_ref.importable():
        return None
    make_ref.activate()
    import pytensor

    import pytensor_amd
    import ref_graphs

    pytensor_amd.register()
    params, outs = ref_graphs.build_c4(vals)
    t0 = time.perf_counter()
    f = pytensor.function(params, outs, mode="hip")
    t_compile = time.perf_counter() - t0
    f.trust_input = True
    pv = [np.asarray(vals[n]) for n in configs_params()]
    first = f(*pv)  # eager; the second call captures the plan, later ones replay it
    for a, b in zip(first, out_hip):
        np.testing.assert_array_equal(a, b)  # same IR, same kernels as the executor-level leg
    for _ in range(3):
        f(*pv)
    return f, pv, t_compile, str(pytensor.config.hip__resident)


def settle(call, seconds=0.15):
    """untimed evaluations for `seconds` (clocks / caches / interpreter at steady state before a timed region that is
    K = 20 steps = 4 ms in the driver's run); counted in ``warmup_effective``"""
    n, t = 0, time.perf_counter()
    while time.perf_counter() - t < seconds:
        call()
        n += 1
    return n


def configs_params():
    from pytensor_amd import configs

    return configs.C4_PARAMS


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000, help="observations N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-via-function", action="store_true", help="time the executor-level loop instead of pytensor.function(mode='hip')")
    ap.add_argument("--no-settle", action="store_true", help="no untimed settle evaluations before --warmup (see warmup_effective)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config measurements (configs #1, #2, #3, #5)")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed PMC summary instead of two rocprofv3 --pmc passes inside this run")
    ap.add_argument("--eager", action="store_true", help="per-node dispatch instead of the frozen hipGraph plan")
    ap.add_argument("--single-stream", action="store_true", help="frozen plan without the two-stream fork")
    ap.add_argument("--hotpath", action="store_true", help="also run the 45-case hot_* sweep (cold + warm), the blocked Cholesky and the GP graph (minutes; detail file only)")
    ap.add_argument("--detail", default=None, help="where the full record goes (default: gpurun_out/bench_detail.json when gpurun_out/ exists, else bench_detail.json)")
    return ap.parse_args(argv)


def measure(args):
    """Runs the timed region and every secondary measurement; returns the FULL record (rank 0) or None (other ranks).
    ``main`` prints only ``compact(record)`` as the final stdout line; the full record goes to the detail file."""
    from pytensor_amd import replicas

    replicas.ensure_world(args.gpus)  # `python bench.py --gpus N` with no launcher: start the N ranks ourselves
    from pytensor_amd import configs, ffi
    from pytensor_amd.executor import HipExecutable

    info = replicas.rank_info()
    lib = ffi.lib()  # load the HIP runtime of /opt/rocm before torch brings its own copy
    dev = replicas.device_for_rank(info, ffi.device_count())
    ffi.init(dev)
    dist = replicas.init_process_group(info)
    graph, names = _load_graph("c4_hier")
    vals = configs.c4_inputs(N=args.n, chain=info.rank)  # same data, one parameter draw per rank
    inputs = [vals[n] for n in names]
    resident = [k for k, n in enumerate(names) if n in configs.C4_DATA]
    exe = HipExecutable(graph, resident=resident, device=dev)

    out = exe(*inputs)  # uploads the resident data; result is parity-gated below
    plan = None
    if not args.eager:
        plan = exe.freeze(*inputs, multi_stream=not args.single_stream)
        for a, b in zip(out, plan(*inputs)):
            np.testing.assert_array_equal(a, b)
    call = plan if plan is not None else exe
    # determinism gate (also what brings clocks, caches and the interpreter to steady state before a
    # timed region that is only K = 20 steps = 4.5 ms in the driver's run): 64 more evaluations must
    # reproduce the parity-gated outputs bit for bit
    for _ in range(64):
        for a, b in zip(out, call(*inputs)):
            np.testing.assert_array_equal(a, b)

    # ---- the timed region: K evaluations through the drop-in API (pytensor.function(mode="hip")) ----
    fn = None if args.no_via_function else make_function(vals, out)
    if fn is not None:
        f, pv, t_compile, resident_mode = fn
        step = lambda: f(*pv)  # every call returns host arrays: nothing is in flight when it returns
    else:
        step = lambda: call(*inputs)
    # (0.15 s of untimed evaluations unless --no-settle: one of three runs of the round-4 tree started its timed region
    # 4 % slower than the leg timed a second later, profiles/r4z_bench_lines.txt)
    n_settle = 0 if args.no_settle else settle(step)
    for _ in range(args.warmup):
        step()
    ffi.check(lib.pthip_synchronize())
    replicas.barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ffi.check(lib.pthip_synchronize())
    elapsed = time.perf_counter() - t0
    replicas.barrier(dist)
    elapsed = replicas.max_over_ranks(dist, elapsed)
    fn_stats = None
    if fn is not None:
        exe_f = f.vm.jit_fn
        fn_stats = {"compile_s": t_compile, "replays": exe_f.stats["replays"], "eager_calls": exe_f.stats["eager_calls"],
                    "resident_uploads": exe_f.stats["resident_uploads"], "resident_mode": resident_mode}

    # ---- secondary: the same K steps on the executor-level loop (the compiled plan called directly, no Function) ----
    for _ in range(args.warmup):
        call(*inputs)
    ffi.check(lib.pthip_synchronize())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call(*inputs)
    ffi.check(lib.pthip_synchronize())
    elapsed_exec = time.perf_counter() - t0

    # ---- roofline of the dominant kernel: per-node HIP-event timing on the context stream ----
    # The dominant node is GemvChain: ONE generated kernel (gchain_*) that streams X once
    # for both X@beta and X.T@w (plus two ~3 us stage-2 reductions of its scalar outputs).
    # Kernel level, not node level: the launch is bracketed by two HIP events recorded on the context
    # stream immediately before and after it (executor.KernelTimer).  A node-level bracket of an eager
    # pass also contains the handler's host time whenever the stream has run dry (the Tail node's
    # Python planning is longer than its two small kernels), so it cannot rank kernels.
    prof = exe.profile_nodes(inputs, reps=20)
    kt = dict(getattr(exe, "last_kernel_times", {}))
    from pytensor_amd.executor import KernelTimer

    bracket_overhead = KernelTimer.calibrate()
    N, K = vals["X"].shape
    kernel_name, ms_kernel = max(kt.items(), key=lambda t: t[1]) if kt else max(((op, ms) for _, op, ms in prof), key=lambda t: t[1])
    op_dom = "GemvChain" if kernel_name.startswith("gchain_") else kernel_name
    if op_dom == "GemvChain":
        # X once + the two N-vectors the kernel streams: y and gidx (int64; read for the gather
        # a[gidx] and again, from cache, for the scatter-add).  w and a[gidx] never touch HBM;
        # the G-entry table and the per-workgroup partials (4 MB) are not counted.
        bytes_per_launch = N * K * 8 + 2 * N * 8
        kernel_name += " (fused Gemv(row) -> Composite -> Gemv(col), one pass over X)"
    else:
        bytes_per_launch = N * K * 8
    ms_dom = max(ms for _, _, ms in prof)
    achieved = bytes_per_launch / (ms_kernel * 1e-3) / 1e9
    node_times = {f"{k}:{op}": round(ms, 5) for k, op, ms in sorted(prof, key=lambda t: -t[2])[:8]}
    kernel_times = {k: round(v, 5) for k, v in sorted(kt.items(), key=lambda t: -t[1])[:8]}

    if info.rank != 0:
        return None

    traffic, traffic_source = None, None
    if args.n == 1_000_000 and info.world == 1 and not args.no_live_pmc and os.environ.get("PTHIP_BENCH_LIVE_PMC", "1") != "0":
        # after the timed region: the same replay loop under rocprofv3, one counter per pass
        traffic, traffic_source = live_pmc_traffic("gchain_" if op_dom == "GemvChain" else op_dom.lower())
    pmc_path = os.path.join(ROOT, "profiles", "pmc_c4_current.json")
    if traffic is None and os.path.exists(pmc_path) and args.n == 1_000_000:
        why = traffic_source
        pmc = json.load(open(pmc_path))
        for k, v in pmc.items():
            if (op_dom == "GemvChain" and k.startswith("gchain_")) or op_dom.lower() in k.lower():
                traffic = v["hbm_bytes"]
                traffic_source = ("profiles/pmc_c4_current.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, committed)"
                                  + (f"; live passes not used: {why}" if why else ""))
                break

    cpu = None
    if not args.no_cpu_baseline and info.world == 1:
        cpu = cpu_baseline(graph, names, vals, inputs, out, args)

    cfgs = None
    if not args.no_configs and info.world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs

        which = ("c1", "c2", "c3", "c5", "wide200", "wide200gemm") + (("chol", "gp", "hotpath") if args.hotpath else ())
        cfgs = bench_configs.measure(which, reps=10, check=False)  # (parity at these sizes: tests/test_gpu_fullsize.py)

    # matrix-pipe utilisation of the BLAS path by COUNTERS (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)):
    # a separate rocprofv3 --pmc pass cannot run inside this process's timed configs, so the committed summary of
    # tools/pmc_mfma.py on the final tree is carried (profiles/r7_mfma_pmc.md says how it was taken)
    mfma = None
    mpath = os.path.join(ROOT, "profiles", "r7_mfma_pmc.json")
    if os.path.exists(mpath):
        mj = json.load(open(mpath))
        pick = lambda pre: [v["mfma_util"] for k, v in mj.items() if k.startswith(pre) and v["avg_us"] < 2500 and (pre != "sgemm256" or v["avg_us"] < 200)]
        d, b = pick("dgemm"), pick("sgemm256")
        mfma = {"dot22_f64_4096": d[0] if d else None, "bdot_f32_512x256": round(sum(b) / len(b), 4) if b else None,
                "source": "profiles/r7_mfma_pmc.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES, committed)"}

    line = {
        "metric": "graph evals/sec (logp+grad, N=1e6 fp64)",
        "value": args.steps * info.world / elapsed,
        "unit": "graph evals/sec",
        "n_gpus": info.world,
        "steps": args.steps,
        "warmup": args.warmup,
        # evaluations actually run before the timed region: 1 eager + 1 plan check + the 64-replay determinism gate
        # + --warmup (the gate is what brings clocks / caches / the interpreter to steady state before K = 20 steps)
        "warmup_effective": args.warmup + n_settle + (4 if fn is not None else 0),
        "untimed_gate_evals_executor_level": 64 + (2 if plan is not None else 1),
        "ms_per_step": elapsed / args.steps * 1e3,
        # `value` = K evaluations through pytensor.function(mode="hip") -> Function.__call__ (trust_input=True): the drop-in
        # API (every rank times its own Function; when oracle/_ref is absent the executor-level loop is timed instead and
        # config.value_is says so).  value_executor_level: the same K steps on the compiled plan called directly.
        "value_executor_level": args.steps / elapsed_exec,
        "ms_per_step_executor_level": elapsed_exec / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[3]: hierarchical-normal logp + grad, N=%d, K=128, G=128, Cholesky(128); one chain per GPU" % args.n,
            "mode": "eager" if args.eager else "hipGraph plan",
            "value_is": ("pytensor.function(params, outs, mode='hip') called K times (Function.__call__, trust_input=True); value_executor_level = the compiled plan called directly"
                         if fn is not None else "executor-level loop (oracle/_ref absent or --no-via-function: no pytensor.function on this box)"),
            "function": fn_stats,
            "parallelism": f"replicas x{info.world}",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source if traffic else None,
            "detail": {
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel_ms": ms_kernel,
                "kernel_ms_is": "mean two-event bracket of the single launch minus the bracket's own overhead (calibrated in this process on a 256-byte fill: bracket - back-to-back per-launch time)",
                "event_bracket_overhead_ms": bracket_overhead,
                "kernel_ms_top8": kernel_times,
                "eager_node_ms_top8 (handler brackets: include host time when the stream runs dry)": node_times,
            },
        },
        "cpu_baseline": cpu,
        "configs": cfgs,
        "mfma_util": mfma,
    }
    return line


MAX_LINE = 4000  # bytes: the driver keeps ~8.7 KB of stdout tail; round 5's 26 KB line could not be parsed

_TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_BASELINE_CONFIGS = ("c1", "c2_cheap", "c2_transc", "c3_dot22", "c3_gemv", "c3_bdot", "c5", "wide_200", "wide_200_gemm")


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 1] + "~"


def _r(x, nd=4):
    return round(float(x), nd) if isinstance(x, float) else x


def compact(full, detail_path=None):
    """The one JSON line the driver parses: the contract's fields + ``roofline`` + ``cpu_baseline`` + one roofline
    fraction per BASELINE config — bounded to ``MAX_LINE`` bytes whatever the full record holds (tests/test_bench_line.py)."""
    out = {k: full.get(k) for k in _TOP}
    for k in ("value", "ms_per_step"):
        out[k] = _r(out[k], 6)
    cfg = full.get("config") or {}
    out["config"] = {"workload": _short(cfg.get("workload"), 160), "parallelism": cfg.get("parallelism"), "api": _short(cfg.get("value_is"), 90)}
    rf = full.get("roofline") or {}
    out["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    out["roofline"]["kernel"] = _short(rf.get("kernel"), 70)
    out["roofline"]["kernel_ms"] = _r((rf.get("detail") or {}).get("kernel_ms"), 6)
    out["roofline"]["traffic_source"] = _short(rf.get("traffic_source"), 60)
    cpu = full.get("cpu_baseline")
    if cpu:
        out["cpu_baseline"] = {k: _r(cpu.get(k)) for k in ("value", "unit", "cores", "kind", "ms_per_eval", "parity_err_over_bound")}
        out["cpu_baseline"]["sample"] = _short(cpu.get("sample"), 200)
        if out["value"] and cpu.get("value"):
            out["speedup_vs_cpu_baseline"] = _r(full["value"] / cpu["value"], 1)
    else:
        out["cpu_baseline"] = None
    cfgs = full.get("configs") or {}
    fr = {}
    for k in _BASELINE_CONFIGS:
        e = cfgs.get(k)
        if isinstance(e, dict):
            # HBM-bound configs whose working set fits the Infinity Cache: the COLD fraction (operands rotated through
            # >= 1.5 GiB per cycle, tools/bench_hotpath.cold_device_time_ms), never the back-to-back replay of one set;
            # otherwise the dominant kernel's own fraction where one was measured, else the whole replay's
            cold = str(e.get("frac_is", "")).startswith("cold")
            fr[k] = _r(e.get("frac") if cold else e.get("kernel_frac", e.get("frac")))
    out["configs"] = fr
    if full.get("mfma_util"):
        out["mfma_util"] = {k: v for k, v in full["mfma_util"].items() if k != "source"}
    out["value_executor_level"] = _r(full.get("value_executor_level"), 2)
    if detail_path:
        out["detail"] = detail_path
    text = json.dumps(out)
    if len(text) > MAX_LINE:  # cannot happen with the fields above; never let a growth of the record void a round again
        for k in ("configs", "detail", "value_executor_level", "speedup_vs_cpu_baseline"):
            out.pop(k, None)
        out["config"] = {"workload": _short(cfg.get("workload"), 100)}
        if out.get("cpu_baseline"):
            out["cpu_baseline"]["sample"] = _short(out["cpu_baseline"]["sample"], 60)
    return out


def main(argv=None):
    args = parse_args(argv)
    full = measure(args)
    if full is None:
        return
    detail = args.detail
    if detail is None:
        d = os.path.join(ROOT, "gpurun_out")
        detail = os.path.join(d, "bench_detail.json") if os.path.isdir(d) else os.path.join(ROOT, "bench_detail.json")
    try:
        with open(detail, "w") as fh:
            json.dump(full, fh, indent=1)
        shown = os.path.relpath(detail, ROOT)
    except OSError as e:  # read-only tree: the compact line still goes out
        shown = None
        print(f"bench.py: detail file not written ({e})", file=sys.stderr)
    sys.stderr.flush()
    # stdout carries exactly one line — the last thing this process writes
    print(json.dumps(compact(full, shown)), flush=True)


if __name__ == "__main__":
    main()
