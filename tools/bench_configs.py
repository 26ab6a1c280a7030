"""Per-config measurements for BASELINE.json configs #1, #2, #3, #5 on one MI355X
(config #4 is bench.py).  Prints one JSON line per config: device time per evaluation
(HIP events around hipGraph replays, inputs resident in HBM), achieved GB/s or TFLOP/s
against the roofline that bounds it, and a parity check against the CPU oracle at the
full size (or a reduced size where the oracle would take minutes).

usage: python tools/bench_configs.py [c1 c2 c3 c5] [--reps R] [--no-check]
"""

from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from pytensor_amd import configs, ffi  # noqa: E402
from pytensor_amd.executor import HipExecutable  # noqa: E402
from pytensor_amd.ir import Graph  # noqa: E402

HBM_PEAK = 8000.0  # GB/s
F64_MFMA_PEAK = 78.6  # TFLOP/s
F32_MFMA_PEAK = 157.3


def load(name):
    d = json.load(open(os.path.join(ROOT, "tests", "golden", f"{name}.json")))
    return Graph.from_dict(d), d["input_names"]


def device_time_ms(plan, reps):
    lib = ffi.lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    ffi.check(lib.pthip_event_create(C.byref(e0)))
    ffi.check(lib.pthip_event_create(C.byref(e1)))
    for _ in range(3):
        plan.launch_async()
    ffi.check(lib.pthip_event_record(e0))
    for _ in range(reps):
        plan.launch_async()
    ffi.check(lib.pthip_event_record(e1))
    ffi.check(lib.pthip_event_synchronize(e1))
    ms = C.c_float()
    ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / reps


COLD = {}  # case -> (ms per evaluation with the Infinity Cache defeated, number of rotated operand sets)


def run_case(name, vals, reps, check=True, rtol=1e-10, oracle_vals=None, kernel_reps=10, cold_bytes=None):
    """``cold_bytes``: the case's algorithmic bytes when its working set is at or under the 256 MiB Infinity Cache
    (configs #2 and #3-Gemv): the replay is then ALSO timed over rotated copies of the operands
    (tools/bench_hotpath.cold_device_time_ms) and that number is the HBM fraction reported."""
    import np_graph

    g, names = load(name)
    inputs = [vals[n] for n in names]
    exe = HipExecutable(g, resident=range(len(inputs)))
    out = exe(*inputs)
    if check:
        ov = oracle_vals or vals
        oin = [ov[n] for n in names]
        ref = np_graph.run_graph(g, oin)
        got = out if oracle_vals is None else HipExecutable(g)(*oin)
        for k, (a, b) in enumerate(zip(got, ref)):
            np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(1.0, float(np.max(np.abs(b)))), err_msg=f"{name} out{k}")
    dplan = exe.freeze(*inputs, fetch_outputs=False)  # kernels only: no output D2H node
    t_dev = device_time_ms(dplan, reps)
    dplan.close()
    plan = exe.freeze(*inputs)
    for _ in range(2):  # untimed: the result ring pins its second block
        plan(*inputs)
    t0 = time.perf_counter()
    for _ in range(reps):
        plan(*inputs)
    t_wall = (time.perf_counter() - t0) / reps * 1e3
    plan.close()
    # per-kernel launch durations (HIP events bracketing each launch, eager pass): what the
    # roofline fraction of the dominant kernel is computed from — a replay additionally carries
    # the graph-launch floor (~8-10 us: config #1 is two ~2 us kernels and replays in 12 us)
    exe.profile_nodes(inputs, reps=kernel_reps)
    KERNELS[name] = dict(exe.last_kernel_times)
    if cold_bytes:
        import bench_hotpath

        del exe
        COLD[name] = bench_hotpath.cold_device_time_ms(g, inputs, cold_bytes, max(2, reps // 2))
    return t_dev, t_wall


KERNELS = {}


def _dominant(name, prefix=None):
    kt = {k: v for k, v in KERNELS.get(name, {}).items() if prefix is None or k.startswith(prefix)}
    if not kt:
        return None, None
    k = max(kt, key=kt.get)
    return k, kt[k]


def _hbm_entry(label, name, nbytes, td, tw, bound):
    """an HBM-bound config: ``frac`` is the COLD fraction (operands rotated through >= 1.5 GiB per cycle) when one was
    taken; the back-to-back replay of one working set is kept as ``frac_warm`` and flagged when it can only have come
    out of the Infinity Cache (> 6.3 TB/s)"""
    warm = nbytes / td / 1e6
    e = {"config": label, "ms_device_warm": td, "ms_call": tw, "achieved_warm": warm, "frac_warm": warm / HBM_PEAK, "l3": bool(warm > 6300.0), "unit": "GB/s",
         "peak": HBM_PEAK, "bound": bound}
    if name in COLD:
        tc, nsets = COLD[name]
        e.update({"ms_device": tc, "achieved": nbytes / tc / 1e6, "frac": nbytes / tc / 1e6 / HBM_PEAK, "frac_is": f"cold: {nsets} operand sets rotated, {nsets * nbytes / 1e6:.0f} MB per cycle"})
    else:
        e.update({"ms_device": td, "achieved": warm, "frac": warm / HBM_PEAK, "frac_is": "warm (one working set replayed)"})
    return e


def _with_kernel(entry, name, work, peak, prefix=None, scale=1e6):
    """adds the dominant kernel's own roofline: ``work`` (bytes or flops) / its launch duration"""
    k, ms = _dominant(name, prefix)
    if k is not None:
        src = "single-launch event bracket minus its calibrated overhead (executor.KernelTimer)"
        td = entry.get("ms_device")
        if td is not None and ms > td:
            # a kernel cannot take longer than the replay that contains it: the bracket correction is a mean, and on a
            # one-kernel plan the replay IS the kernel plus the graph-launch floor — the replay time is the better bound
            ms, src = td, "bounded by the device time of the replay that contains it (bracket read higher)"
        entry.update({"kernel": k, "kernel_ms": ms, "kernel_ms_source": src, "kernel_achieved": work / ms / scale, "kernel_frac": work / ms / scale / peak})
    return entry


def _chol_entries():
    """``pthip_potrf`` beyond one CU's LDS (the blocked factorisation, csrc/linalg.hip chol_blocked): a GP
    marginal likelihood at n = 1000-4000 is an ordinary PyMC graph.  HIP events around back-to-back calls."""
    import ctypes as C

    from pytensor_amd.device import DeviceArray

    lib = ffi.lib()
    out = {}
    for n, reps in ((2048, 6), (4096, 3)):
        rng = np.random.default_rng(n)
        A = rng.normal(size=(n, n + 8))
        S = A @ A.T / n + np.eye(n)
        dS, L = DeviceArray.from_host(S), DeviceArray.empty((n, n), "float64")
        call = lambda: ffi.check(lib.pthip_potrf(ffi.np_dtype_code("float64"), 1, 1, n, dS.ptr, L.ptr))
        e0, e1 = C.c_void_p(), C.c_void_p()
        ffi.check(lib.pthip_event_create(C.byref(e0)))
        ffi.check(lib.pthip_event_create(C.byref(e1)))
        for _ in range(2):
            call()
        ffi.check(lib.pthip_event_record(e0))
        for _ in range(reps):
            call()
        ffi.check(lib.pthip_event_record(e1))
        ffi.check(lib.pthip_event_synchronize(e1))
        ms = C.c_float()
        ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
        t = ms.value / reps
        fl = n**3 / 3
        # the timed result is checked: entry-wise backward bound of a Cholesky factorisation (Higham Thm 10.3;
        # same bound and constant as tests/test_gpu_chol_blocked.py / test_gpu_linalg_4096.py)
        Lh = L.to_host()
        resid = np.abs(Lh @ Lh.T - S)
        bound = 4.0 * n * np.finfo("float64").eps * (np.abs(Lh) @ np.abs(Lh).T)
        rob = float(np.max(resid / bound))
        assert rob <= 1.0 and not np.triu(Lh, 1).any(), f"chol_{n}: residual / bound = {rob}"
        out[f"chol_{n}"] = {"config": f"Cholesky({n}) f64, persistent task-graph kernel", "ms_device": t, "achieved": fl / t / 1e9, "unit": "TFLOP/s",
                            "peak": F64_MFMA_PEAK, "frac": fl / t / 1e9 / F64_MFMA_PEAK, "residual_over_bound": rob, "bound": "dependent chain of the 64-column diagonal tiles (23 us per tile column, profiles/r3t_chol_trace_4096.txt)"}
    return out


def _wide200_entry(reps, check, variant="wide_200"):
    """north_star's literal target graph (oracle/ref_graphs.build_wide200: config #4 + 48 likelihood terms = 202 Elemwise
    + 2 Gemv + 1 Cholesky in the lowered IR) at N = 1e6 through ``pytensor.function(mode="hip")``; the reference C linker
    on the same graph and inputs beside it (1 warm-up + 2 evals) gates the outputs.  ``variant="wide_200_gemm"``: the
    multi-response form (oracle/ref_graphs.build_wide200_gemm, R = 8 response columns): two real ``Gemm`` nodes in the
    lowered IR, X read once for ``O + X @ B`` and once for ``X.T @ R``."""
    import make_ref

    N, T = 1_000_000, configs.WIDE_T
    gemm = variant == "wide_200_gemm"
    vals = configs.wide200_gemm_inputs(N=N) if gemm else configs.wide200_inputs(N=N)
    pnames = configs.wide200_gemm_params() if gemm else configs.wide200_params()
    if gemm:
        R = configs.WIDE_R
        # X twice (forward product, backward product), Y and O once, the residual matrix written + read once, gidx, the terms
        nbytes = 2 * N * configs.C4_K * 8 + 4 * N * R * 8 + N * 8 + T * N * 8
        label = f"north_star target with a real Gemm: multi-response (R={R}) hierarchical-normal, Gemm x2 (O + X@B, X.T@R), Cholesky({configs.C4_K}) + {T} likelihood terms, N=1e6 f64"
    else:
        nbytes = N * configs.C4_K * 8 + 2 * N * 8 + T * N * 8  # X once, y + gidx, every term's vector once
        label = f"north_star target: hierarchical-normal (Gemv x2 fused, Cholesky({configs.C4_K})) + {T} likelihood terms, N=1e6 f64; 202 Elemwise in the lowered IR"
    entry = {"config": label, "algorithmic_MB": nbytes / 1e6, "unit": "GB/s", "peak": HBM_PEAK, "bound": "hbm"}
    if make_ref.importable():
        make_ref.activate()
        import pytensor
        from pytensor.compile.mode import Mode

        import pytensor_amd
        import ref_graphs

        pytensor_amd.register()
        params, outs = (ref_graphs.build_wide200_gemm if gemm else ref_graphs.build_wide200)(vals)
        f = pytensor.function(params, outs, mode="hip")
        f.trust_input = True
        pv = [np.asarray(vals[n]) for n in pnames]
        out = f(*pv)
        for _ in range(4):
            f(*pv)
        t0 = time.perf_counter()
        for _ in range(reps):
            f(*pv)
        t = (time.perf_counter() - t0) / reps * 1e3
        exe = f.vm.jit_fn
        entry.update({"ms_call": t, "api": "pytensor.function(mode='hip'), trust_input=True", "replays": exe.stats["replays"], "nodes_after_passes": len(exe.graph.nodes),
                      "launch_kinds": sorted({n.op for n in exe.graph.nodes if n.op in ("GemvChain", "MultiElemwise", "ElemwiseReduce", "Tail", "CholeskyTrsv", "Elemwise", "Gemm", "GemmFinish", "SolveTriangular")})})
        if check:
            fc = pytensor.function(params, outs, mode=Mode(linker="cvm" if pytensor.config.cxx else "py", optimizer="fast_run"))
            fc.trust_input = True
            ref = fc(*pv)
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                ref = fc(*pv)
                ts.append(time.perf_counter() - t0)
            worst = 0.0
            for k, (a, b) in enumerate(zip(out, ref)):
                b = np.asarray(b)
                # sums of N terms in another order: rtol 1e-12 + 8 eps N max|term| (|r| <= 8 over 5e7 normal draws scaled by
                # exp(0.3): the normal family's d/d log-scale term r^2 - 1 stays below 64)
                tol = 1e-12 * np.abs(b) + 8 * np.finfo("float64").eps * N * 64.0
                worst = max(worst, float(np.max(np.abs(np.asarray(a) - b) / tol)))
            assert worst <= 1.0, f"{variant}: |hip - reference C linker| / tol = {worst}"
            entry.update({"reference_cvm_ms": float(np.median(ts)) * 1e3, "speedup_vs_reference_cvm": float(np.median(ts)) * 1e3 / t, "parity_err_over_tol": worst,
                          "cores": os.cpu_count()})
    else:
        g, names = load(variant)
        inputs = [vals[n] for n in names]
        resident = [k for k, n in enumerate(names) if n not in pnames]
        exe = HipExecutable(g, resident=resident)
        exe(*inputs)
        plan = exe.freeze(*inputs)
        for _ in range(4):
            plan(*inputs)
        t0 = time.perf_counter()
        for _ in range(reps):
            plan(*inputs)
        t = (time.perf_counter() - t0) / reps * 1e3
        plan.close()
        entry.update({"ms_call": t, "api": "executor-level plan (oracle/_ref absent on this box)"})
    entry.update({"achieved": nbytes / entry["ms_call"] / 1e6, "frac": nbytes / entry["ms_call"] / 1e6 / HBM_PEAK})
    return entry


def _alu_roofline(entry, kernel_prefix="ew_"):
    """A kernel that is VALU-issue bound, not HBM bound (config #2 transcendental: 10 tanh + 10 exp per element): its
    roofline is the issue rate.  ``SQ_INSTS_VALU`` (wave-instructions per launch, profiles/r5_c2_pmc.json, collected with
    tools/pmc_kernels.py) x 4 cycles (a wave64 fp64 instruction occupies its SIMD for 4 cycles: 16 lanes/clk) /
    (1024 SIMDs x 2.4 GHz) = the time the chip needs just to issue the kernel's arithmetic."""
    path = os.path.join(ROOT, "profiles", "r5_c2_pmc.json")
    if not os.path.exists(path) or "kernel" not in entry:
        return entry
    pmc = json.load(open(path))
    row = pmc.get(entry["kernel"]) or next((v for k, v in pmc.items() if k.split("_")[1:2] == entry["kernel"].split("_")[1:2]), None)
    if not row or "SQ_INSTS_VALU" not in row:
        return entry
    valu = row["SQ_INSTS_VALU"]
    t_issue_ms = valu * 4 / (1024 * 2.4e9) * 1e3
    entry["alu_roofline"] = {"bound": "fp64 VALU issue", "SQ_INSTS_VALU_per_launch": valu, "valu_lane_instructions_per_element": valu * 64 / 1e7, "issue_bound_ms": t_issue_ms,
                             "frac": t_issue_ms / entry["kernel_ms"], "source": "profiles/r5_c2_pmc.json (rocprofv3 --pmc SQ_INSTS_VALU, tools/pmc_kernels.py)"}
    return entry


def measure(which=("c1", "c2", "c3", "c5", "chol", "gp", "hotpath", "wide200"), reps=20, check=True):
    """Device-event timed replays of BASELINE configs #1, #2, #3, #5 at their stated sizes
    (inputs resident in HBM).  Returns ``{key: {...}}``; imported by ``bench.py`` for the
    ``configs`` field of its JSON line."""
    res = {}
    if "c1" in which:
        v = configs.c1_inputs()
        td, tw = run_case("c1_gauss", v, reps, check=check)
        b = 1.6e6
        res["c1"] = _with_kernel({"config": "C1 exp(-0.5(x-mu)^2).sum()+grad N=1e5 f64", "ms_device": td, "ms_call": tw, "achieved": b / td / 1e6,
                     "unit": "GB/s", "peak": HBM_PEAK, "frac": b / td / 1e6 / HBM_PEAK, "bound": "launch latency (1.6 MB/eval)"}, "c1_gauss", b, HBM_PEAK)
    if "c2" in which:
        v = configs.c2_inputs()
        small = configs.c2_inputs(N=1_000_000)
        for key, nm, label in (("c2_cheap", "c2_cheap", "C2 cheap 52-op Composite+Sum N=1e7 f64"),
                               ("c2_transc", "c2_transc", "C2 transcendental (10 tanh + 10 exp) Composite+Sum N=1e7 f64")):
            b = 160e6
            td, tw = run_case(nm, v, reps, check=check, oracle_vals=small, cold_bytes=b)
            res[key] = _alu_roofline(_with_kernel(_hbm_entry(label, nm, b, td, tw, "hbm" if key == "c2_cheap" else "alu (transcendentals), hbm fraction shown"), nm, b, HBM_PEAK))
    if "c3" in which:
        v = configs.c3_inputs()
        small = configs.c3_inputs(M=512, B=8, Bn=64)
        td, tw = run_case("c3_dot22", v, max(3, reps // 4), check=check, oracle_vals=small)
        fl = 2 * 4096**3
        res["c3_dot22"] = _with_kernel({"config": "C3 Dot22 4096^3 f64", "ms_device": td, "ms_call": tw, "achieved": fl / td / 1e9, "unit": "TFLOP/s",
                           "peak": F64_MFMA_PEAK, "frac": fl / td / 1e9 / F64_MFMA_PEAK, "bound": "mfma f64"}, "c3_dot22", fl, F64_MFMA_PEAK, "gemm_", 1e9)
        b = 4096 * 4096 * 8
        td, tw = run_case("c3_gemv", v, reps, check=check, oracle_vals=small, cold_bytes=b)
        res["c3_gemv"] = _with_kernel(_hbm_entry("C3 Gemv 4096^2 f64", "c3_gemv", b, td, tw, "hbm"), "c3_gemv", b, HBM_PEAK, "gemv_")
        td, tw = run_case("c3_bdot", v, reps, check=check, rtol=1e-4, oracle_vals=small)
        fl = 2 * 512 * 256**3
        res["c3_bdot"] = _with_kernel({"config": "C3 BatchedDot 512x(256x256) f32", "ms_device": td, "ms_call": tw, "achieved": fl / td / 1e9,
                          "unit": "TFLOP/s", "peak": F32_MFMA_PEAK, "frac": fl / td / 1e9 / F32_MFMA_PEAK, "bound": "mfma f32"}, "c3_bdot", fl, F32_MFMA_PEAK, "gemm_", 1e9)
    if "c5" in which:
        T, B, H = 1000, 64, 1024
        v = configs.c5_inputs(T=T, B=B, H=H)
        small = configs.c5_inputs(T=5, B=8, H=64)
        td, tw = run_case("c5_gru", v, 3, check=check, rtol=2e-4, oracle_vals=small, kernel_reps=1)
        fl = 6 * 2 * B * H * H * T
        res["c5"] = {"config": f"C5 GRU Scan T={T} B={B} H={H} f32", "ms_device": td, "ms_call": tw, "us_per_step": td / T * 1e3,
                     "achieved": fl / td / 1e9, "unit": "TFLOP/s", "peak": F32_MFMA_PEAK, "frac": fl / td / 1e9 / F32_MFMA_PEAK,
                     "bound": "mfma f32 (skinny M=64, dependent steps)",
                     "kernels_us": {k: round(ms * 1e3, 2) for k, ms in sorted(KERNELS.get("c5_gru", {}).items(), key=lambda t: -t[1])[:4]}}
    if "chol" in which:
        res.update(_chol_entries())
    if "gp" in which:
        # an ordinary PyMC-shaped graph on the large-matrix kernels: GP marginal log-likelihood + gradient
        # (tests/golden/gp_marginal_likelihood), n = 2048 points, as a frozen plan (the graph's shape asserts
        # are host arithmetic since round 4, pytensor_amd/hostsplit.py); parity: tests/test_gpu_chol_blocked.py
        import bench_gp

        r = bench_gp.measure(2048, reps=max(3, reps // 2), parity=check, eager=False)
        res["gp_2048"] = {"config": "GP marginal likelihood + gradient, n = 2048 points, f64 (Cholesky(2048), 2 vector + 2 square triangular solves, 2048^3 products)",
                          "ms_device": r.get("ms_device"), "ms_call": r.get("ms_plan_call"), "mode": "frozen plan" if "ms_plan_call" in r else "eager: " + r.get("freeze_error", "?"),
                          "bound": "latency chain of the Cholesky / solves (see chol_2048) + 3 products on the MFMA GEMM",
                          **({"err_over_eps_cond": max(r["err_over_eps_cond_scale"]), "cond_K": r["cond_K"]} if "cond_K" in r else {})}
    if "hotpath" in which:
        # the kernel families the BASELINE configs never time (tools/bench_hotpath.py): broadcasting / strided / transposed
        # Elemwise, CAReduce 256^3 x 7 axes x 3 layouts (the reference's own benchmark), Softmax / logsumexp — each checked
        # against NumPy at the size it is timed at
        import bench_hotpath

        for grp, gen in (("ew", bench_hotpath.ew_cases), ("careduce", bench_hotpath.careduce_cases), ("softmax", bench_hotpath.softmax_cases)):
            for a, kw in gen(max(5, reps // 2)):
                try:
                    r = bench_hotpath.run(*a, **kw)
                except (AssertionError, RuntimeError, NotImplementedError, ValueError) as e:
                    r = {"key": a[0], "error": f"{type(e).__name__}: {e}"[:300]}
                res["hot_" + r.pop("key")] = r
    # (parity of both at N = 1e6 against the reference C linker: tests/test_gpu_fullsize.py; inside a bench run the
    #  reference is compiled and timed beside wide_200 only when `check` is on — 25 s of host time)
    if "wide200" in which:
        res["wide_200"] = _wide200_entry(max(5, reps // 2), check)
    if "wide200gemm" in which:
        res["wide_200_gemm"] = _wide200_entry(max(5, reps // 2), False, variant="wide_200_gemm")
    from pytensor_amd.executor import KernelTimer

    if KernelTimer.overhead_ms is not None:
        # kernel_ms / kernels_us are single-launch event brackets with this much (calibrated) bracket overhead removed
        res["_kernel_bracket_overhead_us"] = round(KernelTimer.overhead_ms * 1e3, 3)
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = 20
    if "--reps" in sys.argv:
        reps = int(sys.argv[sys.argv.index("--reps") + 1])
    which = [a for a in args if not a.isdigit()] or ["c1", "c2", "c3", "c5", "chol", "gp", "hotpath", "wide200"]
    ffi.init(0)
    # --no-check: skip the oracle comparison (it runs the graphs at a reduced size as well, which
    # would mix small launches into a rocprofv3 kernel summary of this command)
    for k, r in measure(which, reps, check="--no-check" not in sys.argv).items():
        print(json.dumps({"key": k, **r} if isinstance(r, dict) else {"key": k, "value": r}))


if __name__ == "__main__":
    main()
