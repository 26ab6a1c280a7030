"""Small-graph call overhead: ``Function.__call__`` under ``mode="hip"`` next to the reference C linker (``Mode("cvm")``)
in the same process on the same box — the regime the reference measures in tests/benchmarks/test_function.py:24-56
(minimal random draw, ``exp(x)`` of 1000 elements with / without ``trust_input`` / the thunk called directly, the
identity function) plus BASELINE config #1 (C1) and config #4 at N = 257, and the N at which hip wins on C1's graph.
TEST INFRASTRUCTURE side: needs the importable reference copy (oracle/_ref).

usage: python tools/bench_call_overhead.py [--json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_ref  # noqa: E402

make_ref.activate()
import pytensor  # noqa: E402
import pytensor.tensor as pt  # noqa: E402
from pytensor import In  # noqa: E402
from pytensor.compile.io import Out  # noqa: E402
from pytensor.compile.mode import Mode  # noqa: E402

import pytensor_amd  # noqa: E402
import ref_graphs  # noqa: E402
from pytensor_amd import configs  # noqa: E402

pytensor_amd.register()
CVM = Mode(linker="cvm", optimizer="fast_run")


def per_call_us(f, args, min_s=0.25):
    for _ in range(20):
        f(*args)
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(50):
            f(*args)
        n += 50
        dt = time.perf_counter() - t0
        if dt > min_s:
            return dt / n * 1e6


def both(name, build, args, trust=True, rows=None):
    out = {"case": name}
    for label, mode in (("hip_us", "hip"), ("cvm_us", CVM)):
        ins, outs, kw = build()
        f = pytensor.function(ins, outs, mode=mode, **kw)
        f.trust_input = trust
        out[label] = round(per_call_us(f, args), 2)
        if label == "hip_us":
            exe = f.vm.jit_fn
            out["hip_direct_us"] = round(per_call_us(exe, args), 2) if name.startswith("exp") else None
            out["replays"] = exe.stats["replays"]
    out["hip_over_cvm"] = round(out["hip_us"] / out["cvm_us"], 2)
    return out


def main():
    res = []
    x = pt.vector("x")
    res.append(both("identity (10 elements, borrow in/out)", lambda: ([In(x, borrow=True)], Out(x, borrow=True), {}), [np.zeros(10)]))
    res.append(both("exp(x), 1000 elements, trust_input", lambda: ([x], pt.exp(x), {}), [np.zeros(1000)]))
    res.append(both("exp(x), 1000 elements, checked inputs", lambda: ([x], pt.exp(x), {}), [np.zeros(1000)], trust=False))

    def c1(N):
        v = configs.c1_inputs(N=N)

        def build():
            xx, mu = pt.dvector("x"), pt.dscalar("mu")
            y = pt.exp(-0.5 * (xx - mu) ** 2).sum()
            return [xx, mu], [y, pytensor.grad(y, xx)], {}

        return build, [v["x"], np.asarray(v["mu"])]

    for N in (1000, 10_000, 100_000, 1_000_000):
        b, a = c1(N)
        res.append(both(f"C1 exp(-0.5(x-mu)^2).sum() + grad, N={N}", b, a))
    vals = configs.c4_inputs(N=257, K=16, G=8)

    def c4():
        params, outs = ref_graphs.build_c4(vals)
        return params, outs, {}

    res.append(both("config #4 at N=257 (K=16, G=8): 6 outputs", c4, [np.asarray(vals[n]) for n in configs.C4_PARAMS]))
    # the N at which hip wins on C1's graph (linear interpolation between the measured sizes is enough to say where)
    c1s = [r for r in res if r["case"].startswith("C1")]
    win = next((r["case"].split("N=")[1] for r in c1s if r["hip_over_cvm"] < 1.0), None)
    if "--json" in sys.argv:
        print(json.dumps({"rows": res, "c1_hip_wins_from_N": win, "cores": os.cpu_count()}))
        return
    print(f"host: {os.cpu_count()} cores; Function.__call__ wall time per call, microseconds (steady state, >= 0.25 s each)\n")
    print("| graph | hip | reference C linker (CVM) | hip / CVM | hip thunk called directly |")
    print("|---|---:|---:|---:|---:|")
    for r in res:
        print(f"| {r['case']} | {r['hip_us']} | {r['cvm_us']} | {r['hip_over_cvm']} | {r['hip_direct_us'] or ''} |")
    print(f"\nC1's graph: hip is faster than the C linker from N = {win} on (of the sizes measured).")


if __name__ == "__main__":
    main()
