"""Per-term kernel times of the wide model (tools/wide_model_ir.json: 40 likelihood terms of four families) at N = 1e6,
each term as its OWN launch (PTHIP_WIDE=0) — what the one-launch MultiElemwise form has to beat — and the fused form.
usage: python tools/probe_wide_terms.py [N=1000000]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    from pytensor_amd import ffi
    from pytensor_amd.executor import HipExecutable
    from pytensor_amd.ir import Graph
    N = int(sys.argv[2])
    ffi.init(0)
    g = Graph.from_dict(json.load(open(os.path.join(ROOT, "tools", "wide_model_ir.json"))))
    rng = np.random.default_rng(15)
    ins = [rng.normal(size=40) * 0.1, rng.normal(size=40) * 0.1] + [rng.normal(size=N) + 0.1 * k for k in range(40)]
    exe = HipExecutable(g, resident=range(2, 42))
    exe(*ins)
    exe.profile_nodes(ins, reps=5)
    kt = exe.last_kernel_times
    print(json.dumps({"kernels_us": {k[:40]: round(v * 1e3, 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1])[:12]}, "sum_us": round(sum(kt.values()) * 1e3, 1),
                      "n_kernels": len(kt)}))
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "1000000"
    for wide in ("0", "1"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", N], capture_output=True, text=True, env={**os.environ, "PTHIP_WIDE": wide})
        print("PTHIP_WIDE=" + wide, (p.stdout.strip().splitlines() or [p.stderr[-400:]])[-1])
