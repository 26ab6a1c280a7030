"""Per-call wall time of frozen plans on small graphs (host-overhead regime): single- vs
two-stream capture.   usage: python tools/bench_small.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc  # noqa: E402
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.executor import HipExecutable  # noqa: E402
import numpy as np  # noqa: E402


def main():
    ffi.init(0)
    for name in ("c4_hier_small", "c1_gauss", "c4_hier"):
        g, names = bc.load(name)
        d = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
        ins = [d[f"in{k}"] for k in range(len(names))]
        data = {"y", "X", "gidx", "Sigma", "x"}
        resident = [k for k, n in enumerate(names) if n in data]
        exe = HipExecutable(g, resident=resident)
        exe(*ins)
        res = {"case": name}
        for multi in (False, True):
            plan = exe.freeze(*ins, multi_stream=multi)
            for _ in range(50):
                plan(*ins)
            t0 = time.perf_counter()
            for _ in range(500):
                plan(*ins)
            res["two_stream_us" if multi else "one_stream_us"] = round((time.perf_counter() - t0) / 500 * 1e6, 1)
            res["segmented"] = bool(plan.segmented) if multi else res.get("segmented")
            plan.close()
        t0 = time.perf_counter()
        for _ in range(200):
            exe(*ins)
        res["eager_us"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
