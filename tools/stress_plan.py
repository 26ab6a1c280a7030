"""Stress test: frozen plans must reproduce the eager results bit for bit, whatever the pool
held before (cases are interleaved so that buffers get recycled across graphs).
usage: python tools/stress_plan.py [rounds=8] [case ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from util import load_case  # noqa: E402

from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.executor import HipExecutable  # noqa: E402


def main(rounds, names):
    ffi.init(0)
    cases = {n: load_case(n) for n in names}
    bad = 0
    for rnd in range(rounds):
        for name, (g, ins, cvm, py, meta) in cases.items():
            exe = HipExecutable(g)
            want = exe(*ins)
            try:
                plan = exe.freeze(*ins, multi_stream=bool(rnd & 1))
            except (ffi.HipError, NotImplementedError):
                continue
            for rep in range(3):
                got = plan(*ins)
                for k, (a, b) in enumerate(zip(got, want)):
                    if not np.array_equal(a, b, equal_nan=True):
                        bad += 1
                        d = np.abs(np.asarray(a, dtype="float64") - np.asarray(b, dtype="float64"))
                        print(f"MISMATCH {name} round {rnd} rep {rep} out {k}: max abs diff {np.nanmax(d):.3e}, "
                              f"eager vs golden {np.nanmax(np.abs(np.asarray(b, dtype='float64') - cvm[k])):.3e}, "
                              f"plan vs golden {np.nanmax(np.abs(np.asarray(a, dtype='float64') - cvm[k])):.3e}")
            plan.close()
    print("mismatches:", bad)


if __name__ == "__main__":
    a = sys.argv[1:]
    r = int(a[0]) if a and a[0].isdigit() else 8
    names = [x for x in a if not x.isdigit()] or ["scan_grad", "scan_variants", "c5_gru", "c4_hier_small", "careduce_more", "indexing_more", "fuzz_f64_3", "softmax_shapes"]
    main(r, names)
