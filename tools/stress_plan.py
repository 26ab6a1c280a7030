import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from util import load_case
from pytensor_amd import ffi
from pytensor_amd.executor import HipExecutable
ffi.init(0)
names = sys.argv[1:] or ["scan_grad"]
bad = 0
for name in names:
    g, ins, cvm, py, meta = load_case(name)
    for trial in range(25):
        exe = HipExecutable(g)
        want = exe(*ins)
        plan = exe.freeze(*ins, multi_stream=True)
        for rep in range(4):
            got = plan(*ins)
            for k, (a, b) in enumerate(zip(got, want)):
                if not np.array_equal(a, b):
                    bad += 1
                    print(name, "trial", trial, "rep", rep, "out", k, "segmented", plan.segmented, float(np.abs(np.asarray(a)-np.asarray(b)).max()))
        plan.close()
print("bad", bad)
