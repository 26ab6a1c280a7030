import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests")); sys.path.insert(0,os.path.join(ROOT,"oracle"))
import numpy as np
from util import load_case
from pytensor_amd import ffi
from pytensor_amd.executor import HipExecutable
ffi.init(0)
g, ins, cvm, py, meta = load_case("c1_gauss")
ins=[np.random.default_rng(0).normal(size=100000), np.asarray(0.3)]
exe=HipExecutable(g, resident=[0])
exe(*ins)
plan=exe.freeze(*ins)
ts=[]
for _ in range(40):
    t=time.perf_counter(); plan(*ins); ts.append((time.perf_counter()-t)*1e6)
print("first calls us:", [round(x) for x in ts])
n=2000
t=time.perf_counter()
for _ in range(n): plan(*ins)
print("steady us/call", (time.perf_counter()-t)/n*1e6)
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(n): plan(*ins)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
