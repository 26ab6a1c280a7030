import cProfile, pstats, json, os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests")); sys.path.insert(0,os.path.join(ROOT,"oracle"))
from util import load_case
from pytensor_amd import ffi
from pytensor_amd.executor import HipExecutable
ffi.init(0)
for name,res in (("c1_gauss",[0]),("c4_hier",None)):
    g, ins, cvm, py, meta = load_case(name)
    names = meta["input_names"]
    if name=="c1_gauss":
        import numpy as np
        ins=[np.random.default_rng(0).normal(size=100000), np.asarray(0.3)]
        resident=[0]
    else:
        resident=[k for k,n in enumerate(names) if n in ("y","X","gidx","Sigma")]
    exe=HipExecutable(g, resident=resident, auto_freeze=True)
    for _ in range(5): exe(*ins)
    n=3000
    t=time.perf_counter()
    for _ in range(n): exe(*ins)
    print(name, "us/call", (time.perf_counter()-t)/n*1e6)
    pr=cProfile.Profile(); pr.enable()
    for _ in range(n): exe(*ins)
    pr.disable()
    st=pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
