import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests")); sys.path.insert(0,os.path.join(ROOT,"oracle"))
import numpy as np
from pytensor_amd import ffi
from pytensor_amd.ir import Graph
from pytensor_amd.executor import HipExecutable
ffi.init(0)
def one(op, params, ins, outs):
    g=Graph(name=op); i=[g.new_var(dt,(None,)*nd) for dt,nd in ins]; o=[g.new_var(dt,(None,)*nd) for dt,nd in outs]
    g.add_node(op,params,i,o); g.inputs,g.outputs=i,o; return g
rng=np.random.default_rng(0)
for n in (64,128,256):
    A=rng.normal(size=(n,n)); S=A+A.T
    for name,g,x in (("SVD",one("SVD",{"full_matrices":False,"compute_uv":True},[("float64",2)],[("float64",2),("float64",1),("float64",2)]),A),
                     ("SVDvals",one("SVD",{"full_matrices":False,"compute_uv":False},[("float64",2)],[("float64",1)]),A),
                     ("Eigh",one("Eigh",{"lower":True},[("float64",2)],[("float64",1),("float64",2)]),S),
                     ("QR",one("QR",{"mode":"economic"},[("float64",2)],[("float64",2)]*2),A)):
        exe=HipExecutable(g, resident=[0]); exe(x); exe(x)
        t=time.perf_counter()
        for _ in range(3): exe(x)
        print(n,name,round((time.perf_counter()-t)/3*1e3,3),"ms")
