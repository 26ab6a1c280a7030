"""Worker of tests/test_gpu_two_procs.py: evaluates the linalg_contention graph (Cholesky(2048), a vector triangular
solve at n = 4096, Det(1024)) in a loop on device 0 and checks every result.  argv: seed, seconds."""
import json, os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytensor_amd import ffi
from pytensor_amd.executor import HipExecutable
from pytensor_amd.ir import Graph

seed, seconds = int(sys.argv[1]), float(sys.argv[2])
ffi.init(0)
d = json.load(open(os.path.join(ROOT, "tests", "golden", "linalg_contention.json")))
g, names = Graph.from_dict(d), d["input_names"]
rng = np.random.default_rng(seed)
nS, nT, nM = 2048, 4096, 1024
A = rng.normal(size=(nS, nS + 8))
S = A @ A.T / nS + np.eye(nS)
Tm = np.tril(rng.normal(size=(nT, nT))) / np.sqrt(nT) + 2.0 * np.eye(nT)
b = rng.normal(size=nT)
M = rng.normal(size=(nM, nM)) / np.sqrt(nM) + np.eye(nM)
vals = {"S": S, "Tm": Tm, "b": b, "M": M}
ins = [vals[n] for n in names]
import scipy.linalg
Lref = scipy.linalg.cholesky(S, lower=True)
xref = scipy.linalg.solve_triangular(Tm, b, lower=True)
sref, lref = np.linalg.slogdet(M)
exe = HipExecutable(g, resident=range(len(ins)))
calls, fallbacks = 0, 0
t0 = time.time()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    while time.time() - t0 < seconds or calls < 3:
        L, x, dt = exe(*ins)
        calls += 1
        eps = np.finfo("float64").eps
        assert not np.triu(L, 1).any()
        resid = np.abs(L @ L.T - S)
        assert (resid <= 4.0 * nS * eps * (np.abs(L) @ np.abs(L).T)).all(), "cholesky residual"
        assert np.max(np.abs(Tm @ x - b)) <= 64 * nT * eps * (np.abs(Tm) @ np.abs(x)).max(), "triangular solve residual"
        assert np.sign(dt) == sref and abs(np.log(abs(dt)) - lref) <= 1e-9 * max(1.0, abs(lref)), ("det", dt, sref, lref)
    fallbacks = sum(1 for x_ in w if "launch-per-step" in str(x_.message))
print(json.dumps({"seed": seed, "calls": calls, "fallbacks_to_safe_mode": fallbacks, "seconds": round(time.time() - t0, 2)}))
