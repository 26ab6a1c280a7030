"""Device time of pthip_gemv (row layout) for a few shapes, N(0,1) operands; knobs via env
(PTHIP_GEMV_XLDS, PTHIP_GEMV_ROWS).  usage: python tools/bench_gemv.py"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import timed

ffi.init(0)
lib = ffi.lib()
rng = np.random.default_rng(0)
for dtype, M, N in (("float64", 4096, 4096), ("float64", 16384, 1024), ("float32", 8192, 8192), ("float64", 1000000, 128)):
    a = rng.normal(size=(M, N)).astype(dtype); x = rng.normal(size=N).astype(dtype)
    A = DeviceArray.empty((M, N), dtype); X = DeviceArray.empty((N,), dtype); out = DeviceArray.empty((M,), dtype)
    ffi.check(lib.pthip_h2d(A.ptr, a.ctypes.data, a.nbytes)); ffi.check(lib.pthip_h2d(X.ptr, x.ctypes.data, x.nbytes))
    ffi.check(lib.pthip_synchronize())
    dt = ffi.np_dtype_code(dtype)
    ws_bytes = lib.pthip_gemv_workspace(dt, M, N, N, 1)
    ws = DeviceArray.empty((max(ws_bytes, 1),), "uint8")
    def run():
        ffi.check(lib.pthip_gemv(dt, M, N, 1.0, A.ptr, N, 1, X.ptr, 1, 0.0, None, 0, out.ptr, ws.ptr, ws_bytes))
    ms = timed(lib, run, 20)
    got = out.to_host(); ref = a @ x
    err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    gb = a.nbytes / ms / 1e6
    print(json.dumps({"dtype": dtype, "M": M, "N": N, "us": round(ms * 1e3, 2), "GB/s": round(gb, 1), "frac": round(gb / 8000, 3), "relerr": err,
                      "xlds": os.environ.get("PTHIP_GEMV_XLDS", "default"), "rows": os.environ.get("PTHIP_GEMV_ROWS", "default")}))
