"""Device time of pthip_softmax (one kernel: 1 HBM read + 1 write of the matrix).

usage: python tools/bench_softmax.py
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.device import DeviceArray  # noqa: E402
from tools.bench_gemm import timed  # noqa: E402


def main():
    ffi.init(0)
    lib = ffi.lib()
    for dtype, rows, cols in [("float64", 8192, 2048), ("float64", 1_000_000, 10), ("float32", 65536, 1024), ("float64", 64, 262144)]:
        x = DeviceArray.empty((rows, cols), dtype)
        out = DeviceArray.empty((rows, cols), dtype)
        ffi.check(lib.pthip_memset(x.ptr, 0, x.nbytes))
        dt = ffi.np_dtype_code(dtype)
        ms = timed(lib, lambda: ffi.check(lib.pthip_softmax(dt, 0, rows, cols, x.ptr, out.ptr)), 20)
        gbs = 2 * x.nbytes / ms / 1e6
        print(json.dumps({"dtype": dtype, "rows": rows, "cols": cols, "ms": round(ms, 4), "GBs_rw": round(gbs, 1), "frac_hbm": round(gbs / 8000, 3)}))


if __name__ == "__main__":
    main()
