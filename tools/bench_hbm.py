"""Achievable HBM bandwidth on this device with the library's own streaming kernels:
a contiguous fp64 sum (read-only) at several sizes.  The numbers calibrate what fraction of
the 8 TB/s datasheet peak a pure streaming read can reach (roofline 'achievable' line).

usage: python tools/bench_hbm.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.device import DeviceArray  # noqa: E402


def timed(lib, fn, reps):
    e0, e1 = C.c_void_p(), C.c_void_p()
    ffi.check(lib.pthip_event_create(C.byref(e0)))
    ffi.check(lib.pthip_event_create(C.byref(e1)))
    for _ in range(3):
        fn()
    ffi.check(lib.pthip_event_record(e0))
    for _ in range(reps):
        fn()
    ffi.check(lib.pthip_event_record(e1))
    ffi.check(lib.pthip_event_synchronize(e1))
    ms = C.c_float()
    ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / reps


def main():
    ffi.init(0)
    lib = ffi.lib()
    out = {}
    for n in (10_000_000, 100_000_000, 400_000_000):
        x = DeviceArray.empty((n,), "float64")
        ffi.check(lib.pthip_memset(x.ptr, 0, x.nbytes))
        res = DeviceArray.empty((), "float64")
        ws_bytes = lib.pthip_reduce_workspace(7, 1, n, 1)
        ws = DeviceArray.empty((max(ws_bytes, 8),), "uint8")
        f64 = ffi.np_dtype_code("float64")

        def run():
            ffi.check(lib.pthip_reduce(0, f64, f64, f64, x.ptr, res.ptr, 1, n, 1, 0, 1, 0, ws.ptr, ws_bytes))

        ms = timed(lib, run, 20)
        out[f"sum_f64_{n}"] = {"ms": round(ms, 4), "GBs": round(n * 8 / ms / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
