"""Device time of the single-matrix latency kernels (Cholesky, Cholesky+solve, triangular solve)
through the C-ABI, HIP events around back-to-back launches on the library stream.

usage: python tools/bench_linalg.py [n=128] [reps=200]
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
    for _ in range(5):
        fn()
    ffi.check(lib.pthip_event_record(e0))
    for _ in range(reps):
        fn()
    ffi.check(lib.pthip_event_record(e1))
    ffi.check(lib.pthip_event_synchronize(e1))
    ms = C.c_float()
    ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value * 1e3 / reps


def main(n=128, reps=200):
    ffi.init(0)
    lib = ffi.lib()
    out = {"n": n, "reps": reps}
    for dtype in ("float64", "float32"):
        rng = np.random.default_rng(0)
        A = rng.normal(size=(n, n + 8))
        S = (A @ A.T / n + np.eye(n)).astype(dtype)
        b = rng.normal(size=n).astype(dtype)
        dS, db = DeviceArray.from_host(S), DeviceArray.from_host(b)
        L, x = DeviceArray.empty((n, n), dtype), DeviceArray.empty((n,), dtype)
        dt = ffi.np_dtype_code(dtype)
        fits = n * (n | 1) * S.itemsize <= 160 * 1024 - 256 and n <= 256
        res = {"potrf_us": timed(lib, lambda: ffi.check(lib.pthip_potrf(dt, 1, 1, n, dS.ptr, L.ptr)), reps)}
        if fits:
            res["potrf_trsv_us"] = timed(lib, lambda: ffi.check(lib.pthip_potrf_trsv(dt, 1, n, dS.ptr, db.ptr, L.ptr, x.ptr)), reps)
        res["trsv_lower_us"] = timed(
            lib, lambda: ffi.check(lib.pthip_trsm(dt, 1, 0, 0, 1, n, 1, L.ptr, 0, n, 1, db.ptr, 0, x.ptr)), reps
        )
        res["trsv_lower_trans_us"] = timed(
            lib, lambda: ffi.check(lib.pthip_trsm(dt, 0, 0, 0, 1, n, 1, L.ptr, 0, 1, n, db.ptr, 0, x.ptr)), reps
        )
        out[dtype] = {k: round(v, 2) for k, v in res.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    a = sys.argv
    main(int(a[1]) if len(a) > 1 else 128, int(a[2]) if len(a) > 2 else 200)
