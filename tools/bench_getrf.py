"""Device time of ``pthip_getrf`` (LU with partial pivoting) across sizes, next to SciPy/LAPACK on the host.

usage: python tools/bench_getrf.py [n ...]      (on the MI355X box)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_linalg import timed  # noqa: E402
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.device import DeviceArray  # noqa: E402


def main(sizes):
    import scipy.linalg

    ffi.init(0)
    lib = ffi.lib()
    for dtype in ("float64", "float32"):
        for n in sizes:
            A = np.random.default_rng(n).normal(size=(n, n)).astype(dtype)
            dA = DeviceArray.from_host(A)
            LU = DeviceArray.empty((n, n), dtype)
            perm = DeviceArray.empty((n,), "int64")
            sg = DeviceArray.empty((1,), dtype)
            la = DeviceArray.empty((1,), dtype)
            dt = ffi.np_dtype_code(dtype)
            us = timed(lib, lambda: ffi.check(lib.pthip_getrf(dt, 1, n, dA.ptr, LU.ptr, perm.ptr, sg.ptr, la.ptr, 0)), 5 if n > 1024 else 20)
            t0 = time.perf_counter()
            scipy.linalg.lu_factor(A, check_finite=False)
            cpu_us = (time.perf_counter() - t0) * 1e6
            print(json.dumps({"dtype": dtype, "n": n, "us": round(us, 1), "tflops": round(2 * n**3 / 3 / us / 1e6, 3), "us_per_column": round(us / n, 2),
                              "lapack_host_us": round(cpu_us, 1), "host_cores": os.cpu_count()}), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096])
