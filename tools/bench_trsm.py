"""Device time of ``pthip_trsm`` for matrices beyond the LDS: a vector right-hand side (memory-bound:
n^2/2 entries of T) and a square one (n^3 flops), next to SciPy/LAPACK on the host cores.

usage: python tools/bench_trsm.py [n ...]      (on the MI355X box)
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
        isz = np.dtype(dtype).itemsize
        for n in sizes:
            rng = np.random.default_rng(n)
            Tm = np.tril(rng.normal(size=(n, n)) / np.sqrt(n))
            Tm[np.diag_indices(n)] = 2.0
            Tm = Tm.astype(dtype)
            dT = DeviceArray.from_host(Tm)
            dt = ffi.np_dtype_code(dtype)
            for nrhs in (1, n):
                b = rng.normal(size=(n, nrhs)).astype(dtype)
                db = DeviceArray.from_host(b)
                out = DeviceArray.empty(b.shape, dtype)
                for trans in (0, 1):
                    reps = 20 if nrhs == 1 else 4
                    us = timed(lib, lambda: ffi.check(lib.pthip_trsm(dt, 1, trans, 0, 1, n, nrhs, dT.ptr, n * n, n, 1, db.ptr, n * nrhs, out.ptr)), reps)
                    t0 = time.perf_counter()
                    scipy.linalg.solve_triangular(Tm, b, lower=True, trans=trans, check_finite=False)
                    cpu_us = (time.perf_counter() - t0) * 1e6
                    row = {"dtype": dtype, "n": n, "nrhs": nrhs, "trans": trans, "us": round(us, 1), "lapack_host_us": round(cpu_us, 1)}
                    if nrhs == 1:
                        row["GBps"] = round(n * n / 2 * isz / us / 1e3, 1)
                    else:
                        row["tflops"] = round(n * n * nrhs / us / 1e6, 2)
                    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [512, 2048, 4096])
