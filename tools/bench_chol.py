"""Device time of ``pthip_potrf`` across sizes — the LDS-resident kernel (n <= 141 fp64) and the
blocked multi-workgroup factorisation above it (csrc/linalg.hip ``chol_blocked``) — as TFLOP/s
(n^3/3 flops) and the fraction of the fp64 / fp32 MFMA peak, next to SciPy/LAPACK on the host cores.

usage: python tools/bench_chol.py [n ...]      (on the MI355X box)
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

PEAK = {"float64": 78.6, "float32": 157.3}


def main(sizes):
    import scipy.linalg

    ffi.init(0)
    lib = ffi.lib()
    rows = []
    for dtype in ("float64", "float32"):
        for n in sizes:
            rng = np.random.default_rng(n)
            A = rng.normal(size=(n, n + 8))
            S = (A @ A.T / n + np.eye(n)).astype(dtype)
            dS = DeviceArray.from_host(S)
            L = DeviceArray.empty((n, n), dtype)
            dt = ffi.np_dtype_code(dtype)
            reps = 50 if n <= 512 else (10 if n <= 2048 else 4)
            us = timed(lib, lambda: ffi.check(lib.pthip_potrf(dt, 1, 1, n, dS.ptr, L.ptr)), reps)
            t0 = time.perf_counter()
            for _ in range(3):
                scipy.linalg.cholesky(S, lower=True, check_finite=False)
            cpu_us = (time.perf_counter() - t0) / 3 * 1e6
            tf = n**3 / 3 / us / 1e6
            rows.append({"dtype": dtype, "n": n, "us": round(us, 1), "tflops": round(tf, 3), "frac_mfma_peak": round(tf / PEAK[dtype], 4),
                         "lapack_host_us": round(cpu_us, 1), "host_cores": os.cpu_count()})
            print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048, 4096])
