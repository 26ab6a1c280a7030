"""Device time of ``pthip_eigh`` across sizes next to SciPy/LAPACK on ONE host core (threadpoolctl) and on all
of them — the "correct-first" tier against the yardstick the round-3 verdict used.
usage: python tools/bench_eigh.py [n ...]      (on the MI355X box)"""
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
    from threadpoolctl import threadpool_limits

    ffi.init(0)
    lib = ffi.lib()
    for n in sizes:
        rng = np.random.default_rng(n)
        M = rng.normal(size=(n, n))
        S = (M + M.T) / 2
        dS = DeviceArray.from_host(S)
        w, v = DeviceArray.empty((1, n), "float64"), DeviceArray.empty((1, n, n), "float64")
        call = lambda: ffi.check(lib.pthip_eigh(ffi.np_dtype_code("float64"), 1, n, 1, dS.ptr, w.ptr, v.ptr))
        us = timed(lib, call, 3 if n > 256 else 10)
        wr = np.linalg.eigvalsh(S)
        V = v.to_host()[0]
        err = float(np.abs(w.to_host()[0] - wr).max() / max(1.0, np.abs(wr).max()))
        res = float(np.abs(S @ V - V * w.to_host()[0][None, :]).max())
        with threadpool_limits(limits=1):
            t0 = time.perf_counter()
            scipy.linalg.eigh(S)
            one = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        scipy.linalg.eigh(S)
        allc = (time.perf_counter() - t0) * 1e3
        print(json.dumps({"n": n, "ms_hip": round(us / 1e3, 3), "ms_scipy_one_core": round(one, 2), "ms_scipy_all_cores": round(allc, 2), "host_cores": os.cpu_count(),
                          "max_eigenvalue_err_rel": err, "max_residual": res}), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048])
