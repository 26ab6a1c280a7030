"""Device time of pthip_gemm for a few shapes (HIP events around back-to-back launches).

usage: python tools/bench_gemm.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.device import DeviceArray  # noqa: E402

PEAK = {"float64": 78.6, "float32": 157.3}


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
    shapes = [
        ("float32", 1, 4096, 4096, 4096), ("float32", 1, 8192, 8192, 1024), ("float32", 512, 256, 256, 256),
        ("float32", 64, 1024, 1024, 256), ("float64", 1, 4096, 4096, 4096), ("float32", 1, 64, 2048, 1024),
        ("float32", 1, 64000, 1024, 1024),  # the hoisted x_t @ W of config #5 (T*B rows)
        # mid-size fp64 products (the solve steps of the blocked triangular solve, GP graphs): 64 x 64 tiles since round 5
        ("float64", 1, 512, 2048, 512), ("float64", 1, 1024, 1024, 1024), ("float64", 1, 512, 512, 512), ("float64", 1, 256, 2048, 256),
        ("float64", 1, 1024, 2048, 1024), ("float64", 1, 2048, 2048, 2048),
    ]
    if len(sys.argv) > 1:  # indices of the shapes to run (profiling one kernel at a time)
        shapes = [shapes[int(a)] for a in sys.argv[1:]]
    for dtype, batch, M, N, K in shapes:
        A = DeviceArray.empty((batch, M, K), dtype)
        B = DeviceArray.empty((batch, K, N), dtype)
        out = DeviceArray.empty((batch, M, N), dtype)
        # N(0,1) operands (all-zero ones run ~8 % faster: data-dependent power, tools/clock_probe.py)
        rng = np.random.default_rng(0)
        keep = []
        for x in (A, B):
            chunk = rng.normal(size=min(x.size, 1 << 24)).astype(dtype)
            keep.append(chunk)
            for off in range(0, x.size, chunk.size):
                n = min(chunk.size, x.size - off)
                ffi.check(lib.pthip_h2d(x.ptr + off * chunk.itemsize, chunk.ctypes.data, n * chunk.itemsize))
        ffi.check(lib.pthip_synchronize())
        dt = ffi.np_dtype_code(dtype)

        def run():
            ffi.check(lib.pthip_gemm(dt, batch, M, N, K, 1.0, A.ptr, M * K, K, 1, B.ptr, K * N, N, 1, 0.0, None, 0, 0, 0, out.ptr))

        # the very first launch of this shape on fresh buffers, timed on its own: rocprofv3 summaries of the bench show one
        # 25-27 ms launch per GEMM kernel (profiles/r3z_c3_kernel_stats.md max column) — is it this one?
        e0, e1 = C.c_void_p(), C.c_void_p()
        ffi.check(lib.pthip_event_create(C.byref(e0)))
        ffi.check(lib.pthip_event_create(C.byref(e1)))
        firsts = []
        for _ in range(3):
            ffi.check(lib.pthip_event_record(e0))
            run()
            ffi.check(lib.pthip_event_record(e1))
            ffi.check(lib.pthip_event_synchronize(e1))
            m1 = C.c_float()
            ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(m1)))
            firsts.append(round(m1.value, 4))
        ms = timed(lib, run, 10)
        tf = 2.0 * batch * M * N * K / ms / 1e9
        rec = {"dtype": dtype, "batch": batch, "M": M, "N": N, "K": K, "ms": round(ms, 4), "TFLOPs": round(tf, 1), "frac": round(tf / PEAK[dtype], 3),
               "first_three_launches_ms": firsts}
        if dtype == "float32" and M % 256 == 0 and N % 256 == 0 and os.environ.get("PTHIP_BENCH_ORIENT", "1") != "0":
            # the same product with A and / or B stored transposed (strides swapped: same buffers, other operand values)
            for tag, (a0, a1), (b0, b1) in (("NT", (K, 1), (1, K)), ("TN", (1, M), (N, 1)), ("TT", (1, M), (1, K))):
                def run_o(a0=a0, a1=a1, b0=b0, b1=b1):
                    ffi.check(lib.pthip_gemm(dt, batch, M, N, K, 1.0, A.ptr, M * K, a0, a1, B.ptr, K * N, b0, b1, 0.0, None, 0, 0, 0, out.ptr))
                mo = timed(lib, run_o, 10)
                rec[f"frac_{tag}"] = round(2.0 * batch * M * N * K / mo / 1e9 / PEAK[dtype], 3)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()
