"""Can a big MFMA-bound GEMM run BESIDE the latency-bound Scan step kernels?  Two captured plans
launched on two streams: A = config #5 (GRU Scan, hoisted products inside), B = three
64000x1024x1024 fp32 products.  Device time of A alone, B alone, and A || B (both enqueued, one
event pair around the pair).  If A||B ~ max(A, B) the hoisted products can hide under the loop;
if ~ A + B the step kernels leave no room (or the queues serialise).

usage: python tools/overlap_probe.py     (on the MI355X box)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc  # noqa: E402
from pytensor_amd import configs, ffi  # noqa: E402
from pytensor_amd.executor import HipExecutable  # noqa: E402
from pytensor_amd.ir import Graph  # noqa: E402

ffi.init(0)
lib = ffi.lib()
os.environ["PTHIP_SCAN_OVERLAP"] = "0"
T = 500
v = configs.c5_inputs(T=T, B=64, H=1024)
g, names = bc.load("c5_gru")
ins = [v[n] for n in names]
exeA = HipExecutable(g, resident=range(len(ins)))
exeA(*ins)
planA = exeA.freeze(*ins, fetch_outputs=False, multi_stream=False)

# B: the three sequence products as one graph: Dot22 nodes of the c3-style IR are not needed — call the C ABI under capture
from pytensor_amd.device import DeviceArray  # noqa: E402

X = DeviceArray.from_host(np.random.default_rng(0).normal(size=(T * 64, 1024)).astype("float32"))
Ws = [DeviceArray.from_host((0.03 * np.random.default_rng(k).normal(size=(1024, 1024))).astype("float32")) for k in range(3)]
outs = [DeviceArray.empty((T * 64, 1024), "float32") for _ in range(3)]


def gemms():
    for W, o in zip(Ws, outs):
        ffi.check(lib.pthip_gemm(6, 1, T * 64, 1024, 1024, 1.0, X.ptr, 0, 1024, 1, W.ptr, 0, 1024, 1, 0.0, None, 0, 0, 0, o.ptr))


gemms()
ffi.check(lib.pthip_synchronize())
ffi.check(lib.pthip_capture_begin())
gemms()
gB = C.c_void_p()
ffi.check(lib.pthip_capture_end(C.byref(gB)))
gA = planA._graphs[0][1]


def timed(fn, reps=5):
    e0, e1 = C.c_void_p(), C.c_void_p()
    ffi.check(lib.pthip_event_create(C.byref(e0)))
    ffi.check(lib.pthip_event_create(C.byref(e1)))
    fn()
    ffi.check(lib.pthip_synchronize())
    ts = []
    for _ in range(reps):
        ffi.check(lib.pthip_synchronize())
        ffi.check(lib.pthip_event_record(e0))
        fn()
        ffi.check(lib.pthip_event_record(e1))
        ffi.check(lib.pthip_event_synchronize(e1))
        ffi.check(lib.pthip_synchronize())
        ms = C.c_float()
        ffi.check(lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms)))
        ts.append(ms.value)
    return float(np.median(ts))


def both():
    ffi.check(lib.pthip_stream_wait(1, 0))          # B starts behind the start event
    ffi.check(lib.pthip_graph_launch_on(gB, 1))
    ffi.check(lib.pthip_graph_launch_on(gA, 0))
    ffi.check(lib.pthip_stream_wait(0, 1))          # the end event behind both


a = timed(lambda: ffi.check(lib.pthip_graph_launch_on(gA, 0)))
b = timed(lambda: (ffi.check(lib.pthip_stream_wait(1, 0)), ffi.check(lib.pthip_graph_launch_on(gB, 1)), ffi.check(lib.pthip_stream_wait(0, 1))))
ab = timed(both)
print(json.dumps({"T": T, "A_scan_ms": a, "B_gemms_ms": b, "A_and_B_ms": ab, "sum_ms": a + b, "max_ms": max(a, b),
                  "overlap_fraction_of_B_hidden": (a + b - ab) / b}))
