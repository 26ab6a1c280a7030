"""Device time of the LU-based ops and of Eigh at n = 128 (eager handler launches between HIP events).
usage: python tools/bench_lu.py [n]"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.executor import HipExecutable  # noqa: E402
from pytensor_amd.ir import Graph  # noqa: E402


def unary(op, n_out_dims, params=None):
    g = Graph(name=op)
    a = g.new_var("float64", (None, None), name="A")
    outs = [g.new_var("float64", (None,) * d) for d in n_out_dims]
    g.add_node(op, params or {}, [a], outs)
    g.inputs, g.outputs = [a], outs
    return g


def main(n=128):
    ffi.init(0)
    rng = np.random.default_rng(0)
    A = rng.normal(size=(n, n)) + np.eye(n) * 2
    res = {"n": n}
    for name, g in (("MatrixInverse", unary("MatrixInverse", [2])), ("Det", unary("Det", [0])), ("SLogDet", unary("SLogDet", [0, 0])),
                    ("Eigh", unary("Eigh", [1, 2], {"lower": True}))):
        if name == "Eigh":
            A = (A + A.T) / 2
        exe = HipExecutable(g, resident=[0])
        exe(A)
        plan = exe.freeze(A, fetch_outputs=False)
        lib = ffi.lib()
        e0, e1 = C.c_void_p(), C.c_void_p()
        lib.pthip_event_create(C.byref(e0)); lib.pthip_event_create(C.byref(e1))
        for _ in range(3):
            plan.launch_async()
        lib.pthip_event_record(e0)
        for _ in range(20):
            plan.launch_async()
        lib.pthip_event_record(e1); lib.pthip_event_synchronize(e1)
        ms = C.c_float(); lib.pthip_event_elapsed_ms(e0, e1, C.byref(ms))
        res[name + "_us"] = round(ms.value / 20 * 1e3, 1)
        plan.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 128)
