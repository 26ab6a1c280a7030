"""A fixed sequence of the large-matrix linear-algebra entry points (for rocprofv3 runs):
potrf(4096) x3, trsm with a vector (4096) x5, trsm square (2048) x2, getrf(2048) x1, all fp64."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytensor_amd import ffi  # noqa: E402
from pytensor_amd.device import DeviceArray  # noqa: E402

ffi.init(0)
lib = ffi.lib()
dt = ffi.np_dtype_code("float64")
rng = np.random.default_rng(0)
n = 4096
A = rng.normal(size=(n, n + 8))
S = A @ A.T / n + np.eye(n)
dS, L = DeviceArray.from_host(S), DeviceArray.empty((n, n), "float64")
for _ in range(3):
    ffi.check(lib.pthip_potrf(dt, 1, 1, n, dS.ptr, L.ptr))
b = DeviceArray.from_host(rng.normal(size=n))
x = DeviceArray.empty((n,), "float64")
for k in range(5):
    ffi.check(lib.pthip_trsm(dt, 1, k & 1, 0, 1, n, 1, L.ptr, n * n, n, 1, b.ptr, n, x.ptr))
m = 2048
T = np.tril(rng.normal(size=(m, m)) / np.sqrt(m)); T[np.diag_indices(m)] = 2.0
dT, B = DeviceArray.from_host(T), DeviceArray.from_host(rng.normal(size=(m, m)))
X = DeviceArray.empty((m, m), "float64")
for k in range(2):
    ffi.check(lib.pthip_trsm(dt, 1, k, 0, 1, m, m, dT.ptr, m * m, m, 1, B.ptr, m * m, X.ptr))
G = DeviceArray.from_host(rng.normal(size=(m, m)))
LU, perm = DeviceArray.empty((m, m), "float64"), DeviceArray.empty((m,), "int64")
sg, la = DeviceArray.empty((1,), "float64"), DeviceArray.empty((1,), "float64")
ffi.check(lib.pthip_getrf(dt, 1, m, G.ptr, LU.ptr, perm.ptr, sg.ptr, la.ptr, 0))
ffi.check(lib.pthip_synchronize())
print("ok")
