"""The round-6 log-sum-exp / column-softmax kernels called directly through the C-ABI (for rocprofv3 --kernel-trace
--stats: one row per kernel — partial, finish, normalisation — instead of the node total).
usage: rocprofv3 --kernel-trace --stats -d DIR -o k -- python tools/bench_lse_kernels.py [reps=10]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ffi.init(0)
lib = ffi.lib()
rng = np.random.default_rng(0)
for rows, cols in ((8192, 2048), (1_000_000, 10), (1000, 1000)):
    x = rng.normal(size=(rows, cols)) * 3
    code = ffi.np_dtype_code(x.dtype)
    # rotate over enough copies that a pass never finds its operand in the Infinity Cache
    ncopy = max(2, int(1.5 * 2**30 // x.nbytes) + 1)
    ncopy = min(ncopy, 24)
    xs = [DeviceArray.from_host(x) for _ in range(ncopy)]
    out_r, out_c, out_full = DeviceArray.empty((rows,), x.dtype), DeviceArray.empty((cols,), x.dtype), [DeviceArray.empty(x.shape, x.dtype) for _ in range(min(ncopy, 6))]
    n = int(lib.pthip_colstat_workspace(code, 1, rows, cols))
    ws = DeviceArray.empty((n,), "uint8")
    for it in range(reps):
        dx = xs[it % ncopy]
        if cols <= int(lib.pthip_logsumexp_rows_max(code)):
            ffi.check(lib.pthip_logsumexp_rows(code, rows, cols, dx.ptr, out_r.ptr))
        ffi.check(lib.pthip_logsumexp_cols(code, 1, rows, cols, dx.ptr, out_c.ptr, ws.ptr, n))
        ffi.check(lib.pthip_softmax_cols(code, 0, 1, rows, cols, dx.ptr, out_full[it % len(out_full)].ptr, ws.ptr, n))
        ffi.check(lib.pthip_softmax(code, 0, rows, cols, dx.ptr, out_full[it % len(out_full)].ptr))
    ffi.check(lib.pthip_synchronize())
    print(rows, cols, "done")
