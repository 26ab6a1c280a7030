"""``ARange`` / ``Eye`` (index helpers built from host-known scalars) and the order-defined
scans ``CumOp`` / ``Argmax``.

Reference: pytensor/tensor/basic.py ``ARange`` 3139 (perform: ``np.arange``), ``Eye`` 1351
(``np.eye``); pytensor/tensor/extra_ops.py ``CumOp`` 281 (``np.cumsum``/``np.cumprod``);
pytensor/tensor/math.py ``Argmax`` 142 (reduced axes moved last and flattened, ``np.argmax``).
"""

from __future__ import annotations

import numpy as np

from pytensor_amd import ffi
from pytensor_amd.device import DeviceArray, contiguous_strides
from pytensor_amd.dispatch import handler
from pytensor_amd.executor import HOST_MAX, HostValue


def _host_or_device(env, a: np.ndarray):
    h = HostValue(a)
    return h if a.size <= HOST_MAX else env.to_device(h)


@handler("ARange")
def arange(node, inputs, env):
    # start/stop/step are scalars whose *values* fix the output length: host reads, like the
    # shape arithmetic of Alloc/Reshape (data-dependent in a frozen plan -> the plan refuses)
    start, stop, step = (np.asarray(env.to_host(i)).reshape(()) for i in inputs)
    return [_host_or_device(env, np.arange(start, stop, step, dtype=node.params["dtype"]))]


@handler("Eye")
def eye(node, inputs, env):
    n, m, k = (int(np.asarray(env.to_host(i)).reshape(())) for i in inputs)
    return [_host_or_device(env, np.eye(n, m, k, dtype=node.params["dtype"]))]


@handler("CumOp")
def cumop(node, inputs, env):
    x = env.to_device(inputs[0]).contiguous()
    axis = node.params["axis"]
    if axis >= x.ndim:
        raise ValueError(f"axis(={axis}) out of bounds")
    out = DeviceArray.empty(x.shape, x.dtype)
    if x.size == 0:
        return [out]
    if str(x.dtype) not in ("float64", "float32", "int64", "int32"):
        raise NotImplementedError(f"CumOp: dtype {x.dtype} on the device")
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    inner = int(np.prod(x.shape[axis + 1 :], dtype=np.int64))
    ffi.check(env.lib.pthip_cumulative(ffi.np_dtype_code(x.dtype), int(node.params["mode"] == "mul"), outer, x.shape[axis], inner, x.ptr, out.ptr))
    return [out]


@handler("Softmax")
def softmax(node, inputs, env):
    """``Softmax`` / ``LogSoftmax`` (pytensor/tensor/special.py:26,67) over ``axis``: the reduced
    axes are moved last (a view when they already are), one launch for all rows."""
    x = env.to_device(inputs[0])
    if x.dtype.kind != "f":
        raise NotImplementedError("Softmax: float32/float64 only on the device")
    axes = list(node.params["axis"])
    keep = [d for d in range(x.ndim) if d not in axes]
    perm = keep + axes
    xt = x.view([x.shape[d] for d in perm], [x.strides[d] for d in perm]).contiguous()
    rows = int(np.prod([x.shape[d] for d in keep], dtype=np.int64))
    cols = int(np.prod([x.shape[d] for d in axes], dtype=np.int64))
    out = DeviceArray.empty(xt.shape, x.dtype)
    if out.size:
        ffi.check(env.lib.pthip_softmax(ffi.np_dtype_code(x.dtype), int(bool(node.params["log"])), rows, cols, xt.ptr, out.ptr))
    if perm != list(range(x.ndim)):
        inv = [perm.index(d) for d in range(x.ndim)]
        out = out.view([out.shape[d] for d in inv], [out.strides[d] for d in inv])
    return [out]


@handler("Argmax")
def argmax(node, inputs, env):
    x = env.to_device(inputs[0])
    axes = list(node.params["axis"])
    keep = [d for d in range(x.ndim) if d not in axes]
    perm = keep + axes
    xt = x.view([x.shape[d] for d in perm], [x.strides[d] for d in perm]).contiguous()
    kept = tuple(x.shape[d] for d in keep)
    R = int(np.prod([x.shape[d] for d in axes], dtype=np.int64))
    rows = int(np.prod(kept, dtype=np.int64))
    out = DeviceArray.empty(kept, "int64")
    if R == 0:
        raise ValueError("attempt to get argmax of an empty sequence")
    if rows:
        if str(x.dtype) == "bool":
            raise NotImplementedError("Argmax: bool input on the device")
        ffi.check(env.lib.pthip_argmax(ffi.np_dtype_code(x.dtype), rows, R, xt.ptr, out.ptr))
    return [out]


def _sort(node, inputs, env, want_idx):
    """SortOp / ArgSortOp (pytensor/tensor/sort.py:31, 156): np.sort / np.argsort along ``axis``
    (a runtime scalar input).  ``kind`` does not matter for the values; index ties come out in
    position order (NumPy's stable answer), csrc/sort.hip."""
    x = env.to_device(inputs[0])
    axis = int(np.asarray(env.to_host(inputs[1])))
    if x.ndim == 0:
        raise np.exceptions.AxisError(axis, 0)
    if not -x.ndim <= axis < x.ndim:
        raise np.exceptions.AxisError(axis, x.ndim)
    axis %= x.ndim
    order = [d for d in range(x.ndim) if d != axis] + [axis]
    xt = x.view([x.shape[d] for d in order], [x.strides[d] for d in order]).contiguous()
    n = xt.shape[-1]
    rows = xt.size // n if n else 0
    out = DeviceArray.empty(xt.shape, "int64" if want_idx else x.dtype)
    if rows and n:
        ffi.check(env.lib.pthip_sort(ffi.np_dtype_code(x.dtype), rows, n, xt.ptr, None if want_idx else out.ptr,
                                     out.ptr if want_idx else None))
    inv = [order.index(d) for d in range(x.ndim)]
    res = out.view([out.shape[d] for d in inv], [out.strides[d] for d in inv])
    if want_idx and node.params.get("dtype", "int64") != "int64":
        from pytensor_amd.dispatch.elemwise import _cast

        res = _cast(env, res.contiguous(), node.params["dtype"])
    return [res]


@handler("SortOp")
def sort_op(node, inputs, env):
    return _sort(node, inputs, env, False)


@handler("ArgSortOp")
def argsort_op(node, inputs, env):
    return _sort(node, inputs, env, True)
