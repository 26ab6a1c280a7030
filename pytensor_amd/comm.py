"""The one data-path collective: an explicit all-reduce across the ranks of a job.

north_star: "RCCL over xGMI only for the rare explicit all-reduce Op".  The reference has no
distributed layer (SURVEY.md §5 last row, §8e): independent ``Function`` calls are sharded as
replicas (``replicas.py``) and never communicate; a user who wants, say, the sum of per-shard
log-likelihoods inserts ``pytensor_amd.collective.all_reduce`` into the graph.

One process per GPU, ``torch.distributed`` as the transport: backend ``"nccl"`` IS RCCL on ROCm —
the reduction runs on the device buffer itself over xGMI (ring per link, 7 links x ~153 GB/s per
GPU); with ``gloo`` (CPU tests, ranks sharing one GPU) the value is staged through the host.
World size 1 is the identity.  The result is bit-identical on every rank (RCCL/gloo reduce in a
fixed rank order), so replicated downstream computations stay in lock step.
"""

from __future__ import annotations

import numpy as np

OPS = ("sum", "prod", "max", "min")
_TORCH_DTYPES = ("float32", "float64", "int32", "int64", "int8", "uint8", "int16", "float16", "bool")


def _dist():
    """torch.distributed when a process group is up, else None (single process = identity)."""
    try:
        import torch.distributed as dist
    except Exception:  # torch absent: only single-process use is possible
        return None
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


def world_size() -> int:
    d = _dist()
    return d.get_world_size() if d is not None else 1


def _reduce_op(dist, op: str):
    if op not in OPS:
        raise ValueError(f"all_reduce: unknown reduction {op!r} (one of {OPS})")
    return {"sum": dist.ReduceOp.SUM, "prod": dist.ReduceOp.PRODUCT, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]


def all_reduce_host(a: np.ndarray, op: str = "sum") -> np.ndarray:
    """Element-wise reduction of ``a`` over all ranks (host arrays; gloo or RCCL via a staging
    tensor).  Returns a new array of the same shape and dtype."""
    if op not in OPS:
        raise ValueError(f"all_reduce: unknown reduction {op!r} (one of {OPS})")
    a = np.asarray(a)
    dist = _dist()
    if dist is None:
        return np.array(a, copy=True)
    import torch

    if str(a.dtype) not in _TORCH_DTYPES:
        raise TypeError(f"all_reduce: dtype {a.dtype} has no collective")
    was_bool = a.dtype == np.bool_
    t = torch.from_numpy(np.ascontiguousarray(a.astype("uint8") if was_bool else a).copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=_reduce_op(dist, op))
    out = t.cpu().numpy()
    return (out != 0) if was_bool else out.astype(a.dtype, copy=False).reshape(a.shape)


class _Aliased:
    """A device range exposed through ``__cuda_array_interface__`` so that torch (and through it
    RCCL) operates on the executor's own HBM buffer — no copy, no torch allocator."""

    def __init__(self, ptr: int, shape, dtype):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": np.dtype(dtype).str, "data": (int(ptr), False),
            "version": 2, "strides": None,
        }


def all_reduce_device(x, op: str = "sum", device_index: int = 0):
    """``x``: contiguous ``DeviceArray``.  Returns a fresh ``DeviceArray`` holding the reduction.
    RCCL reduces in place on a private copy; both streams are drained around the collective
    (the executor's stream is not torch's): a rare, explicit synchronisation point."""
    from pytensor_amd import ffi
    from pytensor_amd.device import DeviceArray, copy_into

    if op not in OPS:
        raise ValueError(f"all_reduce: unknown reduction {op!r} (one of {OPS})")
    out = DeviceArray.empty(x.shape, x.dtype)
    copy_into(out, x)
    dist = _dist()
    if dist is None or out.size == 0:
        return out
    lib = ffi.lib()
    ffi.check(lib.pthip_synchronize())
    if dist.get_backend() != "nccl":
        host = all_reduce_host(out.to_host(), op)
        ffi.check(lib.pthip_h2d(out.ptr, host.ctypes.data, host.nbytes))
        ffi.check(lib.pthip_synchronize())
        return out
    import torch

    if str(out.dtype) not in _TORCH_DTYPES or str(out.dtype) == "bool":
        raise TypeError(f"all_reduce: dtype {out.dtype} has no RCCL collective")
    t = torch.as_tensor(_Aliased(out.ptr, (out.size,), out.dtype), device=torch.device("cuda", device_index))
    dist.all_reduce(t, op=_reduce_op(dist, op))
    torch.cuda.synchronize(device_index)
    return out
