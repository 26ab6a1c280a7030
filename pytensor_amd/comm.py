"""The one data-path collective: an explicit all-reduce across the ranks of a job.

north_star: "RCCL over xGMI only for the rare explicit all-reduce Op".  The reference has no
distributed layer (SURVEY.md §5 last row, §8e): independent ``Function`` calls are sharded as
replicas (``replicas.py``) and never communicate; a user who wants, say, the sum of per-shard
log-likelihoods inserts ``pytensor_amd.collective.all_reduce`` into the graph.

One process per GPU.  Device values are reduced by RCCL itself, called through the C-ABI
(``pthip_all_reduce``, csrc/comm.hip) on the executor's own stream: over xGMI (ring per link, 7 links
x ~153 GB/s per GPU), no torch tensors, no host staging.  ``torch.distributed`` is only the control
plane (rendezvous, the broadcast of RCCL's 128-byte unique id, host-side values); with ranks sharing
one GPU (which RCCL refuses) or on CPU the value goes through the host over gloo.  World size 1 is
the identity.  The result is bit-identical on every rank (RCCL/gloo reduce in a
fixed rank order), so replicated downstream computations stay in lock step.
"""

from __future__ import annotations

import numpy as np

OPS = ("sum", "prod", "max", "min")
# dtypes BOTH transports reduce (RCCL has no 16-bit integers, torch/gloo no unsigned types beyond
# uint8): `AllReduce.make_node` rejects everything else, so that a graph never depends on which
# transport the job happens to run on
DTYPES = ("float16", "float32", "float64", "int8", "uint8", "int32", "int64", "bool")
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


_OP_CODE = {"sum": 0, "prod": 1, "max": 2, "min": 3}
_rccl_ready = False


def _local_ranks_share_a_device() -> bool:
    import os

    from pytensor_amd import ffi

    lws = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return lws > max(ffi.device_count(), 1)


def transport() -> str:
    """``"rccl"`` (device buffers reduced over xGMI on the executor's stream) or ``"gloo"`` (staged
    through the host: CPU tests, several ranks sharing one GPU — RCCL refuses that).
    ``PTHIP_COMM=rccl|gloo`` overrides the choice."""
    import os

    forced = os.environ.get("PTHIP_COMM", "auto").lower()
    if forced in ("rccl", "gloo"):
        return forced
    return "gloo" if _local_ranks_share_a_device() else "rccl"


def _ensure_rccl(dist):
    """One RCCL communicator per process: rank 0 draws the unique id (C-ABI ``pthip_comm_unique_id``),
    the control plane (whatever backend ``torch.distributed`` runs on) broadcasts its 128 bytes,
    every rank joins (``pthip_comm_init``)."""
    global _rccl_ready
    if _rccl_ready:
        return
    import ctypes as C

    import torch

    from pytensor_amd import ffi

    lib = ffi.lib()
    rank, world = dist.get_rank(), dist.get_world_size()
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        ffi.check(lib.pthip_comm_unique_id(ident))
    t = torch.tensor(list(ident), dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    ident = (C.c_ubyte * 128)(*t.cpu().tolist())
    ffi.check(lib.pthip_comm_init(world, rank, ident))
    _rccl_ready = True


def shutdown():
    """Destroy the RCCL communicator (before ``torch.distributed.destroy_process_group``)."""
    global _rccl_ready
    if _rccl_ready:
        from pytensor_amd import ffi

        ffi.check(ffi.lib().pthip_comm_destroy())
        _rccl_ready = False


def all_reduce_device(x, op: str = "sum"):
    """``x``: ``DeviceArray``.  Returns a fresh contiguous ``DeviceArray`` holding the reduction
    over all ranks.  RCCL path: ``ncclAllReduce`` in place on a private copy, enqueued on the
    executor's own stream (csrc/comm.hip) — stream-ordered, no synchronisation, no host copy."""
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
    if transport() == "rccl":
        _ensure_rccl(dist)
        ffi.check(lib.pthip_all_reduce(ffi.np_dtype_code(out.dtype), _OP_CODE[op], out.size, out.ptr))
        return out
    ffi.check(lib.pthip_synchronize())
    host = all_reduce_host(out.to_host(), op)
    ffi.check(lib.pthip_h2d(out.ptr, host.ctypes.data, host.nbytes))
    ffi.check(lib.pthip_synchronize())
    return out
