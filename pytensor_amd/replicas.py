"""Replica sharding of independent ``Function`` calls over the GPUs of one node.

The reference has no distributed layer: MCMC chains are independent
evaluations.  One process per GPU (``torch.distributed.run`` sets RANK / LOCAL_RANK /
WORLD_SIZE); each rank owns one device, one resident copy of the data and its own
chains.  There is no data-path collective — ``torch.distributed`` (RCCL on ROCm, gloo in
the CPU tests) is used only for the barrier and for max-over-ranks timing.
"""

from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class RankInfo:
    rank: int
    local_rank: int
    world: int


def rank_info() -> RankInfo:
    return RankInfo(
        int(os.environ.get("RANK", "0")),
        int(os.environ.get("LOCAL_RANK", "0")),
        int(os.environ.get("WORLD_SIZE", "1")),
    )


def _free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def ensure_world(n: int, argv: list[str] | None = None) -> None:
    """``python script.py --gpus N`` started WITHOUT a launcher (no ``WORLD_SIZE`` in the environment)
    starts its own N ranks: the script is re-executed as ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`` with the same arguments — the
    command the driver uses for N > 1 — and this process exits with the launcher's return code.  Under
    a launcher (``WORLD_SIZE`` set) or with N <= 1 it returns at once, so it is safe to call first thing
    in ``main()``.  A launcher world that disagrees with ``--gpus`` is an error, not a silent override."""
    import subprocess
    import sys

    world = os.environ.get("WORLD_SIZE")
    if world is not None:
        if n > 1 and int(world) != n:
            raise SystemExit(f"--gpus {n} but the launcher set WORLD_SIZE={world}")
        return
    if n <= 1:
        return
    argv = list(sys.argv if argv is None else argv)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *argv]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")  # what torchrun would set (and warn about) anyway
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this driver
    raise SystemExit(subprocess.call(cmd, env=env))


def device_for_rank(info: RankInfo, n_devices: int) -> int:
    """LOCAL_RANK → device index (one process per GPU; wraps only when ranks outnumber
    devices, which happens in the single-GPU CI test of the multi-rank path)."""
    return info.local_rank % max(n_devices, 1)


def chains_for_rank(n_chains: int, info: RankInfo):
    """Round-robin assignment chain i → rank i % world (chain i → GPU i mod 8)."""
    return [c for c in range(n_chains) if c % info.world == info.rank]


def init_process_group(info: RankInfo, backend: str | None = None):
    """Returns the torch.distributed module (or None when world == 1)."""
    if info.world <= 1:
        return None
    import torch
    import torch.distributed as dist

    if backend is None:
        backend = os.environ.get("PTHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        try:
            torch.cuda.set_device(info.local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", info.local_rank))
            # one tiny collective now: a broken RCCL setup must show up here, not inside the bench
            t = torch.zeros(1, device="cuda")
            dist.all_reduce(t)
            torch.cuda.synchronize()
            return dist
        except Exception as e:  # control plane only (barrier + max): fall back rather than fail
            import sys

            print(f"[pytensor_amd.replicas] RCCL init failed on rank {info.rank} ({e!r}); using gloo", file=sys.stderr)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
            backend = "gloo"
    dist.init_process_group(backend)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value: float) -> float:
    if dist is None:
        return float(value)
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value: float) -> float:
    if dist is None:
        return float(value)
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
