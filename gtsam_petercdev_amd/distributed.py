"""One-process-per-GPU helpers used by bench.py (torch.distributed: backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).  The hot path itself has no data-path collective in the replica mode of
round 1; these helpers only do what the bench contract needs: rendezvous from the
torchrun environment, a barrier, and MAX-over-ranks of the timed region."""
from __future__ import annotations

import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str = "nccl", device=None):
    """Returns torch.distributed (initialised) or None when WORLD_SIZE == 1."""
    rank, local_rank, world = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def replica_seed(base_seed: int) -> int:
    """Every rank works on its own seeded replica of the workload (weak scaling)."""
    return base_seed + env_rank()[0]


def max_over_ranks(dist, value: float, device="cpu") -> float:
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(dist, steps: int, elapsed_local: float, device="cpu"):
    """(whole-job steps/s, ms per step) from the slowest rank's time: every rank did `steps` steps."""
    world = env_rank()[2]
    elapsed = max_over_ranks(dist, elapsed_local, device)
    return world * steps / elapsed, 1e3 * elapsed / steps
