"""One-process-per-GPU helpers (torch.distributed: backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests and the
one-GPU rehearsals): rendezvous from the torchrun environment, a barrier, MAX-over-ranks of the timed region for the
bench contract, and the all-reduce a SHARDED handle calls (include/gsx.h: gsx_set_shard) — the one data-path
collective of the hot path: the sum of every rank's share of the cap fronts, once per factorization."""
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


class _DeviceDoubles:
    """`count` doubles of device memory at `ptr`, as something torch.as_tensor aliases without a copy."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def torch_allreduce(dist, device, group=None):
    """allreduce(ptr, count) for ProductBackend.set_shard on top of torch.distributed.  backend "nccl": RCCL sums the
    library's own buffer in place (over xGMI between the GPUs of a node); backend "gloo" (rehearsals with several ranks
    on one GPU, where RCCL refuses duplicate devices): staged through host memory."""
    import torch
    staged = dist.get_backend(group) != "nccl"

    def allreduce(ptr: int, count: int):
        t = torch.as_tensor(_DeviceDoubles(ptr, count), device=device)
        if staged:
            h = t.cpu()
            dist.all_reduce(h, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=group)
        torch.cuda.synchronize(device)
    return allreduce
