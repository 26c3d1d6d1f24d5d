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


def torch_allreduce(dist, device, group=None, bounce=False):
    """allreduce(ptr, count) for ProductBackend.set_shard on top of torch.distributed.  backend "nccl": RCCL sums the
    library's own buffer in place (over xGMI between the GPUs of a node); backend "gloo" (rehearsals with several ranks
    on one GPU, where RCCL refuses duplicate devices): staged through host memory.  bounce=True sums a torch-allocated
    copy instead of the library's buffer (the fallback checked_allreduce takes when the in-place path misbehaves)."""
    import torch
    staged = dist.get_backend(group) != "nccl"

    def allreduce(ptr: int, count: int):
        t = torch.as_tensor(_DeviceDoubles(ptr, count), device=device)
        if staged:
            h = t.cpu()
            dist.all_reduce(h, group=group)
            t.copy_(h)
        elif bounce:
            b = t.clone()
            dist.all_reduce(b, group=group)
            t.copy_(b)
        else:
            dist.all_reduce(t, group=group)
        torch.cuda.synchronize(device)
    return allreduce


def checked_allreduce(dist, device, group=None, probe_ptr=None, probe_count=0):
    """torch_allreduce, after a known-answer run of the in-place path on every rank (a buffer of rank + 1 must come back
    as world (world + 1) / 2 everywhere).  The hazard is RCCL reducing in place on memory that torch did NOT allocate
    (the library's hipMalloc'd arena, aliased through __cuda_array_interface__): pass `probe_ptr` / `probe_count` — a
    device range owned by the library that holds nothing yet, e.g. ProductBackend.shard_probe_buffer() — and the probe runs
    on exactly that memory; without it the probe is a torch tensor and only checks the collective itself.  If any rank
    sees a wrong sum, ALL ranks switch to the bounce-buffer variant together (the verdict travels through an ordinary
    tensor).  An EXCEPTION inside a collective is fatal for the group — the peers are still inside it — and is re-raised:
    the launcher (torch.distributed.run) tears the job down.  Returns (allreduce, "in_place" | "bounce" | "staged")."""
    import torch
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) != "nccl":
        return torch_allreduce(dist, device, group), "staged"
    if probe_ptr is not None and probe_count > 0:
        n = int(min(probe_count, 1024))
        view = torch.as_tensor(_DeviceDoubles(int(probe_ptr), n), device=device)
        view.fill_(float(rank + 1))
        torch.cuda.synchronize(device)
        torch_allreduce(dist, device, group)(int(probe_ptr), n)
        ok = int(bool(torch.all(view == world * (world + 1) / 2).item()))
        view.zero_()
    else:
        probe = torch.full((1024,), float(rank + 1), dtype=torch.float64, device=device)
        torch.cuda.synchronize(device)
        torch_allreduce(dist, device, group)(probe.data_ptr(), probe.numel())
        ok = int(bool(torch.all(probe == world * (world + 1) / 2).item()))
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 1:
        return torch_allreduce(dist, device, group), "in_place"
    return torch_allreduce(dist, device, group, bounce=True), "bounce"
