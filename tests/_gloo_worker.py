"""Worker of tests/test_distributed_gloo.py: the N>1 code path of bench.py on CPU (gloo)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_petercdev_amd import _abi as A, _lib, datasets, distributed as D  # noqa: E402

rank, local_rank, world = D.env_rank()
dist = D.init("gloo")
arr = datasets.synth_bal_arrays(12, 300, 1200, seed=D.replica_seed(42), long_range=0.3)
be = _lib.ProductBackend(arr, host_only=True)          # host side only: no GPU in this test
ordering = be.compute_ordering(A.ORDER_SCHUR)
be.set_ordering(ordering)
st = be.stats()
dist.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (rank + 1))                          # ranks finish at different times
dist.barrier()
elapsed = time.perf_counter() - t0 + 0.01 * rank
value, ms = D.aggregate_throughput(dist, 4, elapsed)
import torch
g = [None] * world
dist.all_gather_object(g, (rank, int(arr.values.size), float(arr.values[:50].sum()), st["n_fronts"], elapsed))
if rank == 0:
    print(json.dumps({"value": value, "ms_per_step": ms, "world": world, "ranks": g}))
dist.destroy_process_group()
