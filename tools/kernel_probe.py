"""Diagnostic: a few factorizations of a bench workload, errors ignored (for timing-only experiment builds); run under
rocprofv3 --kernel-trace --stats to read per-kernel durations."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gtsam_petercdev_amd import _lib, _abi as A
name = sys.argv[1] if len(sys.argv) > 1 else "bal1723"
arrays, order = bench.make_problem(name, 42)
pb = _lib.product_backend(arrays)
pb.set_ordering(pb.compute_ordering({"nd": A.ORDER_ND, "schur_nd": A.ORDER_SCHUR_ND}[order]))
pb.linearize()
for _ in range(5):
    try:
        pb.solve(1e-3, False)
    except Exception as e:   # noqa: BLE001 — experiment builds may produce garbage
        print("solve:", type(e).__name__)
