"""Diagnostic: phase stamps of front_small (build with GSX_STAMP=1): one factorization of a bench workload."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gtsam_petercdev_amd import _lib, _abi as A
name = sys.argv[1] if len(sys.argv) > 1 else "pose3_100k"
arrays, order = bench.make_problem(name, 42)
pb = _lib.product_backend(arrays)
pb.set_ordering(pb.compute_ordering({"nd": A.ORDER_ND, "schur_nd": A.ORDER_SCHUR_ND}[order]))
pb.linearize()
pb.solve(1e-3, False)
print("---- second factorization")
pb.solve(1e-3, False)
