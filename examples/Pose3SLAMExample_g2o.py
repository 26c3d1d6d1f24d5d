#!/usr/bin/env python3
"""examples/Pose3SLAMExample_g2o.cpp of the reference on the MI355X backend.

    python examples/Pose3SLAMExample_g2o.py [g2oFile] [outputFile]

readG2o (3-D; also TORO VERTEX3 / EDGE3 files such as sphere2500.txt), prior Diagonal::Variances(1e-6 x3, 1e-4 x3) on the
first pose, Gauss-Newton, initial / final error, writeG2o."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_petercdev_amd import _abi as A, _lib  # noqa: E402

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main(argv):
    g2o_file = argv[1] if len(argv) > 1 else os.path.join(DATA, "pose3example.txt")
    arr = _lib.read_g2o(g2o_file, is3D=True)          # (the reader appends the example's prior on the first pose)
    print("Adding prior to g2o file ")
    be = _lib.ProductBackend(arr)
    be.set_ordering(be.compute_ordering(A.ORDER_ND if arr.n_vars > 2000 else A.ORDER_MINDEGREE))
    print("Optimizing the factor graph")
    r = be.gn_optimize(100)
    print("Optimization complete")
    print(f"initial error={r['initial_error']:.6g}")
    print(f"final error={r['final_error']:.6g}")
    if len(argv) > 2:
        print(f"Writing results to file: {argv[2]}")
        _lib.write_g2o(argv[2], arr, be.get_values())
        print("done! ")
    return r


if __name__ == "__main__":
    main(sys.argv)
