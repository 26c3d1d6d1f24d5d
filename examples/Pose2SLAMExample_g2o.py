#!/usr/bin/env python3
"""examples/Pose2SLAMExample_g2o.cpp of the reference on the MI355X backend: same arguments, same steps, same prints.

    python examples/Pose2SLAMExample_g2o.py [g2oFile] [outputFile] [maxIterations] [none|huber|tukey]

readG2o (2-D: load2D with the g2o information layout and the chosen robust kernel, dataset.cpp:620-633), prior
Diagonal::Variances(1e-6, 1e-6, 1e-8) on pose 0, Gauss-Newton, initial / final error, writeG2o of the result."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gtsam_petercdev_amd import _abi as A, _lib  # noqa: E402

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main(argv):
    g2o_file = argv[1] if len(argv) > 1 else os.path.join(DATA, "noisyToyGraph.txt")
    max_iterations = int(argv[3]) if len(argv) > 3 else 100
    kernel_type = argv[4] if len(argv) > 4 else "none"
    kernel = {"none": 0, "huber": 1, "tukey": 2}[kernel_type]
    if kernel:
        print(f"Using robust kernel: {kernel_type} ")
    arr = _lib.load2d(g2o_file, noise_format=A.NOISE_FORMAT_G2O, kernel=kernel)
    i0 = int(np.nonzero(arr.var_keys == 0)[0][0])
    arr = arr.with_factor(A.F_PRIOR, [i0], 3, [0.0, 0.0, 0.0], A.NOISE_DIAGONAL, np.sqrt([1e-6, 1e-6, 1e-8]))
    print("Adding prior on pose 0 ")
    if len(argv) > 3:
        print(f"User required to perform maximum  {max_iterations} iterations ")
    print("Optimizing the factor graph")
    be = _lib.ProductBackend(arr)
    be.set_ordering(be.compute_ordering(A.ORDER_MINDEGREE))
    r = be.gn_optimize(max_iterations)
    print("Optimization complete")
    print(f"initial error={r['initial_error']:.6g}")
    print(f"final error={r['final_error']:.6g}")
    result = be.get_values()
    if len(argv) < 3:
        for k, p in zip(arr.var_keys, result.reshape(-1, 3)):
            print(f"Value {int(k)}: (gtsam::Pose2) ({p[0]:.6g}, {p[1]:.6g}, {p[2]:.6g})")
    else:
        print(f"Writing results to file: {argv[2]}")
        plain = _lib.load2d(g2o_file, noise_format=A.NOISE_FORMAT_G2O)   # the graph without kernel, as the reference
        _lib.write_g2o(argv[2], plain, result)
        print("done! ")
    return r


if __name__ == "__main__":
    main(sys.argv)
