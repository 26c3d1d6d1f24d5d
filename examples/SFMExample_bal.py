#!/usr/bin/env python3
"""examples/SFMExample_bal.cpp of the reference on the MI355X backend.

    python examples/SFMExample_bal.py [balFile] [outputFile]

SfmData::FromBalFile, one GeneralSFMFactor<SfmCamera, Point3> per measurement with Isotropic(2, 1.0), priors
Isotropic(9, 0.1) on camera 0 and Isotropic(3, 0.1) on point 0, Levenberg-Marquardt (default parameters, Schur ordering of
the landmarks first as SFMExample_bal_COLAMD_METIS.cpp's constrained orderings do), final error; optionally writeBAL."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gtsam_petercdev_amd import _abi as A, _lib  # noqa: E402

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main(argv):
    filename = argv[1] if len(argv) > 1 else os.path.join(DATA, "dubrovnik-3-7-pre.txt")
    arr = _lib.read_bal(filename, priors=True)
    n_cams = int((arr.var_types == A.VAR_CAMERA).sum())
    print(f"read {arr.n_vars - n_cams} tracks on {n_cams} cameras")
    be = _lib.ProductBackend(arr)
    be.set_ordering(be.compute_ordering(A.ORDER_SCHUR_ND))
    r = be.lm_optimize(A.lm_params_legacy())
    print(f"final error: {r['final_error']:.6g}")
    if len(argv) > 2:
        _lib.write_bal(argv[2], arr, be.get_values())
    return r


if __name__ == "__main__":
    main(sys.argv)
