#!/usr/bin/env python3
"""Generates tests/golden/metis_perm_<workload>_seed42.npy — the elimination order the REFERENCE's own METIS 5
(oracle/_ref/libmetis_ref.so, compiled by oracle/Makefile from gtsam/3rdparty/metis where it lies) returns for
the seeded full-size bench problems through Ordering::Metis / MetisIndex (oracle/oracle.py::metis_ordering,
gtsam/inference/Ordering.cpp:211-251, MetisIndex-inl.h:27-82), on the WHOLE graph as
examples/SFMExample_bal_COLAMD_METIS.cpp:83-117 does.

Stored as int32 positions into the problem's ascending-key variable table (ordering = var_keys[perm]); the
reference tree is needed to build libmetis_ref.so, so this runs in the build container only.

    python tests/golden/make_metis_orderings.py [workload ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    names = sys.argv[1:] or ["bal1723", "pose3_100k", "pose2_100k", "bal49"]
    for wl in names:
        arr, _ = bench.make_problem(wl, 42)
        keys = oracle.metis_ordering(arr)
        perm = np.searchsorted(arr.var_keys, keys).astype(np.int32)
        assert np.array_equal(arr.var_keys[perm], keys) and np.array_equal(np.sort(perm), np.arange(arr.n_vars))
        np.save(os.path.join(HERE, f"metis_perm_{wl}_seed42.npy"), perm)
        print(wl, perm.size, "variables")


if __name__ == "__main__":
    main()
