"""Time gsx_relinearize_partial + back-substitution against a full relinearize + factorize + back-substitution on the bench
workloads, for a few sizes of the moved set (the most recent poses / a random sample), with the full and with ISAM2's
partial ("wildfire") back-substitution.  python tools/partial_probe.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gtsam_petercdev_amd import _abi as A, _lib  # noqa: E402


def main():
    rows = []
    for name in ("pose3_100k", "pose2_100k", "bal1723"):
        arr, order = bench.make_problem(name, 42)
        kind = {"schur_nd": A.ORDER_SCHUR_ND, "nd": A.ORDER_ND}[order]
        be = _lib.ProductBackend(arr, device=0)
        be.set_ordering(be.compute_ordering(kind))
        off = np.concatenate([[0], np.cumsum(arr.state_dims())])

        def full():
            be.linearize()
            be.solve(0.0, False, want_delta=False)

        for _ in range(3):
            full()
        be.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            full()
        be.synchronize()
        t_full = (time.perf_counter() - t0) / 10
        poses = np.nonzero(arr.var_types != A.VAR_VECTOR)[0]
        rng = np.random.default_rng(1)
        for frac, pick in ((0.001, "recent"), (0.01, "recent"), (0.1, "recent"), (0.01, "random")):
            n = max(1, int(frac * poses.size))
            idx = np.sort(poses[-n:] if pick == "recent" else rng.choice(poses, n, replace=False))
            keys = arr.var_keys[idx]
            states = np.concatenate([arr.values[off[i]:off[i + 1]] for i in idx])   # (same values: the work is the same)
            full()
            stats = be.relinearize_partial(keys, states)
            be.solve(0.0, False, want_delta=False)
            be.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                be.relinearize_partial(keys, states)
                be.solve(0.0, False, want_delta=False)
            be.synchronize()
            t_part = (time.perf_counter() - t0) / 10
            be.set_profiling(0)
            be.reset_stats()
            be.relinearize_partial(keys, states)
            be.synchronize()
            st = be.stats()
            be.set_profiling(-1)
            dev_ms = st["ms_linearize"] + st["ms_assemble_hessian"] + st["ms_factorize"]
            # the same update followed by ISAM2's partial back-substitution (default wildfireThreshold 1e-3); the moved
            # states are nudged so that the solution does change and the change has to be chased down the tree
            nudged = states + 1e-4 * rng.standard_normal(states.size)
            be.relinearize_partial(keys, nudged)
            _, n_wf = be.backsubstitute_wildfire(1e-3, want_delta=False)
            be.synchronize()
            t0 = time.perf_counter()
            for k in range(10):
                be.relinearize_partial(keys, nudged if k % 2 else states)
                be.backsubstitute_wildfire(1e-3, want_delta=False)
            be.synchronize()
            t_wf = (time.perf_counter() - t0) / 10
            rows.append(dict(workload=name, moved=f"{pick} {n} poses", full_ms=1e3 * t_full, partial_ms=1e3 * t_part,
                             partial_wildfire_ms=1e3 * t_wf, wildfire_vars_solved=n_wf, n_vars=int(arr.n_vars),
                             partial_device_ms_without_backsolve=dev_ms, **stats))
            print(json.dumps(rows[-1]), flush=True)
        be.close()


if __name__ == "__main__":
    main()
