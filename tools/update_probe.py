"""gsx_update at scale (VisualISAM2Example-style growth): a visual-SLAM graph of K keyframes and M
landmarks, already optimised on the handle up to keyframe K-2; the probe times adding keyframe K-1 (its pose, the landmarks
that become observable with it, its projection and odometry factors) and the solve that follows, next to a full
re-linearization + solve of the same graph.

  python tools/update_probe.py [K] [M] [OBS]        # default 2000 200000 900000; the config names 10000 1000000 4500000
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gtsam_petercdev_amd import _abi as A, _lib, datasets   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
OBS = int(sys.argv[3]) if len(sys.argv) > 3 else 900000

t0 = time.perf_counter()
a1, id1 = datasets.synth_visual_slam(K, M, OBS, upto=K - 1)
a2, id2 = datasets.synth_visual_slam(K, M, OBS)
t_gen = time.perf_counter() - t0
pos = {int(i): k for k, i in enumerate(id1)}
origin = np.array([pos.get(int(i), -1) for i in id2], dtype=np.int32)
so, old = a2.state_offsets(), set(int(k) for k in a1.var_keys)
new_states = np.concatenate([a2.values[so[i]:so[i + 1]] for i, k in enumerate(a2.var_keys) if int(k) not in old])

pb = _lib.product_backend(a1)
t0 = time.perf_counter()
pb.set_ordering(pb.compute_ordering(A.ORDER_SCHUR_ND))
t_sym0 = time.perf_counter() - t0
p = A.lm_params_legacy()
p.max_iterations = 3
r0 = pb.lm_optimize(p)                     # the estimate before the new keyframe arrives
pb.linearize()
pb.solve(0.0, False, want_delta=False)
pb.synchronize()

t0 = time.perf_counter()
st = pb.update(a2, origin, new_states)
pb.synchronize()
t_update = time.perf_counter() - t0
t0 = time.perf_counter()
pb.solve(0.0, False, want_delta=False)     # H assembly + elimination of the whole tree + back-substitution
pb.synchronize()
t_solve = time.perf_counter() - t0
t0 = time.perf_counter()
pb.linearize()
pb.solve(0.0, False, want_delta=False)     # the same with every factor re-linearized
pb.synchronize()
t_full = time.perf_counter() - t0
# ---- the relinearization step of an iSAM2 update on the same handle (config 5's "iSAM2 relinearize + partial Bayes-tree
#      re-elimination"): the R most recent keyframes move, their factors are re-linearized, the cliques that hold them and
#      their ancestors re-eliminated (gsx_relinearize_partial), then ISAM2's partial back-substitution (wildfire, 1e-3)
partial = []
so2 = a2.state_offsets()
pose_idx = np.nonzero(a2.var_types == A.VAR_POSE3)[0]
rng = np.random.default_rng(5)
for R in (1, 10, 100):
    idx = pose_idx[-R:]
    keys = a2.var_keys[idx]
    cur = pb.get_values()
    base = np.concatenate([cur[so2[i]:so2[i + 1]] for i in idx])
    nudged = base.copy()
    for k in range(R):                      # translations only (the rotation block stays orthonormal)
        nudged[12 * k + 9:12 * k + 12] += 1e-3 * rng.standard_normal(3)
    pb.relinearize_partial(keys, nudged)    # warm the scratch tables
    pb.backsubstitute_wildfire(1e-3, want_delta=False)
    pb.synchronize()
    t_p, t_w, n_wf, stp = 0.0, 0.0, 0, None
    reps = 6
    for k in range(reps):
        t0 = time.perf_counter()
        stp = pb.relinearize_partial(keys, base if k % 2 == 0 else nudged)
        pb.synchronize()
        t1 = time.perf_counter()
        _, n_wf = pb.backsubstitute_wildfire(1e-3, want_delta=False)
        pb.synchronize()
        t2 = time.perf_counter()
        t_p += t1 - t0
        t_w += t2 - t1
    t0 = time.perf_counter()
    pb.solve(0.0, False, want_delta=False)  # the plain back-substitution of the same factorization, for comparison
    pb.synchronize()
    t_bs = time.perf_counter() - t0
    partial.append(dict(moved_keyframes=R, relinearize_partial_ms=1e3 * t_p / reps, wildfire_backsub_ms=1e3 * t_w / reps,
                        wildfire_vars_solved=n_wf, plain_backsub_ms=1e3 * t_bs, **stp))
s = pb.stats()
print(json.dumps({
    "probe": "gsx_update", "keyframes": K, "landmarks": int(a2.meta["n_landmarks"]), "projection_factors": int(a2.meta["n_obs"]),
    "variables": a2.n_vars, "factors": a2.n_factors, "initial_ordering_and_symbolic_s": t_sym0,
    "error_before_after_3_lm_iterations": [r0["initial_error"], r0["final_error"]],
    "update": dict(st, wall_s=t_update), "solve_after_update_ms": 1e3 * t_solve, "relinearize_all_and_solve_ms": 1e3 * t_full,
    "isam2_relinearization_step": partial,
    "tree": {k: s[k] for k in ("n_fronts", "n_levels", "max_front_dim", "factor_flops")},
    "generate_s": t_gen}))
