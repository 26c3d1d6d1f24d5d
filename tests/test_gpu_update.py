"""gsx_update — ISAM2::update's structural part on a live handle (gtsam/nonlinear/ISAM2.cpp:395-484): a visual-SLAM
graph grown keyframe by keyframe as examples/VisualISAM2Example.cpp:88-131 does (a pose, the landmarks it sees for the
first time, their projection factors; priors on the first pose and landmark), with factors removed and re-added on the way.
After every update the handle must solve exactly what a fresh handle on the same graph, values and ordering solves, and
the final estimate must be the batch optimum of the final graph (oracle)."""
import math

import numpy as np
import pytest

from gtsam_petercdev_amd import _abi as A
from gtsam_petercdev_amd.graph import (X, L, Pose3, Rot3, Point3, Values, NonlinearFactorGraph, PriorFactor,
                                       BetweenFactor, GenericProjectionFactor, Cal3_S2, noiseModel)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from gtsam_petercdev_amd import _lib
    assert _lib.device_count() > 0
    return _lib


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def _scene(n_poses, n_points, seed):
    rng = np.random.default_rng(seed)
    K = Cal3_S2(50.0, 50.0, 0.0, 50.0, 50.0)
    pts = rng.uniform(-2, 2, (n_points, 3))
    poses = []
    for i in range(n_poses):
        th = 2 * math.pi * i / n_poses
        t = np.array([8 * math.cos(th), 8 * math.sin(th), 1.0])
        zc = -t / np.linalg.norm(t)
        xc = np.cross([0, 0, 1.0], zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        poses.append((np.stack([xc, yc, zc], axis=1), t))
    return rng, K, pts, poses


def _grow(n_poses=9, n_points=60, seed=21, visible=0.55, hard_prior=False):
    """Yields (factors, values-dict) after each keyframe: factor objects are appended, so a factor keeps its identity."""
    rng, K, pts, poses = _scene(n_poses, n_points, seed)
    noise = noiseModel.Isotropic.Sigma(2, 1.0)
    odo = noiseModel.Diagonal.Sigmas([0.05] * 3 + [0.1] * 3)
    factors, init, pending = [], {}, {}
    for i, (R, t) in enumerate(poses):
        init[X(i)] = Pose3(Rot3(R), Point3(*(t + rng.normal(0, 0.05, 3))))
        if i == 0:
            # (hard_prior: the first pose is KNOWN — noiseModel::Constrained::All(6), a clique with constraint pivots)
            factors.append(PriorFactor(X(0), Pose3(Rot3(R), Point3(*t)),
                                       noiseModel.Constrained.All(6) if hard_prior
                                       else noiseModel.Diagonal.Sigmas([0.1] * 3 + [0.3] * 3)))
        else:
            Rp, tp = poses[i - 1]
            factors.append(BetweenFactor(X(i - 1), X(i), Pose3(Rot3(Rp.T @ R), Point3(*(Rp.T @ (t - tp)))), odo))
        for j in range(n_points):
            q = R.T @ (pts[j] - t)
            if q[2] <= 0.5 or rng.random() > visible:
                continue
            uv = np.array([K.v[0] * q[0] / q[2] + K.v[3], K.v[1] * q[1] / q[2] + K.v[4]]) + rng.normal(0, 1.0, 2)
            f = GenericProjectionFactor(uv, noise, X(i), L(j), K)
            if L(j) in init:
                factors.append(f)
            elif j in pending:   # second sighting: the landmark is observable — it enters with both its factors
                init[L(j)] = Point3(*(pts[j] + rng.normal(0, 0.05, 3)))
                factors.extend([pending.pop(j), f])
            else:
                pending[j] = f
        yield list(factors), dict(init)


def _arrays(factors, init):
    g, v = NonlinearFactorGraph(), Values()
    for f in factors:
        g.add(f)
    for k in sorted(init):
        v.insert(k, init[k])
    return g.to_arrays(v)


def _new_states(arr, old_keys):
    so = arr.state_offsets()
    old = set(int(k) for k in old_keys)
    return np.concatenate([arr.values[so[i]:so[i + 1]] for i, k in enumerate(arr.var_keys) if int(k) not in old]
                          or [np.zeros(0)])


@pytest.mark.parametrize("hard_prior", [False, True])
def test_grow_keyframe_by_keyframe_matches_fresh_handles_and_the_batch_optimum(gpu, oracle, hard_prior):
    steps = list(_grow(hard_prior=hard_prior))
    cur_factors, cur_init = steps[1]            # VisualISAM2Example starts its first update with two poses
    arr = _arrays(cur_factors, cur_init)
    gb = gpu.product_backend(arr)
    gb.set_amalgamation(0.25, 64)
    gb.set_ordering(gb.compute_ordering(A.ORDER_SCHUR_ND))
    gb.linearize()
    gb.solve(0.0, False)
    assert gb.stats()["n_constraint_rows"] == (6 if hard_prior else 0)
    p1 = A.lm_params_legacy()
    p1.max_iterations = 1
    dropped = None
    moved = False       # the values changed since the handle's last linearization
    n_kept_checks = 0
    for k in range(2, len(steps)):
        new_factors, new_init = steps[k]
        new_factors = list(new_factors)
        if k == 4:      # removeFactorIndices: drop one projection factor of an old pose ...
            seen = {}
            for f in new_factors:
                if f.ftype == A.F_PROJECTION:
                    seen[f.keys_[1]] = seen.get(f.keys_[1], 0) + 1
            dropped = next(f for f in new_factors        # (of a landmark that stays well observed without it)
                           if f.ftype == A.F_PROJECTION and f.keys_[0] == X(1) and seen[f.keys_[1]] >= 4)
            new_factors.remove(dropped)
        elif dropped is not None and k < 6:
            new_factors.remove(dropped)
        # (from k = 6 on the dropped factor is back: it is a NEW factor for the handle)
        origin = [next((i for i, g in enumerate(cur_factors) if g is f), -1) for f in new_factors]
        arr2 = _arrays(new_factors, new_init)
        st = gb.update(arr2, origin, _new_states(arr2, arr.var_keys))
        assert st["n_factors_added"] == sum(1 for o in origin if o < 0)
        assert st["n_factors_removed"] == len(cur_factors) - sum(1 for o in origin if o >= 0)
        assert st["n_vars_added"] == arr2.n_vars - arr.n_vars and st["n_vars_affected"] >= st["n_vars_added"]
        # the same graph, values and ordering on a fresh handle
        fresh = _arrays(new_factors, new_init)
        fresh.values = gb.get_values()
        fb = gpu.product_backend(fresh)
        fb.set_amalgamation(0.25, 64)
        fb.set_ordering(gb.get_ordering())
        assert abs(gb.error() - fb.error()) <= 1e-12 * abs(fb.error())
        fb.linearize()
        if moved:
            gb.linearize()          # the linearization point moved: everything is re-linearized, as on any handle
        else:
            n_kept_checks += 1      # kept factors keep their [A b] (copied, NOT re-linearized); only the new ones are computed
        assert np.max(np.abs(gb.jacobians() - fb.jacobians())) <= 1e-12 * np.max(np.abs(fb.jacobians())), k
        d_upd, d_new = gb.solve(0.0, False), fb.solve(0.0, False)
        assert relerr(d_upd, d_new) < 1e-9, (k, relerr(d_upd, d_new))
        fb.close()
        moved = False
        if k % 2 == 1:
            gb.lm_optimize(p1)      # one step on the live handle (ISAM2::update ends with one, ISAM2.cpp:466-476)
            moved = True
        cur_factors, arr = new_factors, arr2
    assert n_kept_checks >= 3
    # the final graph: the live handle converges to the batch optimum the oracle finds from the initial values
    p = A.lm_params_legacy()
    p.max_iterations, p.relative_error_tol, p.absolute_error_tol = 60, 1e-13, 1e-13
    rg = gb.lm_optimize(p)
    ob = oracle.oracle_backend(_arrays(cur_factors, steps[-1][1]))
    ob.set_ordering(gb.get_ordering())
    ro = ob.lm_optimize(p)
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"]
    assert relerr(gb.get_values(), ob.get_values()) < 1e-5


def test_update_argument_checks(gpu):
    steps = list(_grow(n_poses=4, n_points=20, seed=3))
    f1, i1 = steps[1]
    arr = _arrays(f1, i1)
    gb = gpu.product_backend(arr)
    f2, i2 = steps[2]
    arr2 = _arrays(f2, i2)
    origin = list(range(len(f1))) + [-1] * (len(f2) - len(f1))
    from gtsam_petercdev_amd import GsxError
    with pytest.raises(GsxError):            # no ordering yet
        gb.update(arr2, origin, _new_states(arr2, arr.var_keys))
    gb.set_ordering(gb.compute_ordering(A.ORDER_SCHUR_ND))
    with pytest.raises(GsxError):            # wrong number of new states
        gb.update(arr2, origin, _new_states(arr2, arr.var_keys)[:-1])
    bad = list(origin)
    bad[1] = bad[0]
    with pytest.raises(GsxError):            # two factors claim the same origin
        gb.update(arr2, bad, _new_states(arr2, arr.var_keys))
    gb.update(arr2, origin, _new_states(arr2, arr.var_keys))   # the handle is still usable
    assert gb.error() > 0


def test_failed_update_poisons_the_handle_instead_of_leaving_mismatched_tables(gpu, monkeypatch):
    """a failure in the middle of gsx_update's switch (injected here where an allocation would fail at
    config-5 sizes) must not leave a handle that still claims an ordering / a factorization for tables of two different
    graphs: every later numeric call is refused (GSX_E_STATE) until values and ordering are set again — and then it works."""
    from gtsam_petercdev_amd import GsxError
    steps = list(_grow(n_poses=4, n_points=20, seed=5))
    f1, i1 = steps[1]
    arr = _arrays(f1, i1)
    gb = gpu.product_backend(arr)
    gb.set_ordering(gb.compute_ordering(A.ORDER_SCHUR_ND))
    gb.linearize()
    gb.solve(1e-3, False)
    f2, i2 = steps[2]
    arr2 = _arrays(f2, i2)
    origin = list(range(len(f1))) + [-1] * (len(f2) - len(f1))
    monkeypatch.setenv("GSX_INJECT_UPDATE_FAILURE", "1")
    with pytest.raises(GsxError) as e:
        gb.update(arr2, origin, _new_states(arr2, arr.var_keys))
    assert e.value.status == A.GSX_E_NOMEM
    monkeypatch.delenv("GSX_INJECT_UPDATE_FAILURE")
    for call in (lambda: gb.solve(1e-3, False), gb.linearize, gb.error, lambda: gb.marginal_covariance(int(arr.var_keys[0]))):
        with pytest.raises(GsxError) as e:
            call()
        assert e.value.status == A.GSX_E_STATE
    # the handle now holds the NEW problem (the wrapper followed it): values + ordering make it whole again
    assert gb.arrays is arr2
    gb.set_values(arr2.values)
    gb.set_ordering(gb.compute_ordering(A.ORDER_SCHUR_ND))
    fresh = gpu.product_backend(arr2)
    fresh.set_ordering(gb.get_ordering())
    gb.linearize()
    fresh.linearize()
    assert np.array_equal(gb.solve(1e-3, False), fresh.solve(1e-3, False))


@pytest.mark.parametrize("seed", range(6))
def test_update_on_random_growth(gpu, seed):
    """Structure fuzz: a random Pose2 graph (chain, chords, a hub) revealed a few variables at a time, with random factors
    removed and put back on the way; after every gsx_update the handle equals a fresh handle on the same graph, values and
    ordering (Jacobians to 1e-12, Gauss-Newton step to 1e-9), also after a partial relinearization on top of it."""
    from gtsam_petercdev_amd.graph import Pose2
    rng = np.random.default_rng(700 + seed)
    nv = int(rng.choice([20, 60, 120]))
    pos = np.cumsum(rng.normal(0.5, 0.2, (nv, 2)), axis=0)
    th = np.cumsum(rng.normal(0, 0.1, nv))
    init = {k: Pose2(pos[k, 0] + rng.normal(0, 0.03), pos[k, 1] + rng.normal(0, 0.03), th[k] + rng.normal(0, 0.01))
            for k in range(nv)}
    pairs = {(k, k + 1) for k in range(nv - 1)}
    for a, b in rng.integers(0, nv, (nv, 2)):
        if a != b:
            pairs.add((int(min(a, b)), int(max(a, b))))
    hub = int(rng.integers(0, nv // 2))
    for b in rng.choice(nv, size=min(nv - 1, 25), replace=False):
        if int(b) != hub:
            pairs.add((min(hub, int(b)), max(hub, int(b))))
    sig = noiseModel.Diagonal.Sigmas(np.array([0.2, 0.2, 0.1]))
    all_factors = [PriorFactor(0, Pose2(pos[0, 0], pos[0, 1], th[0]), noiseModel.Isotropic.Sigma(3, 0.1))]
    for a, b in sorted(pairs, key=lambda ab: ab[1]):
        c, s_ = math.cos(th[a]), math.sin(th[a])
        dx, dy = pos[b, 0] - pos[a, 0], pos[b, 1] - pos[a, 1]
        all_factors.append(BetweenFactor(a, b, Pose2(c * dx + s_ * dy + rng.normal(0, 0.05), -s_ * dx + c * dy + rng.normal(0, 0.05),
                                                     th[b] - th[a] + rng.normal(0, 0.02)), sig))
    def active(n_known, removed):
        return [f for f in all_factors if max(f.keys_) < n_known and id(f) not in removed]
    n_known = max(4, nv // 5)
    removed = set()
    cur = active(n_known, removed)
    arr = _arrays(cur, {k: init[k] for k in range(n_known)})
    gb = gpu.product_backend(arr)
    amalg = (0.0, 128) if seed % 2 else (0.5, 64)   # (explicit on both sides: the library's own choice is made per graph)
    gb.set_amalgamation(*amalg)
    gb.set_ordering(gb.compute_ordering([A.ORDER_ND, A.ORDER_MINDEGREE][seed % 2]))
    gb.linearize()
    gb.solve(0.0, False)
    step = 0
    while n_known < nv:
        step += 1
        n_known = min(nv, n_known + int(rng.integers(1, 8)))
        if step % 3 == 1 and len(cur) > 10:        # drop two between factors that are not chain links
            cand = [f for f in cur if f.ftype != A.F_PRIOR and abs(f.keys_[0] - f.keys_[1]) > 1]
            for f in list(rng.choice(cand, size=min(2, len(cand)), replace=False)) if cand else []:
                removed.add(id(f))
        elif step % 3 == 0:
            removed.clear()                            # ... and put everything back
        new = active(n_known, removed)
        origin = [next((i for i, g in enumerate(cur) if g is f), -1) for f in new]
        arr2 = _arrays(new, {k: init[k] for k in range(n_known)})
        st = gb.update(arr2, origin, _new_states(arr2, arr.var_keys))
        assert st["n_vars_added"] == arr2.n_vars - arr.n_vars
        fresh = _arrays(new, {k: init[k] for k in range(n_known)})
        fresh.values = gb.get_values()
        fb = gpu.product_backend(fresh)
        fb.set_amalgamation(*amalg)
        fb.set_ordering(gb.get_ordering())
        fb.linearize()
        assert np.max(np.abs(gb.jacobians() - fb.jacobians())) <= 1e-12 * np.max(np.abs(fb.jacobians())), (seed, step)
        assert relerr(gb.solve(0.0, False), fb.solve(0.0, False)) < 1e-9, (seed, step)
        # an iSAM2 relinearization step on top of the updated handle
        idx = np.arange(max(0, n_known - 3), n_known)
        so = arr2.state_offsets()
        vals = gb.get_values()
        states = np.concatenate([vals[so[i]:so[i + 1]] + rng.normal(0, 0.01, 3) for i in idx])
        gb.relinearize_partial(arr2.var_keys[idx], states)
        dp, _ = gb.backsubstitute_wildfire(0.0)
        fb.set_values(gb.get_values())
        fb.linearize()
        assert np.array_equal(dp, fb.solve(0.0, False)), (seed, step)
        fb.close()
        cur, arr = new, arr2
    assert step >= 3
