"""ISAM2's partial ("wildfire") back-substitution on the device (include/gsx.h:
gsx_backsubstitute_wildfire) against a literal restatement of the reference's traversal
(gtsam/nonlinear/ISAM2Clique.cpp:68-90 isDirty, :175-201 valuesChanged / restoreFromOriginals, :237-287
optimizeWildfireNode / optimizeWildfireNonRecursive; gtsam/nonlinear/ISAM2-impl.cpp:48-77) run in numpy on the ORACLE's
Bayes tree and its [R S d] conditionals at the same values.  The reference holds no known answers for the traversal
(its ISAM2 tests compare against batch solutions), so the expected numbers come from the restatement: parity of the
traversal itself is pinned by construction (same conditionals, same rule), not by a reference fixture."""
import numpy as np
import pytest

from gtsam_petercdev_amd import _abi as A
from gtsam_petercdev_amd import datasets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from gtsam_petercdev_amd import _lib
    assert _lib.device_count() > 0
    return _lib


def _state_slices(arr):
    return np.concatenate([[0], np.cumsum(arr.state_dims())])


def wildfire_restatement(parent, fronts, conditional, dims, delta, replaced, threshold):
    """optimizeWildfireNonRecursive (ISAM2Clique.cpp:261-287) over cliques given as (frontal variables, separator
    variables) with conditional(c) = [R S d].  `replaced` = set of cliques that were re-eliminated.  Returns the new delta
    and the number of frontal variables back-substituted."""
    toff = np.concatenate([[0], np.cumsum(dims)]).astype(np.int64)
    sl = lambda vs: np.concatenate([np.arange(toff[v], toff[v + 1]) for v in vs]) if len(vs) else np.zeros(0, np.int64)
    children = [[] for _ in parent]
    roots = []
    for c, p in enumerate(parent):
        (children[p] if p >= 0 else roots).append(c)
    delta = delta.copy()
    changed = set()
    count = 0
    stack = list(roots)
    while stack:
        c = stack.pop()
        fv, sv = fronts[c]
        dirty = (c in replaced) or any(s in changed for s in sv)       # ISAM2Clique::isDirty
        if not dirty:
            continue
        fi, si = sl(fv), sl(sv)
        RSd = conditional(c)
        nf = fi.size
        orig = delta[fi].copy()
        rhs = RSd[:, -1] - RSd[:, nf:nf + si.size] @ delta[si]
        sol = np.linalg.solve(RSd[:, :nf], rhs)                        # R upper triangular
        delta[fi] = sol
        count += len(fv)
        if (c in replaced) or np.max(np.abs(orig - sol)) >= threshold:  # valuesChanged
            changed.update(fv)
        else:
            delta[fi] = orig                                            # restoreFromOriginals
        stack.extend(children[c])
    return delta, count


def replaced_cliques(arr, parent, fronts, moved):
    """The cliques gsx_relinearize_partial re-eliminates: those holding a variable of a factor that touches a moved
    variable, and all their ancestors (ISAM2.cpp:725-783: the top of the tree that contains the marked keys)."""
    fv = arr.f_vars
    kp = arr.f_key_ptr
    moved = set(int(m) for m in moved)
    dirty_vars = set()
    for f in range(arr.n_factors):
        vs = fv[kp[f]:kp[f + 1]]
        if any(int(v) in moved for v in vs):
            dirty_vars.update(int(v) for v in vs)
    front_of = {}
    for c, (f, _s) in enumerate(fronts):
        for v in f:
            front_of[v] = c
    out = set()
    for v in dirty_vars:
        c = front_of[v]
        while c >= 0 and c not in out:
            out.add(c)
            c = parent[c]
    return out


def _problem(name):
    if name == "pose3":
        return datasets.synth_manhattan_pose3(3000, seed=9), A.ORDER_ND
    if name == "pose2":
        return datasets.synth_manhattan_pose2(2000, seed=5), A.ORDER_ND
    # (ring-local co-visibility: moving one camera re-eliminates its neighbourhood's cliques and their ancestors, not all)
    return datasets.synth_bal_arrays(200, 6000, 30000, seed=21, long_range=0.0), A.ORDER_SCHUR_ND


@pytest.mark.parametrize("name", ["pose3", "pose2", "bal"])
def test_wildfire_matches_the_reference_traversal(gpu, oracle, name):
    arr, kind = _problem(name)
    P, F = gpu.product_backend(arr), gpu.product_backend(arr)
    O = oracle.oracle_backend(arr)
    ordering = P.compute_ordering(kind)
    for be in (P, F):
        be.set_amalgamation(0.0, 128)      # the reference's cliques
        be.set_ordering(ordering)
    O.set_ordering(ordering)
    with pytest.raises(A.GsxError) as ei:
        P.backsubstitute_wildfire(1e-3)    # nothing resident yet
    assert ei.value.status == A.GSX_E_STATE
    P.linearize()
    d_prev = P.solve(0.0, False)
    F.linearize()
    F.solve(0.0, False)
    F.retract(None, commit=True)
    x1 = F.get_values()                    # where a Gauss-Newton step moves everything
    off = _state_slices(arr)
    current = arr.values.copy()
    dims = arr.var_dims
    if name == "bal":
        moves = [np.array([3]), np.array([101, 102])]                     # cameras
    else:
        moves = [np.arange(arr.n_vars - 10, arr.n_vars), np.arange(arr.n_vars // 2, arr.n_vars // 2 + 12)]
    seen_partial = False
    for it, idx in enumerate(moves):
        states = np.concatenate([x1[off[i]:off[i + 1]] for i in idx])
        for i in idx:
            current[off[i]:off[i + 1]] = x1[off[i]:off[i + 1]]
        stats = P.relinearize_partial(arr.var_keys[idx], states)
        # the oracle at the same values: its tree, its conditionals, its full solution
        O.set_values(current)
        O.linearize()
        d_full = O.solve(0.0, False)
        po, fo = O.get_tree()
        rep = replaced_cliques(arr, po, fo, idx)
        if stats["n_fronts_reeliminated"] == stats["n_fronts"]:
            rep = set(range(len(fo)))      # (the call took its full path)
        else:
            assert len(rep) == stats["n_fronts_reeliminated"], (len(rep), stats)
        scale = max(np.abs(d_full).max(), 1e-300)
        # a threshold in the range of the solution's own changes: some cliques pass it on, some stop it
        thr = 0.02 * float(np.max(np.abs(d_full - d_prev))) if it == 0 else 1e-3 * scale
        d_exp, n_exp = wildfire_restatement(po, fo, O.conditional, dims, d_prev, rep, thr)
        d_wf, n_wf = P.backsubstitute_wildfire(thr)
        assert n_wf == n_exp, (name, it, n_wf, n_exp)
        # (1e-7: the long Pose2 chain's undamped Gauss-Newton system is ill-conditioned — steps of 1e2 — and the device and
        #  the oracle already differ by 1e-8 relative on its plain solve)
        assert np.linalg.norm(d_wf - d_exp) <= 1e-7 * np.linalg.norm(d_exp), (name, it, float(np.abs(d_wf - d_exp).max()), scale)
        assert len(rep) <= n_wf <= arr.n_vars
        seen_partial |= n_wf < arr.n_vars
        # what the threshold left unpropagated is of the threshold's order
        assert np.abs(d_wf - d_full).max() <= 50 * thr + 1e-8 * scale
        d_prev = d_wf
    assert seen_partial, "no run stopped anywhere: the thresholds of this test are too small to test anything"
    # threshold <= 0: everything, bit for bit the plain back-substitution of the same factorization
    idx = np.arange(5)
    states = np.concatenate([arr.values[off[i]:off[i + 1]] for i in idx])
    for i in idx:
        current[off[i]:off[i + 1]] = arr.values[off[i]:off[i + 1]]
    P.relinearize_partial(arr.var_keys[idx], states)
    d_all, n_all = P.backsubstitute_wildfire(0.0)
    assert n_all == arr.n_vars
    F.set_values(current)
    F.linearize()
    assert np.array_equal(d_all, F.solve(0.0, False))
    # and nothing replaced + nothing changed: nothing is visited below the roots' first clean cliques
    d_same, n_same = P.backsubstitute_wildfire(1e-3)
    assert n_same == 0 and np.array_equal(d_same, d_all)


def test_wildfire_on_a_relaxed_tree(gpu):
    """With relaxed amalgamation a front is visited or skipped as a whole: the restatement runs on the product's own tree
    and conditionals (gsx_get_tree / gsx_get_conditional), which the boundary tests pin against the oracle."""
    arr, kind = _problem("pose3")
    P = gpu.product_backend(arr)
    P.set_amalgamation(0.5, 32)
    P.set_ordering(P.compute_ordering(kind))
    P.linearize()
    d_prev = P.solve(0.0, False)
    off = _state_slices(arr)
    rng = np.random.default_rng(3)
    idx = np.arange(arr.n_vars - 8, arr.n_vars)
    states = np.concatenate([arr.values[off[i]:off[i + 1]] for i in idx])
    states = states + 1e-3 * rng.standard_normal(states.size)   # (rotation blocks drift off SO(3) by 1e-3: fine for a linear test)
    stats = P.relinearize_partial(arr.var_keys[idx], states)
    pg, fg = P.get_tree()
    rep = replaced_cliques(arr, pg, fg, idx)
    assert len(rep) == stats["n_fronts_reeliminated"]
    thr = 1e-4 * max(np.abs(d_prev).max(), 1e-300)
    d_exp, n_exp = wildfire_restatement(pg, fg, P.conditional, arr.var_dims, d_prev, rep, thr)
    d_wf, n_wf = P.backsubstitute_wildfire(thr)
    assert n_wf == n_exp and 0 < n_wf < arr.n_vars
    assert np.abs(d_wf - d_exp).max() <= 1e-8 * max(np.abs(d_exp).max(), 1e-300)
