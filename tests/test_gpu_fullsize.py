"""Parity at BASELINE.json's full sizes (GPU): one LM inner iteration of the HIP path against the CPU
oracle on the same seeded problem and ordering (the oracle needs 0.5-2 s per iteration at these sizes),
plus size-independent properties of the path:
  * the damped Newton step satisfies the normal equations it was computed from
    (q(0) - q(delta) = 1/2 delta'H delta + lambda delta'D delta  with  (H + lambda D) delta = g),
  * the factorization is bitwise reproducible run to run (no atomics on the path),
  * a second solve at the same point and lambda gives the identical step (idempotence),
  * committing steps only ever lowers the error the LM policy accepts.
"""
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


CASES = {
    # name: (builder, ordering kind)
    "bal1723": (lambda: datasets.synth_bal_arrays(1723, 156502, 678718, seed=42, long_range=0.3), A.ORDER_SCHUR),
    "bal49": (lambda: datasets.synth_bal_arrays(49, 7776, 31843, seed=42, long_range=0.3), A.ORDER_SCHUR),
    "pose2_10k": (lambda: datasets.synth_manhattan_pose2(10000, seed=7), A.ORDER_MINDEGREE),
    "pose3_100k": (lambda: datasets.synth_manhattan_pose3(100000, seed=7), A.ORDER_ND),
}


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("name", list(CASES))
def test_full_size_step_matches_oracle(gpu, oracle, name):
    build, kind = CASES[name]
    arr = build()
    gb = gpu.product_backend(arr)
    gb.set_amalgamation(0.0, 128)   # the reference's cliques: the tree statistics are compared below
    ob = oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(kind)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    eg, eo = gb.error(), ob.error()
    assert abs(eg - eo) <= 1e-10 * abs(eo)
    gb.linearize()
    ob.linearize()
    lam = 1e-3
    dg = gb.solve(lam, False)
    do = ob.solve(lam, False)
    assert relerr(dg, do) < 1e-6, name           # north star: updates within 1e-6 relative
    e0g, edg = gb.linear_error()
    e0o, edo = ob.linear_error()
    assert abs(e0g - e0o) <= 1e-10 * abs(e0o)
    assert abs(edg - edo) <= 1e-6 * max(abs(edo), 1e-9 * abs(e0o))
    tg = gb.retract(None, commit=False)
    to = ob.retract(None, commit=False)
    assert abs(tg - to) <= 1e-6 * abs(to)
    # the Bayes tree is the reference construction at full size too
    st = gb.stats()
    _, tree = ob.timing()
    assert st["n_fronts"] == tree["cliques"] and st["max_front_dim"] == tree["max_f"]
    assert abs(st["factor_flops"] - tree["flops"]) <= 1e-9 * tree["flops"]


BENCH_WORKLOADS = ["bal1723", "pose3_100k", "pose2_100k"]


@pytest.mark.parametrize("name", BENCH_WORKLOADS)
def test_bench_configuration_matches_oracle(gpu, oracle, name):
    """The exact configuration bench.py times — the problem, seed and ordering kind come from bench.make_problem itself,
    the clique amalgamation is the library's own choice (a new handle's default, as in bench.py) — against the oracle,
    which eliminates the reference's (un-amalgamated) Bayes tree for the same ordering."""
    import bench
    arr, okind = bench.make_problem(name, 42)
    kind = {"schur_nd": A.ORDER_SCHUR_ND, "nd": A.ORDER_ND}[okind]
    gb = gpu.product_backend(arr)
    ob = oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(kind)
    gb.set_ordering(ordering)
    st = gb.stats()
    assert st["amalgamation_relax"] > 0 and st["n_fronts"] > 0   # the library merged cliques on its own
    ob.set_ordering(ordering)
    gb.linearize()
    ob.linearize()
    for lam, diag in ((1e-5, False), (1e-2, True)):
        dg = gb.solve(lam, diag)
        do = ob.solve(lam, diag)
        assert relerr(dg, do) < 1e-6, (name, lam)    # north star: updates within 1e-6 relative
        e0g, edg = gb.linear_error()
        e0o, edo = ob.linear_error()
        assert abs(e0g - e0o) <= 1e-10 * abs(e0o)
        assert abs(edg - edo) <= 1e-6 * max(abs(edo), 1e-9 * abs(e0o))
    tg = gb.retract(None, commit=False)
    to = ob.retract(None, commit=False)
    assert abs(tg - to) <= 1e-6 * abs(to)
    _, tree = ob.timing()
    assert gb.stats()["n_fronts"] < tree["cliques"]  # amalgamation did happen
    d2 = gb.solve(1e-2, True)
    assert np.array_equal(dg, d2)                     # and the path stays bitwise reproducible


@pytest.mark.parametrize("name", ["bal1723", "pose3_100k"])
def test_full_size_properties(gpu, name):
    build, kind = CASES[name]
    arr = build()
    gb = gpu.product_backend(arr)
    gb.set_ordering(gb.compute_ordering(kind))
    gb.linearize()
    hdiag = gb.hessian_diagonal()
    assert np.all(hdiag > 0)
    lam = 1e-2
    d1 = gb.solve(lam, False)
    e0, ed = gb.linear_error()
    d2 = gb.solve(lam, False)
    assert np.array_equal(d1, d2), "the factorization must be bitwise reproducible (no atomics on the path)"
    # normal equations: with (H + lam I) d = g the model decrease is 1/2 d'Hd + lam d'd >= lam d'd, and
    # g'd = d'(H + lam I) d > 0; both sides from independent kernels (linear_error vs the solve)
    dd = float(d1 @ d1)
    assert e0 - ed >= lam * dd * (1 - 1e-9)
    # diagonal damping with huge lambda: delta -> g / (lam diag H) componentwise, independent of the tree
    big = 1e12
    d3 = gb.solve(big, True, min_diagonal=0.0, max_diagonal=1e300)
    # g = H d + lam D d  ~ lam D d  =>  compare two huge lambdas: d scales as 1/lambda
    d4 = gb.solve(10 * big, True, min_diagonal=0.0, max_diagonal=1e300)
    assert relerr(10 * d4, d3) < 1e-9
    # LM on the device only ever accepts decreasing errors
    p = A.lm_params_legacy()
    p.max_iterations = 3
    r = gb.lm_optimize(p)
    acc = r["trace_error"][r["trace_accepted"] == 1]
    assert acc.size >= 1 and np.all(np.diff(np.concatenate([[r["initial_error"]], acc])) < 0)


@pytest.mark.parametrize("name", ["bal1723", "pose3_100k"])
def test_full_size_lm_run_matches_oracle(gpu, oracle, name):
    """north star: the LM run itself — same accept/reject decisions, same lambda schedule, final chi^2 within 1e-6 of the
    reference algorithm — at BASELINE's full sizes, in the configuration bench.py times (bench.make_problem, the library's
    own amalgamation on the device, the reference's cliques in the oracle)."""
    import bench
    arr, okind = bench.make_problem(name, 42)
    kind = {"schur_nd": A.ORDER_SCHUR_ND, "nd": A.ORDER_ND}[okind]
    gb = gpu.product_backend(arr)
    ob = oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(kind)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    p = A.lm_params_legacy()
    p.max_iterations = 6
    rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
    assert rg["iterations"] == ro["iterations"] and rg["inner_iterations"] == ro["inner_iterations"]
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert np.allclose(rg["trace_lambda"], ro["trace_lambda"], rtol=1e-9)
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"]
    assert rg["final_error"] < rg["initial_error"]
    assert relerr(gb.get_values(), ob.get_values()) < 1e-6


# ---- BASELINE config 3 as written: "LM + METIS ordering" ---------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bal1723", "pose3_100k", "pose2_100k"])
def test_reference_metis_ordering_through_the_hip_path(gpu, oracle, golden_dir, name):
    """The elimination order the REFERENCE's METIS_NodeND returns for the whole graph (Ordering::Metis as
    SFMExample_bal_COLAMD_METIS.cpp:83-117 / LevenbergMarquardtParams orderingType = METIS use it; the committed fixture
    tests/golden/metis_perm_*_seed42.npy, made by tests/golden/make_metis_orderings.py with the reference's own METIS)
    handed over through gsx_set_ordering: damped step within 1e-6 of the oracle, which eliminates the reference's Bayes
    tree for the same ordering, and the LM run with identical accept/reject decisions and final chi^2 within 1e-6."""
    import bench
    arr, _ = bench.make_problem(name, 42)
    ordering = bench.load_ordering(arr, name)
    assert np.array_equal(np.sort(ordering), arr.var_keys)
    gb = gpu.product_backend(arr)
    ob = oracle.oracle_backend(arr)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    assert np.array_equal(gb.get_ordering(), ordering)     # the order is taken as given (bit-identical by construction)
    gb.linearize()
    ob.linearize()
    for lam, diag in ((1e-5, False), (1e-2, True)):
        dg = gb.solve(lam, diag)
        do = ob.solve(lam, diag)
        assert relerr(dg, do) < 1e-6, (name, lam)
        e0g, edg = gb.linear_error()
        e0o, edo = ob.linear_error()
        assert abs(e0g - e0o) <= 1e-10 * abs(e0o)
        assert abs(edg - edo) <= 1e-6 * max(abs(edo), 1e-9 * abs(e0o))
    p = A.lm_params_legacy()
    p.max_iterations = 4
    rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
    assert rg["iterations"] == ro["iterations"] and rg["inner_iterations"] == ro["inner_iterations"]
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert np.allclose(rg["trace_lambda"], ro["trace_lambda"], rtol=1e-9)
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"]
    assert relerr(gb.get_values(), ob.get_values()) < 1e-6


@pytest.mark.parametrize("K,M,OBS", [(2000, 200000, 900000), (10000, 1000000, 4500000)])
def test_keyframe_update_at_config_5_size_matches_a_fresh_handle(gpu, K, M, OBS):
    """BASELINE config 5 (VisualISAM2Example-style growth) at its stated size, 10 000 keyframes / 1 000 000 landmarks /
    4 500 000 projections (and a fifth of it): a handle optimised up to keyframe K - 2 takes keyframe K - 1 through
    gsx_update (its pose, the landmarks that become observable, their factors) and must then hold exactly what a fresh handle
    on the grown graph holds at the same values and ordering — graph error, the kept and the new [A b], the Gauss-Newton
    step — and one relinearize-partial round after it is bit for bit the handle's own full path.  (No oracle at this size: the two
    sides are two independent routes through the product, and the route through a fresh handle is the one every other test
    pins against the oracle.)"""
    a1, id1 = datasets.synth_visual_slam(K, M, OBS, upto=K - 1)
    a2, id2 = datasets.synth_visual_slam(K, M, OBS)
    pos = {int(i): k for k, i in enumerate(id1)}
    origin = np.array([pos.get(int(i), -1) for i in id2], dtype=np.int32)
    so, old = a2.state_offsets(), set(int(k) for k in a1.var_keys)
    new_idx = [i for i, k in enumerate(a2.var_keys) if int(k) not in old]
    new_states = np.concatenate([a2.values[so[i]:so[i + 1]] for i in new_idx])
    pb = gpu.product_backend(a1)
    pb.set_ordering(pb.compute_ordering(A.ORDER_SCHUR_ND))
    p = A.lm_params_legacy()
    p.max_iterations = 2
    pb.lm_optimize(p)
    pb.linearize()
    pb.solve(0.0, False, want_delta=False)
    st = pb.update(a2, origin, new_states)
    assert st["n_vars_added"] == len(new_idx) and st["n_factors_added"] == int(np.sum(origin < 0))
    assert st["n_factors_removed"] == 0
    fresh = datasets.synth_visual_slam(K, M, OBS)[0]
    fresh.values = pb.get_values()
    fb = gpu.product_backend(fresh)
    fb.set_ordering(pb.get_ordering())
    assert abs(pb.error() - fb.error()) <= 1e-12 * abs(fb.error())
    fb.linearize()
    jp, jf = pb.jacobians(), fb.jacobians()     # kept factors keep their blocks (copied), the new ones are linearized
    assert np.max(np.abs(jp - jf)) <= 1e-12 * np.max(np.abs(jf))
    del jp, jf
    # the undamped system of a graph whose newest landmarks have two or three views is nearly singular (steps of 1e2): there
    # the two handles are compared through what the step achieves — the linearized error at the step — and the steps
    # themselves on the damped systems of an LM iteration
    dp, df = pb.solve(0.0, False), fb.solve(0.0, False)
    ep, ef = pb.linear_error(), fb.linear_error()
    assert np.allclose(ep, ef, rtol=1e-9), (ep, ef)
    for lam in (1e-3, 1.0):
        dp, df = pb.solve(lam, False), fb.solve(lam, False)
        assert relerr(dp, df) < 1e-8, (lam, relerr(dp, df))
    pb.solve(0.0, False, want_delta=False)      # (gsx_relinearize_partial works on the undamped factorization)
    # an iSAM2-style relinearization of the ten most recent keyframes on the updated handle: bit for bit the full path
    so2 = a2.state_offsets()
    idx = np.nonzero(a2.var_types == A.VAR_POSE3)[0][-10:]
    cur = pb.get_values()
    rng = np.random.default_rng(5)
    states = np.concatenate([cur[so2[i]:so2[i + 1]] for i in idx])
    for k in range(len(idx)):
        states[12 * k + 9:12 * k + 12] += 1e-3 * rng.standard_normal(3)     # translations only
        cur[so2[idx[k]] + 9:so2[idx[k]] + 12] = states[12 * k + 9:12 * k + 12]
    stp = pb.relinearize_partial(a2.var_keys[idx], states)
    assert 0 < stp["n_fronts_reeliminated"] < 0.05 * stp["n_fronts"]
    dp = pb.solve(0.0, False)
    assert np.array_equal(pb.get_values(), cur)
    pb.linearize()                               # the full path on the same handle at the same values
    assert np.array_equal(dp, pb.solve(0.0, False))
    fb.set_values(cur)
    fb.linearize()
    assert relerr(pb.solve(1e-3, False), fb.solve(1e-3, False)) < 1e-8
