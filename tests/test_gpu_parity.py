"""GPU parity tests proper: the HIP path (through the C-ABI of include/gsx.h) against the CPU
oracle on the same seeded inputs, and against the reference's golden values.

Tolerances: the path is FP64; Jacobians/residuals are compared at 1e-11 relative, solve results
(delta, errors) at 1e-8 relative to the vector norm (the north star asks for 1e-6), integer
structure (Bayes tree, orderings) exactly.
"""
import math

import numpy as np
import pytest

import gtsam_petercdev_amd as gt
from gtsam_petercdev_amd import _abi as A
from gtsam_petercdev_amd import datasets
from gtsam_petercdev_amd.graph import (X, L, Pose2, Pose3, Rot3, Point2, Point3, Values, NonlinearFactorGraph,
                                       GaussianFactorGraph, JacobianFactor, PriorFactor, BetweenFactor,
                                       GeneralSFMFactor, Cal3Bundler, PinholeCameraCal3Bundler, noiseModel,
                                       Ordering, LevenbergMarquardtOptimizer, LevenbergMarquardtParams,
                                       GaussNewtonOptimizer)

pytestmark = pytest.mark.gpu
# GSX_FUZZ_OFFSET=<n> moves every structure fuzz below to other seeds (an occasional wider sweep; the default is what CI runs).
# End of round 3: twelve offsets x 155 cases, one miss — test_random_bal_structures[2] at offset 77000, a bundle whose
# rejected trials reach errors of 1e11: after three LM iterations the final error differs by 9e-6 from the oracle's, while
# the oracle's own two orderings differ by 1e-6 there (conditioning, not a kernel).
FUZZ_OFFSET = int(__import__("os").environ.get("GSX_FUZZ_OFFSET", "0"))


@pytest.fixture(scope="module")
def gpu():
    from gtsam_petercdev_amd import _lib
    assert _lib.device_count() > 0, "no GPU visible: the HIP path has no fallback"
    return _lib


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# ---- dense kernel: choleskyPartial (gtsam/base/tests/testCholesky.cpp) ------------------------------
def _gpu_cholesky(gpu, abc, nf):
    import ctypes as C
    m = np.asfortranarray(abc, dtype=np.float64).copy(order="F")
    ok = C.c_int32()
    st = gpu.load().gsx_cholesky_partial(m.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(m.shape[0]),
                                         C.c_int32(nf), C.c_int32(0), C.byref(ok))
    assert st == 0
    return m, bool(ok.value)


def test_choleskyPartial_golden(gpu, oracle):
    from tests.test_oracle_golden import ABC7
    got, ok = _gpu_cholesky(gpu, ABC7, 3)
    exp, ok2 = oracle.cholesky_partial(ABC7, 3)
    assert ok and ok2
    assert np.allclose(np.triu(got), np.triu(exp), atol=1e-12)
    got0, ok = _gpu_cholesky(gpu, ABC7[:3, :3], 0)
    assert ok and np.allclose(got0, ABC7[:3, :3])


@pytest.mark.parametrize("n,nf", [(5, 5), (43, 3), (64, 12), (97, 33), (140, 64), (141, 31), (200, 200), (333, 150),
                                   (515, 257)])
def test_choleskyPartial_random(gpu, oracle, n, nf):
    rng = np.random.default_rng(n * 1000 + nf)
    Bm = rng.normal(size=(n + 5, n))
    S = Bm.T @ Bm + n * np.eye(n)
    got, ok = _gpu_cholesky(gpu, S, nf)
    exp, ok2 = oracle.cholesky_partial(S, nf)
    assert ok and ok2
    assert relerr(np.triu(got), np.triu(exp)) < 1e-12


def test_cholesky_underconstrained(gpu):
    Lm = np.array([
        [1, 0, 0, 0, 0, 0],
        [1.11177808157954, 1.06204809504665, 0.507342638873381, 1.34953401829486, 1, 0],
        [0.155864888199928, 1.10933048588373, 0.501255576961674, 1, 0, 0],
        [1.12108665967793, 1.01584408366945, 1, 0, 0, 0],
        [0.776164062474843, 0.117617236580373, -0.0236628691347294, 0.814118199972143, 0.694309975328922, 1],
        [0.1197220685104, 1, 0, 0, 0, 0]])
    d = [0.814723686393179, 0.811780089277421, 1.82596950680844, 0.240287537694585]
    for tail in ([1.34342584865901, 1e-12], [0, 0], [-0.5, -0.6]):
        _, ok = _gpu_cholesky(gpu, Lm @ np.diag(d + tail) @ Lm.T, 6)
        assert not ok


# ---- per-step parity on seeded problems ---------------------------------------------------------------------
def _mixed_noise(arr, rng):
    """Give the factors of a synthetic problem a mix of Unit / Isotropic / Diagonal / Gaussian noise."""
    kinds, ptr, vals = [], [0], []
    for f in range(arr.n_factors):
        m = int(arr.f_rows[f])
        k = int(rng.integers(0, 4))
        if k == A.NOISE_UNIT:
            p = np.zeros(0)
        elif k == A.NOISE_ISOTROPIC:
            p = rng.uniform(0.5, 2.0, 1)
        elif k == A.NOISE_DIAGONAL:
            p = rng.uniform(0.5, 2.0, m)
        else:
            p = np.triu(rng.normal(size=(m, m)) * 0.2 + np.eye(m) * rng.uniform(0.8, 1.5)).reshape(-1)
        kinds.append(k)
        vals.append(p)
        ptr.append(ptr[-1] + p.size)
    return A.ProblemArrays(arr.var_keys, arr.var_types, arr.var_dims, arr.f_type, arr.f_rows, arr.f_key_ptr,
                           arr.f_vars, arr.f_meas_ptr, arr.meas, kinds, ptr, np.concatenate(vals), arr.values,
                           dict(arr.meta))


def _problems():
    rng = np.random.default_rng(5)
    bal = datasets.synth_bal_arrays(6, 40, 150, seed=11, priors=True)
    yield "bal_small", bal
    yield "bal_mixed_noise", _mixed_noise(bal, rng)
    yield "bal_bigfront", datasets.synth_bal_arrays(24, 300, 2400, seed=12, long_range=0.5)
    # landmarks seen by ~30 cameras each: leaf cliques of ~270 rows whose Schur complement is never stored
    yield "bal_wide_landmarks", datasets.synth_bal_arrays(40, 60, 1800, seed=13, long_range=0.3)
    p2 = datasets.synth_manhattan_pose2(400, seed=3)
    yield "pose2", p2
    yield "pose2_mixed_noise", _mixed_noise(p2, rng)
    p3 = datasets.synth_manhattan_pose3(300, seed=4)
    yield "pose3", p3
    yield "pose3_mixed_noise", _mixed_noise(p3, rng)


PROBLEMS = dict(_problems())


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_step_parity(gpu, oracle, name):
    arr = PROBLEMS[name]
    gb = gpu.product_backend(arr)
    gb.set_amalgamation(0.0, 128)   # the reference's cliques: the tree itself is compared below
    ob = oracle.oracle_backend(arr)
    # nonlinear error
    assert abs(gb.error() - ob.error()) <= 1e-11 * abs(ob.error())
    # Jacobians
    gb.linearize()
    ob.linearize()
    jg, jo = gb.jacobians(), ob.jacobians()
    assert np.max(np.abs(jg - jo)) <= 1e-11 * max(1.0, np.max(np.abs(jo)))
    kinds = (A.ORDER_MINDEGREE, A.ORDER_ND) + ((A.ORDER_SCHUR_ND, A.ORDER_SCHUR) if name.startswith("bal") else ())
    for kind in kinds:
        ordering = gb.compute_ordering(kind)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.linearize()
        hd_g, hd_o = gb.hessian_diagonal(), ob.hessian_diagonal()
        assert relerr(hd_g, hd_o) < 1e-12
        for lam, diag in ((1e-3, False), (1.0, True), (1e-5, False)):
            dg = gb.solve(lam, diag)
            do = ob.solve(lam, diag)
            assert relerr(dg, do) < 1e-8, (name, kind, lam, diag)
            e0g, edg = gb.linear_error()
            e0o, edo = ob.linear_error()
            assert abs(e0g - e0o) <= 1e-10 * abs(e0o) and abs(edg - edo) <= 1e-8 * max(abs(edo), 1e-12 * abs(e0o))
            tg = gb.retract(None, commit=False)
            to = ob.retract(None, commit=False)
            assert abs(tg - to) <= 1e-8 * max(abs(to), 1e-9)
        # the Bayes tree is the reference's (integer structure: exact)
        pg, fg = gb.get_tree()
        po, fo = ob.get_tree()
        tg = {tuple(sorted(f)): tuple(sorted(s)) for f, s in fg}
        to = {tuple(sorted(f)): tuple(sorted(s)) for f, s in fo}
        assert tg == to
    # committing a retraction moves the values identically
    d = ob.solve(1e-3, False)
    gb.solve(1e-3, False)
    gb.retract(None, commit=True, want_error=False)
    ob.retract(None, commit=True, want_error=False)
    assert relerr(gb.get_values(), ob.get_values()) < 1e-10


@pytest.mark.parametrize("name", list(PROBLEMS))
@pytest.mark.parametrize("relax,maxf", [(0.25, 128), (1.0, 64), (50.0, 4096)])
def test_step_parity_with_relaxed_amalgamation(gpu, oracle, name, relax, maxf):
    """gsx_set_amalgamation: the product eliminates merged cliques (explicit zeros), the oracle the reference's tree;
    the solution of the damped system must not move."""
    arr = PROBLEMS[name]
    gb = gpu.product_backend(arr)
    ob = oracle.oracle_backend(arr)
    gb.linearize()
    ob.linearize()
    kinds = (A.ORDER_MINDEGREE, A.ORDER_ND) + ((A.ORDER_SCHUR_ND,) if name.startswith("bal") else ())
    for kind in kinds:
        ordering = gb.compute_ordering(kind)
        gb.set_amalgamation(relax, maxf)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.linearize()
        for lam, diag in ((1e-3, False), (1.0, True)):
            dg = gb.solve(lam, diag)
            do = ob.solve(lam, diag)
            assert relerr(dg, do) < 1e-8, (name, kind, lam, diag)
            e0g, edg = gb.linear_error()
            e0o, edo = ob.linear_error()
            assert abs(e0g - e0o) <= 1e-10 * abs(e0o) and abs(edg - edo) <= 1e-8 * max(abs(edo), 1e-12 * abs(e0o))
        _, tree = ob.timing()
        assert gb.stats()["n_fronts"] <= tree["cliques"]


@pytest.mark.parametrize("name", ["bal_small", "bal_bigfront", "pose2", "pose3"])
@pytest.mark.parametrize("preset", ["legacy", "ceres"])
@pytest.mark.parametrize("relax", [0.0, 0.5])
def test_lm_trajectory_parity(gpu, oracle, name, preset, relax):
    """Same accept/reject decisions and the same (error, lambda) trace as the oracle; final chi^2 within 1e-6 — with
    the reference's cliques and with relaxed amalgamation (the oracle always eliminates the reference tree)."""
    arr = PROBLEMS[name]
    p = A.lm_params_legacy() if preset == "legacy" else A.lm_params_ceres()
    p.max_iterations = 12
    gb = gpu.product_backend(arr)
    ob = oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(A.ORDER_MINDEGREE)
    gb.set_amalgamation(relax, 128)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
    assert rg["iterations"] == ro["iterations"] and rg["inner_iterations"] == ro["inner_iterations"]
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert np.allclose(rg["trace_lambda"], ro["trace_lambda"], rtol=1e-9)
    fin = np.isfinite(ro["trace_error"])
    assert np.allclose(rg["trace_error"][fin], ro["trace_error"][fin], rtol=1e-6)
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"]
    assert rg["final_error"] < 0.5 * rg["initial_error"]


# ---- the reference's own known-answer tests through the HIP path --------------------------------------------------
@pytest.mark.parametrize("reader", ["python", "native"])
def test_PinholeCamera_BAL(gpu, golden_dir, reader):
    """tests/testGeneralSFMFactorB.cpp:44-63: dubrovnik-3-7-pre, default LM -> graph.error = 0.0199833 +- 1e-5 (file
    read by the Python reader and by the native gsx_read_bal)."""
    path = golden_dir + "/dubrovnik-3-7-pre.txt"
    arrays = datasets.bal_arrays(datasets.read_bal(path), priors=False) if reader == "python" else gpu.read_bal(path)
    be = gpu.product_backend(arrays)
    be.set_ordering(np.load(golden_dir + "/dubrovnik_colamd_ordering.npy"))  # the reference's CCOLAMD result
    res = be.lm_optimize(A.lm_params_legacy())
    assert abs(res["final_error"] - 0.0199833) < 1e-5
    assert abs(be.error() - 0.0199833) < 1e-5


def test_optimizeMultiFrontal2(gpu):
    """tests/testGaussianJunctionTreeB.cpp:127-140 through gsx_solve_gfg-style linear solve."""
    from tests.test_oracle_golden import small_gaussian_factor_graph, CORRECT_DELTA
    for ordering in ([L(1), X(1), X(2)], [X(2), L(1), X(1)], None):
        actual = small_gaussian_factor_graph().optimize(ordering)
        for k, v in CORRECT_DELTA.items():
            assert np.allclose(actual[k], v, atol=1e-9)


def test_GaussianBayesTree_chain(gpu):
    """gtsam/linear/tests/testGaussianBayesTree.cpp:84-129 on the device: tree (x3 x4) <- (x2 x1 : x3), x = (0,1,0,1)."""
    from tests.test_oracle_golden import chain_graph
    actual = chain_graph().optimize([2, 1, 3, 4])
    for k, v in {1: 0.0, 2: 1.0, 3: 0.0, 4: 1.0}.items():
        assert np.allclose(actual[k], [v], atol=1e-9)
    arrays = chain_graph().to_arrays(None)
    arrays.values = np.zeros(4)
    be = gpu.product_backend(arrays)
    be.set_amalgamation(0.0, 128)   # the reference's cliques
    be.set_ordering([2, 1, 3, 4])
    parent, fronts = be.get_tree()
    keys = arrays.var_keys.tolist()
    cl = {tuple(keys[i] for i in f): c for c, (f, s) in enumerate(fronts)}
    assert set(cl) == {(3, 4), (2, 1)} and parent[cl[(2, 1)]] == cl[(3, 4)]


def test_HessianFactor_hessianDiagonal(gpu):
    """gtsam/linear/tests/testHessianFactor.cpp:447-477: diagonal of the expected 7x7 information matrix."""
    from gtsam_petercdev_amd.graph import GaussianFactorGraph, JacobianFactor
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(0, 11.1803399 * np.eye(2), 1, -2.23606798 * np.eye(2), 2, -8.94427191 * np.eye(2),
                          [2.23606798, -1.56524758], noiseModel.Diagonal.Sigmas([1.0, 1.0])))
    arrays = fg.to_arrays(None)
    arrays.values = np.zeros(6)
    be = gpu.product_backend(arrays)
    be.set_ordering([0, 1, 2])
    be.linearize()
    assert np.allclose(be.hessian_diagonal(), [125.0, 125.0, 5.0, 5.0, 80.0, 80.0], atol=1e-4)


def test_solve_gfg_entry_point(gpu):
    import ctypes as C
    from tests.test_oracle_golden import small_gaussian_factor_graph, CORRECT_DELTA
    arrays = small_gaussian_factor_graph().to_arrays(None)
    desc = arrays.desc()
    out = np.zeros(6)
    bad = C.c_uint64()
    st = gpu.load().gsx_solve_gfg(C.byref(desc), None, C.c_int32(0), out.ctypes.data_as(C.POINTER(C.c_double)),
                                  C.c_int64(6), C.byref(bad))
    assert st == 0
    exp = np.concatenate([CORRECT_DELTA[int(k)] for k in arrays.var_keys])
    assert np.allclose(out, exp, atol=1e-9)


def test_smoother_zero_delta(gpu):
    from tests.test_oracle_golden import nonlinear_smoother
    g, v = nonlinear_smoother(7)
    be = gpu.product_backend(g.to_arrays(v))
    be.set_ordering([X(1), X(3), X(5), X(7), X(2), X(6), X(4)])
    be.linearize()
    assert np.allclose(be.solve(0.0), 0.0, atol=1e-9)


def test_Factorization(gpu):
    """tests/testNonlinearOptimizer.cpp:185-208."""
    config = Values()
    config.insert(X(1), Pose2(0., 0., 0.))
    config.insert(X(2), Pose2(1.5, 0., 0.))
    graph = NonlinearFactorGraph()
    graph.addPrior(X(1), Pose2(0., 0., 0.), noiseModel.Isotropic.Sigma(3, 1e-10))
    graph.add(BetweenFactor(X(1), X(2), Pose2(1., 0., 0.), noiseModel.Isotropic.Sigma(3, 1)))
    opt = LevenbergMarquardtOptimizer(graph, config, Ordering([X(1), X(2)]), LevenbergMarquardtParams.LegacyDefaults())
    opt.iterate()
    res = opt.values()
    assert res.at(X(1)).equals(Pose2(0., 0., 0.), 1e-5) and res.at(X(2)).equals(Pose2(1., 0., 0.), 1e-5)


def test_MoreOptimization(gpu):
    """tests/testNonlinearOptimizer.cpp:248-281 — with the library's own ordering."""
    from tests.test_oracle_golden import more_optimization_graph
    fg = more_optimization_graph()
    init = Values()
    init.insert(0, Pose2(3, 4, -math.pi))
    init.insert(1, Pose2(10, 2, -math.pi))
    init.insert(2, Pose2(11, 7, -math.pi))
    expected = {0: Pose2(0, 0, 0), 1: Pose2(1, 0, math.pi / 2), 2: Pose2(1, 1, math.pi)}
    actual = LevenbergMarquardtOptimizer(fg, init, LevenbergMarquardtParams.LegacyDefaults()).optimize()
    for k, e in expected.items():
        assert actual.at(k).equals(e, 1e-6)
    actual = GaussNewtonOptimizer(fg, actual).optimize()
    for k, e in expected.items():
        assert actual.at(k).equals(e, 1e-6)


def test_disconnected_graph(gpu):
    """tests/testNonlinearOptimizer.cpp:485-503 on the device: a forest with two roots."""
    from tests.test_oracle_golden import disconnected_graph
    graph, init, expected = disconnected_graph()
    actual = LevenbergMarquardtOptimizer(graph, init, Ordering([X(1), X(2), X(3)]), LevenbergMarquardtParams()).optimize()
    for k, e in expected.items():
        assert actual.at(k).equals(e, 1e-9)
    # ... and with the library's own ordering (no ordering given)
    actual = LevenbergMarquardtOptimizer(graph, init).optimize()
    for k, e in expected.items():
        assert actual.at(k).equals(e, 1e-9)


def test_single_variable_and_tiny_graphs(gpu, oracle):
    """Smallest inputs: one variable with one prior (a 1-clique tree with an empty separator), and a 2-variable chain."""
    g = NonlinearFactorGraph()
    g.addPrior(7, Pose2(1., 2., 0.3), noiseModel.Diagonal.Sigmas([0.1, 0.2, 0.3]))
    v = Values()
    v.insert(7, Pose2(0., 0., 0.))
    arr = g.to_arrays(v)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    for be in (gb, ob):
        be.set_ordering([7])
        be.linearize()
    assert relerr(gb.solve(0.0), ob.solve(0.0)) < 1e-12
    out = LevenbergMarquardtOptimizer(g, v).optimize()
    assert out.at(7).equals(Pose2(1., 2., 0.3), 1e-9)


def test_indeterminate_system(gpu):
    fg = NonlinearFactorGraph()
    fg.add(BetweenFactor(0, 1, Pose2(1, 0, 0), noiseModel.Isotropic.Sigma(3, 1)))
    v = Values()
    v.insert(0, Pose2(0, 0, 0))
    v.insert(1, Pose2(1, 0, 0))
    be = gpu.product_backend(fg.to_arrays(v))
    be.set_ordering([0, 1])
    be.linearize()
    with pytest.raises(gt.IndeterminantLinearSystemException) as ei:
        be.solve(0.0)
    assert ei.value.key in (0, 1)
    # ... and LM recovers by raising lambda (LevenbergMarquardtOptimizer.cpp:158-160)
    res = be.lm_optimize(A.lm_params_legacy())
    assert res["n_solve_failures"] >= 0 and np.isfinite(res["final_error"])


def test_underconstrained_clique_inside_a_relaxed_front(gpu, oracle):
    """choleskyPartial's conditioning test (cholesky.cpp:145-158) belongs to every clique of the REFERENCE tree.  Variable
    p is a clique of its own there (separator {q}); its two pivots are 16 binary orders apart — the reference throws.  With
    relaxed amalgamation p's clique is folded into q's front, where its pivots are no longer "the last two": the verdict
    must not change (it is taken per reference clique after the factorization, not per front inside the kernels)."""
    iso = noiseModel.Isotropic.Sigma(2, 1.0)
    fg, v = NonlinearFactorGraph(), Values()
    fg.add(PriorFactor(0, Point2(0, 0), noiseModel.Diagonal.Sigmas([1e-5, 1.0])))     # p: L_00 ~ 1e5, L_11 ~ 1
    fg.add(BetweenFactor(0, 1, Point2(1, 0), iso))                                    # p - q
    fg.add(BetweenFactor(1, 2, Point2(1, 0), iso))                                    # q - r
    fg.add(BetweenFactor(1, 3, Point2(0, 1), iso))                                    # q - s
    fg.add(BetweenFactor(2, 3, Point2(-1, 1), iso))                                   # r - s
    fg.add(PriorFactor(2, Point2(2, 0), iso))
    for k, xy in enumerate([(0, 0), (1, 0), (2, 0), (1, 1)]):
        v.insert(k, Point2(*xy))
    arr = fg.to_arrays(v)
    ob = oracle.oracle_backend(arr)
    ob.set_ordering([0, 1, 2, 3])
    ob.linearize()
    with pytest.raises(gt.IndeterminantLinearSystemException):
        ob.solve(0.0)
    for relax in (0.0, 50.0, A.AMALGAMATION_AUTO):
        be = gpu.product_backend(arr)
        be.set_amalgamation(relax, 128)
        be.set_ordering([0, 1, 2, 3])
        if relax == 0.0:
            assert be.stats()["n_fronts"] >= 2       # p's clique is separate in the reference tree ...
        if relax == 50.0:
            assert be.stats()["n_fronts"] == 1       # ... and swallowed here
        be.linearize()
        with pytest.raises(gt.IndeterminantLinearSystemException):
            be.solve(0.0)
        be.close()
    # the same graph with a sane prior on p solves in every mode, identically
    fg.factors[0] = PriorFactor(0, Point2(0, 0), iso)
    arr = fg.to_arrays(v)
    ob = oracle.oracle_backend(arr)
    ob.set_ordering([0, 1, 2, 3])
    ob.linearize()
    want = ob.solve(0.0)
    for relax in (0.0, 50.0):
        be = gpu.product_backend(arr)
        be.set_amalgamation(relax, 128)
        be.set_ordering([0, 1, 2, 3])
        be.linearize()
        assert relerr(be.solve(0.0), want) < 1e-9


def test_cheirality_zeroes_factor(gpu, oracle):
    g, v = NonlinearFactorGraph(), Values()
    g.add(GeneralSFMFactor(Point2(3., 0.), noiseModel.Unit.Create(2), X(1), L(1)))
    g.add(GeneralSFMFactor(Point2(1., 1.), noiseModel.Unit.Create(2), X(1), L(2)))
    v.insert(X(1), PinholeCameraCal3Bundler(Pose3(Rot3(), Point3(0, 0, -6)), Cal3Bundler(1.0, 0.0, 0.0)))
    v.insert(L(1), Point3(0, 0, 0))
    v.insert(L(2), Point3(0, 0, -10))  # behind the camera
    arr = g.to_arrays(v)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    gb.linearize()
    ob.linearize()
    assert np.allclose(gb.jacobians(), ob.jacobians(), atol=1e-13)
    assert np.all(gb.jacobians()[26:] == 0.0)
    assert gb.stats()["n_cheirality"] == 1
    assert abs(gb.error() - 4.5) < 1e-12


def test_bad_ordering_is_rejected(gpu):
    arr = PROBLEMS["pose2"]
    be = gpu.product_backend(arr)
    with pytest.raises(gt.GsxError) as ei:
        be.set_ordering(arr.var_keys[:-1])
    assert ei.value.status == A.GSX_E_BAD_ORDERING
    bad = arr.var_keys.copy()
    bad[0] = bad[1]
    with pytest.raises(gt.GsxError):
        be.set_ordering(bad)


# ---- Dogleg (gsx_dogleg_optimize) -----------------------------------------------------------------------------
def test_dogleg_point_matches_the_oracle(gpu, oracle):
    """gsx_dogleg_point (host) against the oracle's restatement of ComputeDoglegPoint in its three regimes + edges."""
    rng = np.random.default_rng(3)
    for n in (3, 10, 1000):
        xn = rng.normal(size=n)
        xu = 0.3 * xn + 0.05 * rng.normal(size=n)
        for delta in (0.1 * np.linalg.norm(xu), 0.5 * (np.linalg.norm(xu) + np.linalg.norm(xn)), 2 * np.linalg.norm(xn),
                      np.linalg.norm(xu), np.linalg.norm(xn)):
            assert np.allclose(gpu.dogleg_point(delta, xu, xn), oracle.dogleg_point(delta, xu, xn), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("name", ["bal_small", "bal_bigfront", "pose2", "pose3"])
@pytest.mark.parametrize("delta0", [1.0, 0.05])
def test_dogleg_trajectory_parity(gpu, oracle, name, delta0):
    """DoglegOptimizer on the device against the oracle: same trust-region schedule, errors within 1e-6."""
    arr = PROBLEMS[name]
    gb = gpu.product_backend(arr)
    ob = oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(A.ORDER_MINDEGREE)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    rg, ro = gb.dogleg_optimize(delta0, 10), ob.dogleg_optimize(delta0, 10)
    assert rg["iterations"] == ro["iterations"]
    assert np.allclose(rg["trace_lambda"], ro["trace_lambda"], rtol=1e-6)   # the trust-region radius
    assert np.allclose(rg["trace_error"], ro["trace_error"], rtol=1e-6)
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * max(ro["final_error"], 1e-12)
    assert rg["final_error"] < rg["initial_error"]
    assert relerr(gb.get_values(), ob.get_values()) < 1e-6


# ---- robust noise models on the device ------------------------------------------------------------------------------
def test_robust_functions_and_optimization(gpu):
    """gtsam/linear/tests/testNoiseModel.cpp:476-569 (Huber / Cauchy / Tukey weight and loss) and
    tests/testNonlinearOptimizer.cpp:351-482 (GN, LM, Dogleg with Huber factors) through the HIP path."""
    from tests.test_oracle_golden import ROBUST_FUNCTIONS, robust_function_check, robust_optimization_check
    for name, k, table in ROBUST_FUNCTIONS:
        robust_function_check(gpu.product_backend, name, k, table)
    robust_optimization_check(gpu.product_backend)


def test_robust_step_parity(gpu, oracle):
    """A pose graph with outlier loop closures under Huber / Tukey / Cauchy: errors, Jacobians, damped step and LM run
    against the oracle."""
    base = PROBLEMS["pose2"]
    rng = np.random.default_rng(9)
    kinds, ptr, vals = [], [0], []
    for f in range(base.n_factors):
        m = int(base.f_rows[f])
        p = base.noise[base.f_noise_ptr[f]:base.f_noise_ptr[f + 1]]
        k = int(base.f_noise_kind[f])
        if base.f_type[f] == A.F_BETWEEN:
            code = (A.NOISE_ROBUST_HUBER, A.NOISE_ROBUST_TUKEY, A.NOISE_ROBUST_CAUCHY)[f % 3]
            k |= code
            p = np.concatenate([p, [rng.uniform(0.5, 3.0) if code != A.NOISE_ROBUST_TUKEY else rng.uniform(20.0, 40.0)]])
        kinds.append(k)
        vals.append(p)
        ptr.append(ptr[-1] + p.size)
    arr = A.ProblemArrays(base.var_keys, base.var_types, base.var_dims, base.f_type, base.f_rows, base.f_key_ptr,
                          base.f_vars, base.f_meas_ptr, base.meas, kinds, ptr, np.concatenate(vals), base.values,
                          dict(base.meta))
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    assert abs(gb.error() - ob.error()) <= 1e-11 * abs(ob.error())
    ordering = gb.compute_ordering(A.ORDER_ND)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    gb.linearize()
    ob.linearize()
    jg, jo = gb.jacobians(), ob.jacobians()
    assert np.max(np.abs(jg - jo)) <= 1e-11 * max(1.0, np.max(np.abs(jo)))
    assert relerr(gb.solve(1e-3, False), ob.solve(1e-3, False)) < 1e-8
    p = A.lm_params_legacy()
    p.max_iterations = 8
    rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"]


# ---- GenericProjectionFactor on the device ----------------------------------------------------------------------------
def _visual_slam_arrays(n_poses=12, n_points=80, seed=5, with_outliers=False):
    """Poses on a circle looking at a cloud of landmarks (examples/VisualISAM2Example.cpp style), Cal3_S2, pixel noise
    sigma 1; priors on the first pose and the first landmark as SFMExample.cpp does."""
    from gtsam_petercdev_amd.graph import Cal3_S2, GenericProjectionFactor
    rng = np.random.default_rng(seed)
    K = Cal3_S2(50.0, 50.0, 0.0, 50.0, 50.0)
    pts = rng.uniform(-2, 2, (n_points, 3))
    g, v = NonlinearFactorGraph(), Values()
    noise = noiseModel.Isotropic.Sigma(2, 1.0)
    if with_outliers:
        noise = noiseModel.Robust.Create(noiseModel.mEstimator.Huber.Create(1.345), noise)
    poses = []
    for i in range(n_poses):
        th = 2 * math.pi * i / n_poses
        t = np.array([8 * math.cos(th), 8 * math.sin(th), 1.0])
        zc = -t / np.linalg.norm(t)
        xc = np.cross([0, 0, 1.0], zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc], axis=1)
        poses.append((R, t))
        for j in range(n_points):
            q = R.T @ (pts[j] - t)
            if q[2] <= 0.5:
                continue
            uv = np.array([K.v[0] * q[0] / q[2] + K.v[3], K.v[1] * q[1] / q[2] + K.v[4]]) + rng.normal(0, 1.0, 2)
            if with_outliers and rng.random() < 0.02:
                uv += rng.normal(0, 60.0, 2)
            g.add(GenericProjectionFactor(uv, noise, X(i), L(j), K))
        v.insert(X(i), Pose3(Rot3(R), Point3(*(t + rng.normal(0, 0.05, 3)))))
    for j in range(n_points):
        v.insert(L(j), Point3(*(pts[j] + rng.normal(0, 0.05, 3))))
    g.add(PriorFactor(X(0), Pose3(Rot3(poses[0][0]), Point3(*poses[0][1])),
                      noiseModel.Diagonal.Sigmas([0.1] * 3 + [0.3] * 3)))
    g.add(PriorFactor(L(0), Point3(*pts[0]), noiseModel.Isotropic.Sigma(3, 0.1)))
    return g.to_arrays(v)


def test_ProjectionFactor(gpu):
    """gtsam/slam/tests/testProjectionFactor.cpp:96-115,141-163 through the HIP path."""
    from tests.test_oracle_golden import projection_factor_check
    projection_factor_check(gpu.product_backend)


@pytest.mark.parametrize("with_outliers", [False, True])
def test_visual_slam_parity(gpu, oracle, with_outliers):
    """A visual-SLAM problem of GenericProjectionFactors (plain and Huber-robust): errors, Jacobians, damped steps and
    an LM run against the oracle."""
    arr = _visual_slam_arrays(with_outliers=with_outliers)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    assert abs(gb.error() - ob.error()) <= 1e-11 * abs(ob.error())
    for kind in (A.ORDER_MINDEGREE, A.ORDER_SCHUR_ND):
        ordering = gb.compute_ordering(kind)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.linearize()
        ob.linearize()
        jg, jo = gb.jacobians(), ob.jacobians()
        assert np.max(np.abs(jg - jo)) <= 1e-11 * max(1.0, np.max(np.abs(jo)))
        for lam, diag in ((1e-3, False), (1.0, True)):
            assert relerr(gb.solve(lam, diag), ob.solve(lam, diag)) < 1e-8
    p = A.lm_params_legacy()
    p.max_iterations = 10
    rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"]
    assert rg["final_error"] < rg["initial_error"]  # (the residual floor is the pixel noise itself)


# ---- marginal covariances (gsx_marginal_covariance) ---------------------------------------------------------------
def test_planarSLAMmarginals(gpu):
    """tests/testMarginals.cpp:40-107: the reference's expected marginal covariances, for several elimination orders."""
    from tests.test_oracle_golden import planar_slam_marginals_check
    planar_slam_marginals_check(gpu.product_backend, [[1, 2, 3, 11, 12], [11, 12, 1, 2, 3], [3, 12, 2, 11, 1]])


@pytest.mark.parametrize("name", ["bal_small", "bal_bigfront", "bal_wide_landmarks", "pose2", "pose3"])
@pytest.mark.parametrize("relax", [0.0, 0.5])
def test_marginal_covariance_matches_oracle(gpu, oracle, name, relax):
    """Marginal covariance blocks from the device factorization (forward substitution up the clique path) against the
    oracle's dense inverse: first / last eliminated variables, landmarks (lean leaves), cameras (big fronts)."""
    arr = PROBLEMS[name]
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    kind = A.ORDER_SCHUR_ND if name.startswith("bal") else A.ORDER_ND
    ordering = gb.compute_ordering(kind)
    gb.set_amalgamation(relax, 128)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    gb.linearize()
    ob.linearize()
    keys = [ordering[0], ordering[1], ordering[len(ordering) // 2], ordering[-2], ordering[-1]]
    for key in keys:
        cg, co = gb.marginal_covariance(key), ob.marginal_covariance(key)
        assert np.allclose(cg, cg.T, rtol=1e-9, atol=1e-14 * np.max(np.abs(co)))
        assert np.max(np.abs(cg - co)) <= 1e-7 * np.max(np.abs(co)), (name, key)
    # the factorization left behind is the undamped one: a following solve is unaffected
    assert relerr(gb.solve(1e-3, False), ob.solve(1e-3, False)) < 1e-8


# ---- BearingRangeFactor<Pose2, Point2> on the device -------------------------------------------------------------------
def test_BearingRangeFactor2D(gpu):
    """tests/testMarginals.cpp:40-107 with the nonlinear factors, testBearingRangeFactor.cpp's derivative check, angle
    wrapping and an LM run, all through the HIP path."""
    from tests.test_oracle_golden import bearing_range_check
    bearing_range_check(gpu.product_backend, [[1, 2, 3, 11, 12], [11, 12, 1, 2, 3]])


def _planar_slam_arrays(n_poses=400, n_landmarks=120, seed=11, robust=False):
    """A robot driving a noisy spiral among landmarks (examples/PlanarSLAMExample.cpp at scale): odometry
    BetweenFactor<Pose2> + BearingRangeFactor<Pose2, Point2> to the landmarks within 6 m, prior on the first pose."""
    from gtsam_petercdev_amd.graph import BearingRangeFactor
    rng = np.random.default_rng(seed)
    lms = rng.uniform(-15, 15, (n_landmarks, 2))
    g, v = NonlinearFactorGraph(), Values()
    odo = noiseModel.Diagonal.Sigmas([0.05, 0.05, 0.02])
    br = noiseModel.Diagonal.Sigmas([0.03, 0.1])
    if robust:
        br = noiseModel.Robust.Create(noiseModel.mEstimator.Cauchy.Create(1.0), br)
    x, y, th = 0.0, 0.0, 0.0
    ex, ey, eth = 0.0, 0.0, 0.0  # dead-reckoned estimate
    g.add(PriorFactor(X(0), Pose2(0, 0, 0), noiseModel.Diagonal.Sigmas([0.1, 0.1, 0.05])))
    seen = set()
    for i in range(n_poses):
        if i:
            u = np.array([0.5, 0.0, 0.04 + 0.02 * math.sin(i / 17.0)])
            x, y, th = x + math.cos(th) * u[0], y + math.sin(th) * u[0], th + u[2]
            z = u + rng.normal(0, [0.05, 0.05, 0.02])
            g.add(BetweenFactor(X(i - 1), X(i), Pose2(*z), odo))
            ex, ey = ex + math.cos(eth) * z[0] - math.sin(eth) * z[1], ey + math.sin(eth) * z[0] + math.cos(eth) * z[1]
            eth += z[2]
        v.insert(X(i), Pose2(ex, ey, eth))
        for j in range(n_landmarks):
            d = lms[j] - (x, y)
            r = math.hypot(*d)
            if r < 6.0:
                b = math.atan2(d[1], d[0]) - th + rng.normal(0, 0.03)
                g.add(BearingRangeFactor(X(i), L(j), b, r + rng.normal(0, 0.1), br))
                if j not in seen:  # initialise where the first sighting puts it
                    seen.add(j)
                    v.insert(L(j), Point2(ex + r * math.cos(eth + b), ey + r * math.sin(eth + b)))
    return g.to_arrays(v)


@pytest.mark.parametrize("robust", [False, True])
def test_planar_slam_parity(gpu, oracle, robust):
    """Planar SLAM with bearing-range measurements: errors, Jacobians, damped steps, an LM trajectory and landmark
    marginals against the oracle."""
    arr = _planar_slam_arrays(robust=robust)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    assert abs(gb.error() - ob.error()) <= 1e-11 * abs(ob.error())
    for kind in (A.ORDER_MINDEGREE, A.ORDER_ND):
        ordering = gb.compute_ordering(kind)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.linearize()
        ob.linearize()
        jg, jo = gb.jacobians(), ob.jacobians()
        assert np.max(np.abs(jg - jo)) <= 1e-11 * max(1.0, np.max(np.abs(jo)))
        for lam, diag in ((1e-3, False), (1.0, True)):
            assert relerr(gb.solve(lam, diag), ob.solve(lam, diag)) < 1e-8
    p = A.lm_params_legacy()
    p.max_iterations = 15
    rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"]
    assert rg["final_error"] < (0.5 if robust else 0.05) * rg["initial_error"]
    gb.linearize()
    ob.linearize()
    gb.solve(0.0, False)
    ob.solve(0.0, False)
    for key in (int(arr.var_keys[0]), int(arr.var_keys[-1]), L(0) if L(0) in set(arr.var_keys.tolist()) else int(arr.var_keys[1])):
        cg, co = gb.marginal_covariance(key), ob.marginal_covariance(key)
        assert np.max(np.abs(cg - co)) <= 1e-7 * np.max(np.abs(co))


def test_lm_trial_equals_separate_calls(gpu):
    """gsx_lm_trial = linearize + damped solve + both linearized errors + retract + trial error in one call."""
    arr = PROBLEMS["pose3"]
    a, b = gpu.product_backend(arr), gpu.product_backend(arr)
    ordering = a.compute_ordering(A.ORDER_ND)
    a.set_ordering(ordering)
    b.set_ordering(ordering)
    for lam, diag in ((1e-3, False), (0.5, True)):
        e0, ed, et = a.lm_trial(True, lam, diag)
        b.linearize()
        b.solve(lam, diag, want_delta=False)
        f0, fd = b.linear_error()
        ft = b.retract(None, commit=False)
        assert et == ft
        # the trial takes its two linearized errors from the right-hand sides and the solved step
        # (e(delta) = e(0) - 1/2 g'delta - 1/2 lambda delta'D delta), gsx_linear_error evaluates 1/2 sum |A delta - b|^2
        assert abs(e0 - f0) <= 1e-13 * f0
        assert abs((e0 - ed) - (f0 - fd)) <= 1e-9 * (f0 - fd) + 1e-13 * f0
    with pytest.raises(gt.GsxError) as ei:
        gpu.product_backend(arr).lm_trial(True, 0.0, False)   # no ordering yet
    assert ei.value.status == A.GSX_E_STATE


def test_toro_example_runs_match_oracle(gpu, oracle, golden_dir):
    """The reference's example drivers on its own TORO / "graph" data (w100.graph, example.graph with bearing-range
    factors, sphere2500.txt): LM on the device against the oracle — same accept/reject trace, final error to 1e-6."""
    from tests.test_oracle_golden import toro_example_problems
    for name, arr, kind in toro_example_problems(golden_dir):
        gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
        ordering = gb.compute_ordering(kind)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        assert abs(gb.error() - ob.error()) <= 1e-11 * abs(ob.error()), name
        p = A.lm_params_legacy()
        p.max_iterations = 12 if name == "sphere2500" else 100
        rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
        assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"]), name
        assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"], name
        assert rg["final_error"] < 0.25 * rg["initial_error"], name


def test_planarSLAMjointMarginals(gpu):
    """tests/testMarginals.cpp:109-157 through the HIP path, for several elimination orders."""
    from tests.test_oracle_golden import planar_slam_joint_marginals_check
    planar_slam_joint_marginals_check(gpu.product_backend, [[1, 2, 3, 11, 12], [11, 12, 1, 2, 3], [3, 12, 2, 11, 1]])


@pytest.mark.parametrize("name", ["bal_small", "pose3", "bal_bigfront"])
def test_joint_marginal_matches_oracle(gpu, oracle, name):
    """Joint covariances of variables in different subtrees / the same clique / leaf and root, against the oracle's dense
    inverse, with relaxed amalgamation."""
    arr = PROBLEMS[name]
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(A.ORDER_SCHUR_ND if name.startswith("bal") else A.ORDER_ND)
    gb.set_amalgamation(0.5, 128)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    gb.linearize()
    ob.linearize()
    keys = [int(ordering[0]), int(ordering[1]), int(ordering[len(ordering) // 2]), int(ordering[-1]), int(ordering[-2])]
    Jg, Jo = gb.joint_marginal_covariance(keys), ob.joint_marginal_covariance(keys)
    assert np.max(np.abs(Jg - Jo)) <= 1e-7 * np.max(np.abs(Jo))
    with pytest.raises(gt.GsxError):
        gb.joint_marginal_covariance([keys[0], keys[0]])


# ---- partial relinearization / re-elimination (gsx_relinearize_partial) -------------------------------------------
def _state_slices(arr):
    off = np.concatenate([[0], np.cumsum(arr.state_dims())])
    return off


@pytest.mark.parametrize("name", ["pose3", "pose2", "bal_small", "bal_bigfront", "visual_slam"])
@pytest.mark.parametrize("relax", [0.0, 0.5])
def test_partial_reelimination_is_bit_identical(gpu, name, relax):
    """Move a few variables, redo only what they touch: the factorization, the solution, the Jacobians and a marginal are
    bit for bit those of a full relinearization + elimination at the same values; several updates in a row; what was
    redone is a fraction of the tree."""
    arr = _visual_slam_arrays() if name == "visual_slam" else PROBLEMS[name]
    kind = A.ORDER_SCHUR_ND if name.startswith("bal") or name == "visual_slam" else A.ORDER_ND
    P, F = gpu.product_backend(arr), gpu.product_backend(arr)
    ordering = P.compute_ordering(kind)
    for be in (P, F):
        be.set_amalgamation(relax, 128)
        be.set_ordering(ordering)
    with pytest.raises(gt.GsxError) as ei:
        P.relinearize_partial([int(arr.var_keys[0])])      # nothing resident yet
    assert ei.value.status == A.GSX_E_STATE
    P.linearize()
    d0 = P.solve(0.0, False)
    # where a Gauss-Newton step would move everything
    F.linearize()
    F.solve(0.0, False)
    F.retract(None, commit=True)
    x0, x1 = arr.values.copy(), F.get_values()
    off = _state_slices(arr)
    rng = np.random.default_rng(7)
    current = x0.copy()
    for round_ in range(3):
        n_move = max(1, arr.n_vars // (40 if round_ < 2 else 6))
        # a stretch of consecutive variables (one region of the trajectory / a few cameras) first, scattered sets after
        idx = np.arange(n_move) if round_ == 0 else np.sort(rng.choice(arr.n_vars, n_move, replace=False))
        target = x0 if round_ == 2 else x1        # two sets forward to the Gauss-Newton point, then a large set back
        states = np.concatenate([target[off[i]:off[i + 1]] for i in idx])
        for i in idx:
            current[off[i]:off[i + 1]] = target[off[i]:off[i + 1]]
        stats = P.relinearize_partial(arr.var_keys[idx], states)
        assert 0 < stats["n_fronts_reeliminated"] <= stats["n_fronts"]
        # (a set that dirties most of a small tree takes the full path inside the call: same bits)
        dp = P.solve(0.0, False)
        F.set_values(current)
        F.linearize()
        df = F.solve(0.0, False)
        assert np.array_equal(P.get_values(), current)
        assert np.array_equal(P.jacobians(), F.jacobians()), (name, round_)
        assert np.array_equal(dp, df), (name, round_, float(np.max(np.abs(dp - df))))
        assert np.array_equal(P.hessian_diagonal(), F.hessian_diagonal())
        k = int(arr.var_keys[idx[0]])
        assert np.array_equal(P.marginal_covariance(k), F.marginal_covariance(k))
    assert not np.array_equal(dp, d0)
    with pytest.raises(gt.GsxError):
        P.relinearize_partial([int(arr.var_keys[0]), int(arr.var_keys[0])])


def test_partial_reelimination_touches_a_fraction(gpu):
    """3000 poses, the 10 most recent ones move (an iSAM2-style update): a few dozen of the ~700 cliques are redone, the
    result is bit for bit the full one, and a second update on top of the first stays exact."""
    arr = datasets.synth_manhattan_pose3(3000, seed=9)
    P, F = gpu.product_backend(arr), gpu.product_backend(arr)
    ordering = P.compute_ordering(A.ORDER_ND)
    for be in (P, F):
        be.set_ordering(ordering)
    P.linearize()
    P.solve(0.0, False)
    F.linearize()
    F.solve(0.0, False)
    F.retract(None, commit=True)
    x1 = F.get_values()
    off = _state_slices(arr)
    current = arr.values.copy()
    for idx in (np.arange(arr.n_vars - 10, arr.n_vars), np.arange(1500, 1520)):
        states = np.concatenate([x1[off[i]:off[i + 1]] for i in idx])
        for i in idx:
            current[off[i]:off[i + 1]] = x1[off[i]:off[i + 1]]
        stats = P.relinearize_partial(arr.var_keys[idx], states)
        assert stats["n_fronts_reeliminated"] < 0.15 * stats["n_fronts"], stats
        assert stats["n_factors_relinearized"] < 0.05 * arr.n_factors, stats
        dp = P.solve(0.0, False)
        F.set_values(current)
        F.linearize()
        assert np.array_equal(dp, F.solve(0.0, False))


def test_empty_and_ragged_inputs(gpu, oracle):
    """Edge inputs: an empty graph (the reference's optimizers return at once: NonlinearOptimizer.cpp:75-83, error 0); a
    variable that no factor touches next to a well-posed part (indeterminate at lambda = 0 with ITS key, solvable once
    damped); factors of different row counts and noise kinds on one variable."""
    g, v = NonlinearFactorGraph(), Values()
    arr = g.to_arrays(v)
    for be in (gpu.product_backend(arr), oracle.oracle_backend(arr)):
        be.set_ordering([])
        assert be.error() == 0.0
        r = be.lm_optimize(A.lm_params_legacy())
        assert r["iterations"] == 0 and r["final_error"] == 0.0
        assert be.get_values().size == 0
    # a loose variable
    g = NonlinearFactorGraph()
    g.addPrior(1, Pose2(0., 0., 0.), noiseModel.Isotropic.Sigma(3, 0.1))
    g.add(BetweenFactor(1, 2, Pose2(1., 0., 0.), noiseModel.Isotropic.Sigma(3, 0.2)))
    v = Values()
    v.insert(1, Pose2(0.1, 0., 0.))
    v.insert(2, Pose2(1.2, 0.1, 0.))
    v.insert(9, Point2(3.0, 4.0))        # nobody measures it
    arr = g.to_arrays(v)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    for be in (gb, ob):
        be.set_ordering([9, 1, 2])
        be.linearize()
        with pytest.raises(gt.IndeterminantLinearSystemException) as ei:
            be.solve(0.0)
        assert ei.value.key == 9
    assert relerr(gb.solve(1e-3), ob.solve(1e-3)) < 1e-10   # lambda I makes it positive definite
    assert np.allclose(gb.solve(1e-3)[-2:], 0.0)            # (tangent order follows the keys 1, 2, 9: the loose one is last)
    # ragged: priors of 2 and 3 rows, unit / diagonal / full Gaussian noise on the same landmark and pose
    g = NonlinearFactorGraph()
    g.addPrior(1, Pose2(0., 0., 0.), noiseModel.Diagonal.Sigmas([0.1, 0.2, 0.05]))
    g.addPrior(5, Point2(1.0, 1.0), noiseModel.Unit.Create(2))
    g.addPrior(5, Point2(1.2, 0.9), noiseModel.Gaussian.Covariance(np.array([[0.04, 0.01], [0.01, 0.09]])))
    from gtsam_petercdev_amd.graph import BearingRangeFactor
    g.add(BearingRangeFactor(1, 5, 0.7, 1.5, noiseModel.Isotropic.Sigma(2, 0.1)))
    v = Values()
    v.insert(1, Pose2(0.05, -0.02, 0.01))
    v.insert(5, Point2(0.8, 1.3))
    arr = g.to_arrays(v)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    for be in (gb, ob):
        be.set_ordering([5, 1])
        be.linearize()
    assert np.max(np.abs(gb.jacobians() - ob.jacobians())) < 1e-12
    assert relerr(gb.solve(0.0), ob.solve(0.0)) < 1e-11
    rg, ro = gb.lm_optimize(A.lm_params_legacy()), ob.lm_optimize(A.lm_params_legacy())
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-9 * max(ro["final_error"], 1e-12)


def test_assembly_kernel_corner_shapes(gpu, oracle):
    """The specialised H-assembly kernels at their edges: a landmark seen by 80 cameras (two record chunks of the star
    kernel), a landmark observed twice by the same camera (NOT a star: the two partner blocks must be summed — generic
    kernel), cameras with > 64 factors including the 9-row prior (matrix-core kernel, tall-factor branch)."""
    rng = np.random.default_rng(3)
    base = datasets.synth_bal_arrays(80, 3, 150, seed=21, long_range=1.0, priors=True)
    # ... and every camera that does not see landmark 0 yet gets an observation of it: > 64 factors on that landmark
    n_sfm = int((base.f_type == A.F_SFM).sum())
    cams_of = lambda a, lm: set(a.f_vars[0:2 * n_sfm:2][a.f_vars[1:2 * n_sfm:2] == lm].tolist())
    lm0 = 80
    z = base.meas[base.f_meas_ptr[0]:base.f_meas_ptr[1]]
    for c in sorted(set(range(80)) - cams_of(base, lm0)):
        base = base.with_factor(A.F_SFM, [c, lm0], 2, z + rng.normal(0, 2.0, 2), int(base.f_noise_kind[0]),
                                base.noise[base.f_noise_ptr[0]:base.f_noise_ptr[1]])
    assert base.n_factors - 152 + len(cams_of(base, lm0)) > 64
    wide = datasets.synth_bal_arrays(4, 200, 760, seed=22, long_range=1.0, priors=True)   # ~190 factors per camera
    # duplicate an observation: camera c sees landmark l twice
    sfm = np.nonzero(wide.f_type == A.F_SFM)[0]
    f = int(sfm[5])
    kp = wide.f_key_ptr[f]
    dup = wide.with_factor(A.F_SFM, wide.f_vars[kp:kp + 2], 2,
                           wide.meas[wide.f_meas_ptr[f]:wide.f_meas_ptr[f + 1]] + rng.normal(0, 0.5, 2),
                           int(wide.f_noise_kind[f]), wide.noise[wide.f_noise_ptr[f]:wide.f_noise_ptr[f + 1]])
    for arr in (base, wide, dup):
        gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
        ordering = gb.compute_ordering(A.ORDER_SCHUR)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.linearize()
        ob.linearize()
        assert relerr(gb.hessian_diagonal(), ob.hessian_diagonal()) < 1e-12
        for lam, diag in ((1e-3, False), (1.0, True)):
            assert relerr(gb.solve(lam, diag), ob.solve(lam, diag)) < 1e-8


@pytest.mark.parametrize("m", [20, 21, 22, 23])
def test_leaf_clique_with_many_trailing_rows(gpu, oracle, m):
    """A childless one-variable clique whose separator has m Pose2 neighbours: 3 m + 1 trailing rows under 3 frontal
    columns.  Its Schur complement is an outer product with a thread per trailing row, so the launch needs at least
    that many threads: 22 neighbours = 67 rows sits just above one wave (a 64-thread launch left rows 64.. unwritten)."""
    g, v = NonlinearFactorGraph(), Values()
    rng = np.random.default_rng(m)
    noise = noiseModel.Diagonal.Sigmas(np.array([0.2, 0.2, 0.1]))
    v.insert(0, Pose2(0.1, -0.1, 0.05))
    for k in range(1, m + 1):
        a = 2 * np.pi * k / m
        v.insert(k, Pose2(2 * np.cos(a) + rng.normal(0, 0.05), 2 * np.sin(a) + rng.normal(0, 0.05), a + rng.normal(0, 0.02)))
        g.add(BetweenFactor(0, k, Pose2(2 * np.cos(a), 2 * np.sin(a), a), noise))
        g.addPrior(k, Pose2(2 * np.cos(a), 2 * np.sin(a), a), noiseModel.Isotropic.Sigma(3, 0.5))
        if k > 1:
            g.add(BetweenFactor(k - 1, k, Pose2(0.3, 0.2, 2 * np.pi / m), noiseModel.Isotropic.Sigma(3, 0.3)))
    # one more pose behind pose 1, eliminated last: the neighbours' clique then holds a variable the hub does not see, so
    # the reference's merge rule (child separator == all of the parent) leaves the hub's clique alone
    v.insert(m + 1, Pose2(3.0, 0.5, 0.3))
    g.add(BetweenFactor(1, m + 1, Pose2(1.0, 0.3, 0.2), noise))
    g.addPrior(m + 1, Pose2(3.0, 0.5, 0.3), noiseModel.Isotropic.Sigma(3, 0.5))
    arr = g.to_arrays(v)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    order = list(range(m + 2))            # the hub first: its clique is a leaf with all m neighbours in the separator
    for be in (gb, ob):
        be.set_amalgamation(0.0, 128) if be is gb else None
        be.set_ordering(order)
        be.linearize()
    parent, fronts = gb.get_tree()
    hub = [c for c, (f, s) in enumerate(fronts) if f == [0]]
    assert hub and len(fronts[hub[0]][1]) == m
    for lam in (0.0, 1e-3):
        assert relerr(gb.solve(lam, False), ob.solve(lam, False)) < 1e-9, (m, lam)
    assert relerr(gb.marginal_covariance(1), ob.marginal_covariance(1)) < 1e-8


@pytest.mark.parametrize("medium", [False, True])
@pytest.mark.parametrize("seed,dense", [(1, False), (2, True)])
def test_linear_graph_with_wide_variables(gpu, oracle, seed, dense, medium, monkeypatch):
    """Vector variables of 1 to 40 dimensions (blocks wider than one 16 x 16 matrix-core tile: the gather's wide path, H
    panels and fronts with ragged blocks) in a linear-Gaussian graph — the NonlinearOptimizer::solve seam hands over
    whatever dimensions the caller's variables have.  dense=True adds a cluster that makes blocked (n > 140) fronts with
    wide children below them."""
    # medium=True: the MEDIUM-front path (gsx_internal.h; frontal panel in LDS, trailing block in HBM — measured slower on
    # the bench graphs and off by default), which these graphs' 150-290-row cliques take
    if medium:
        monkeypatch.setenv("GSX_MEDIUM", "1")
    rng = np.random.default_rng(seed)
    dims = [int(d) for d in rng.choice([1, 2, 5, 9, 17, 24, 33, 40], size=36)]
    fg = GaussianFactorGraph()
    def block(m, d):
        return rng.normal(0, 0.3, (m, d))
    for k, d in enumerate(dims):
        fg.add(JacobianFactor(k, np.eye(d) + 0.1 * rng.normal(size=(d, d)), rng.normal(size=d),
                              noiseModel.Isotropic.Sigma(d, 1.0 + 0.1 * k)))
    pairs = [(k, k + 1) for k in range(len(dims) - 1)] + [(int(a), int(b)) for a, b in rng.integers(0, len(dims), (30, 2)) if a != b]
    if dense:
        pairs += [(a, b) for a in range(8, 16) for b in range(a + 1, 16)]
    for a, b in pairs:
        m = max(1, min(dims[a], dims[b], 6))
        fg.add(JacobianFactor(a, block(m, dims[a]), b, block(m, dims[b]), rng.normal(size=m),
                              noiseModel.Diagonal.Sigmas(0.5 + rng.random(m))))
    arr = fg.to_arrays(None)
    arr.values = np.zeros(int(arr.var_dims.sum()))
    assert int(arr.var_dims.max()) > 16
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    for kind, relax in ((A.ORDER_MINDEGREE, 0.0), (A.ORDER_ND, 0.0), (A.ORDER_NATURAL, None)):
        ordering = gb.compute_ordering(kind)
        if relax is not None:
            gb.set_amalgamation(relax, 128)
        else:
            gb.set_amalgamation(0.5, 64)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.linearize()
        ob.linearize()
        assert relerr(gb.hessian_diagonal(), ob.hessian_diagonal()) < 1e-12
        for lam, diag in ((0.0, False), (1e-2, True)):
            assert relerr(gb.solve(lam, diag), ob.solve(lam, diag)) < 1e-9, (kind, lam)
        gb.solve(0.0, False)
        ob.solve(0.0, False)
        # marginals of the widest variable of at most 16 dimensions (one pass of the path kernel) and of the widest of all
        # (Marginals.cpp:107-136 knows no limit: its columns go through the kernel in groups), and a joint of the two
        kw = int(arr.var_keys[int(np.argmax(np.where(arr.var_dims <= 16, arr.var_dims, 0)))])
        assert relerr(gb.marginal_covariance(kw), ob.marginal_covariance(kw)) < 1e-8
        kx = int(arr.var_keys[int(np.argmax(arr.var_dims))])
        assert relerr(gb.marginal_covariance(kx), ob.marginal_covariance(kx)) < 1e-8
        assert relerr(gb.joint_marginal_covariance([kw, kx]), ob.joint_marginal_covariance([kw, kx])) < 1e-8
    st = gb.stats()
    if dense:
        assert st["n_big_fronts"] > 0
    assert (st["n_medium_fronts"] > 0) == medium   # (NATURAL ordering, the last one set: 150-290-row cliques)


@pytest.mark.parametrize("seed", range(24))
def test_random_linear_graphs(gpu, oracle, seed):
    """Structure fuzz: random linear-Gaussian graphs — chains with random chords, hubs with many neighbours (leaf cliques
    with tall separators), dense clusters (blocked fronts), variables of 1-9 dimensions — under every ordering and three
    amalgamation settings, against the oracle.  Aimed at the boundaries between the kernels' size classes."""
    rng = np.random.default_rng(1000 + seed + FUZZ_OFFSET)
    nv = int(rng.choice([4, 9, 30, 70, 140, 260]))
    dim_sets = ([3], [6], [1, 2, 3], [2, 6, 9], [9, 3])
    ds = dim_sets[seed % len(dim_sets)]
    dims = [int(d) for d in rng.choice(ds, size=nv)]
    fg = GaussianFactorGraph()
    for k, d in enumerate(dims):
        fg.add(JacobianFactor(k, np.eye(d) * (0.5 + rng.random()), rng.normal(size=d), noiseModel.Isotropic.Sigma(d, 2.0)))
    pairs = {(k, k + 1) for k in range(nv - 1)}
    for a, b in rng.integers(0, nv, (int(nv * rng.choice([0.2, 1.0, 2.5])), 2)):
        if a != b:
            pairs.add((int(min(a, b)), int(max(a, b))))
    for hub in rng.integers(0, nv, 2):                       # hubs: one variable with up to 40 neighbours
        for b in rng.choice(nv, size=min(nv - 1, int(rng.choice([5, 22, 40]))), replace=False):
            if int(b) != int(hub):
                pairs.add((int(min(hub, b)), int(max(hub, b))))
    if nv >= 30 and seed % 3 == 0:                           # a dense cluster
        c0 = int(rng.integers(0, nv - 20))
        pairs |= {(a, b) for a in range(c0, c0 + 18) for b in range(a + 1, c0 + 18)}
    for a, b in sorted(pairs):
        m = int(rng.integers(1, 1 + min(dims[a] + dims[b], 6)))
        fg.add(JacobianFactor(a, rng.normal(0, 0.4, (m, dims[a])), b, rng.normal(0, 0.4, (m, dims[b])), rng.normal(size=m),
                              noiseModel.Diagonal.Sigmas(0.5 + rng.random(m))))
    arr = fg.to_arrays(None)
    arr.values = np.zeros(int(arr.var_dims.sum()))
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    kinds = [A.ORDER_NATURAL, A.ORDER_MINDEGREE, A.ORDER_ND]
    for kind, amalg in zip(kinds, ((0.0, 128), None, (1.0, 48))):
        ordering = gb.compute_ordering(kind)
        if kind == A.ORDER_NATURAL and seed % 2:
            ordering = ordering[::-1].copy()
        if amalg is not None:
            gb.set_amalgamation(*amalg)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.linearize()
        ob.linearize()
        for lam, diag in ((0.0, False), (1e-2, True)):
            assert relerr(gb.solve(lam, diag), ob.solve(lam, diag)) < 1e-8, (seed, nv, kind, lam)
        e_g, e_o = gb.linear_error(), ob.linear_error()
        assert abs(e_g[1] - e_o[1]) <= 1e-9 * max(abs(e_o[1]), 1e-12)
        # marginals from the undamped factorization: a variable deep in the tree, one near the root, a joint of three
        gb.solve(0.0, False)
        ob.solve(0.0, False)
        ks = [int(ordering[0]), int(ordering[-1]), int(ordering[len(ordering) // 2])]
        for k in ks[:2]:
            assert relerr(gb.marginal_covariance(k), ob.marginal_covariance(k)) < 1e-7, (seed, kind, k)
        if len(set(ks)) == 3:
            assert relerr(gb.joint_marginal_covariance(sorted(ks)), ob.joint_marginal_covariance(sorted(ks))) < 1e-7


def _split_dims(total, rng):
    out = []
    while total > 0:
        d = int(min(total, rng.integers(1, 9)))
        out.append(d)
        total -= d
    return out


@pytest.mark.parametrize("dim_a", [15, 17, 33, 64, 65, 127])
def test_two_clique_medium_fronts(gpu, oracle, dim_a, monkeypatch):
    """The same two-clique trees with the MEDIUM-front path switched on (off by default: DESIGN §4): the child (A | B) with
    131 separator scalars has more than 140 rows and a panel that fits LDS."""
    monkeypatch.setenv("GSX_MEDIUM", "1")
    _two_clique_case(gpu, oracle, dim_a, 131)


@pytest.mark.parametrize("dim_b", [12, 60, 131])
@pytest.mark.parametrize("dim_a", [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 230])
def test_two_clique_size_classes(gpu, oracle, dim_a, dim_b):
    """A Bayes tree of two cliques, (A | B) under (B, c), with the frontal width of the child swept over the boundaries
    of the kernels' size classes (leaf kernel up to 16 frontal scalars; back-substitution kernels at 32 / 64; one chunk of
    the blocked factorization at 192) and its height over the LDS / blocked boundary (140 rows)."""
    _two_clique_case(gpu, oracle, dim_a, dim_b)


@pytest.mark.parametrize("dim_a", [1, 3, 6, 16])
def test_leaf_clique_heights(gpu, oracle, dim_a):
    """The same tree with a leaf-kernel child and the height of its separator swept row by row through the thread classes
    of the leaf launch (64 / 128 / 256 threads; a thread per trailing row of the stored complement) and over the point
    where the parent becomes a blocked front and the leaf keeps only its panel."""
    for dim_b in list(range(58, 76)) + list(range(104, 114)) + [127, 128, 129, 135, 136, 137, 138, 139, 140]:
        _two_clique_case(gpu, oracle, dim_a, dim_b, light=True)


def _two_clique_case(gpu, oracle, dim_a, dim_b, light=False):
    rng = np.random.default_rng(dim_a * 1000 + dim_b)
    da, db = _split_dims(dim_a, rng), _split_dims(dim_b, rng)
    dims = da + db + [3]
    nv = len(dims)
    fg = GaussianFactorGraph()
    for k, d in enumerate(dims):
        fg.add(JacobianFactor(k, np.eye(d) * (0.7 + rng.random()), rng.normal(size=d), noiseModel.Isotropic.Sigma(d, 1.5)))
    nab = len(da) + len(db)                # A and B mutually adjacent: a two-row factor on every pair
    for a in range(nab):
        for b in range(a + 1, nab):
            fg.add(JacobianFactor(a, rng.normal(0, 0.3, (2, dims[a])), b, rng.normal(0, 0.3, (2, dims[b])), rng.normal(size=2),
                                  noiseModel.Isotropic.Sigma(2, 1.0)))
    kb, kc = len(da), nv - 1               # the last variable hangs on B only: (A | B) keeps its own clique
    fg.add(JacobianFactor(kb, rng.normal(0, 0.4, (3, dims[kb])), kc, rng.normal(0, 0.4, (3, 3)), rng.normal(size=3),
                          noiseModel.Isotropic.Sigma(3, 0.5)))
    arr = fg.to_arrays(None)
    arr.values = np.zeros(int(arr.var_dims.sum()))
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    gb.set_amalgamation(0.0, 128)
    order = list(range(nv))
    gb.set_ordering(order)
    ob.set_ordering(order)
    parent, fronts = gb.get_tree()
    assert len(fronts) == 2 and sorted(len(f) for f, _ in fronts) == sorted([len(da), len(db) + 1])
    gb.linearize()
    ob.linearize()
    for lam, diag in ((0.0, False), (0.1, True)):
        assert relerr(gb.solve(lam, diag), ob.solve(lam, diag)) < 1e-9, (dim_a, dim_b, lam)
    gb.solve(0.0, False)
    ob.solve(0.0, False)
    for k in (0, kb, kc):
        assert relerr(gb.marginal_covariance(k), ob.marginal_covariance(k)) < 1e-8, (dim_a, dim_b, k)
    if light:
        gb.close()
        ob.close()
        return
    # the conditional of the child clique, entry by entry
    child = [c for c, (f, _) in enumerate(fronts) if f[0] == 0][0]
    po, fo = ob.get_tree()
    co = [c for c, (f, _) in enumerate(fo) if f[0] == 0][0]
    Rg, Ro = gb.conditional(child), ob.conditional(co)
    assert Rg.shape == Ro.shape and fo[co][1] == fronts[child][1]
    assert np.abs(Rg - Ro).max() <= 1e-9 * np.abs(Ro).max()


@pytest.mark.parametrize("seed", range(8))
def test_random_bal_structures(gpu, oracle, seed):
    """Structure fuzz of the bundle-adjustment kernels: small to medium camera systems (landmark leaves with stored or
    product-form complements), one landmark seen by every camera, one camera that sees a landmark it also sees through a
    second (duplicate) observation, under the Schur orderings, plain and relaxed trees: H diagonal, damped steps, three LM
    iterations against the oracle."""
    rng = np.random.default_rng(300 + seed + FUZZ_OFFSET)
    nc = int(rng.choice([3, 7, 12, 17, 30, 55]))
    npts = int(rng.choice([40, 150, 600]))
    nobs = int(npts * min(rng.choice([2.5, 4.0, 7.0]), 0.8 * nc))
    arr = datasets.synth_bal_arrays(nc, npts, max(nobs, 2 * npts), seed=300 + seed + FUZZ_OFFSET, long_range=float(rng.choice([0.0, 0.3, 1.0])),
                                    priors=True)
    n_sfm = int((arr.f_type == A.F_SFM).sum())
    cams = arr.f_vars[0:2 * n_sfm:2]
    pts = arr.f_vars[1:2 * n_sfm:2]
    lm0 = int(pts[0])
    z = arr.meas[arr.f_meas_ptr[0]:arr.f_meas_ptr[1]]
    nz = arr.noise[arr.f_noise_ptr[0]:arr.f_noise_ptr[1]]
    for c in sorted(set(range(nc)) - set(cams[pts == lm0].tolist())):      # lm0: seen by every camera
        arr = arr.with_factor(A.F_SFM, [c, lm0], 2, z + rng.normal(0, 2.0, 2), int(arr.f_noise_kind[0]), nz)
    if seed % 2:                                                            # a duplicate observation
        arr = arr.with_factor(A.F_SFM, [int(cams[3]), int(pts[3])], 2, arr.meas[arr.f_meas_ptr[3]:arr.f_meas_ptr[4]] + 0.3,
                              int(arr.f_noise_kind[3]), nz)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    for kind, amalg in ((A.ORDER_SCHUR, (0.0, 128)), (A.ORDER_SCHUR_ND, (0.5, 64)), (A.ORDER_MINDEGREE, None)):
        ordering = gb.compute_ordering(kind)
        if amalg is not None:
            gb.set_amalgamation(*amalg)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        gb.set_values(arr.values)
        ob.set_values(arr.values)
        gb.linearize()
        ob.linearize()
        assert relerr(gb.hessian_diagonal(), ob.hessian_diagonal()) < 1e-12
        for lam, diag in ((1e-3, False), (1.0, True)):
            assert relerr(gb.solve(lam, diag), ob.solve(lam, diag)) < 1e-8, (seed, kind, lam)
        p = A.lm_params_legacy()
        p.max_iterations = 3
        rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
        assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"]), (seed, kind)
        assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * ro["final_error"], (seed, kind)


def _random_pose2_graph(rng, nv, cluster):
    """Random Pose2 graph: a noisy chain with chords, one hub with up to 30 neighbours, optionally a dense cluster."""
    g, v = NonlinearFactorGraph(), Values()
    pos = np.cumsum(rng.normal(0.5, 0.2, (nv, 2)), axis=0)
    th = np.cumsum(rng.normal(0, 0.1, nv))
    for k in range(nv):
        v.insert(k, Pose2(pos[k, 0] + rng.normal(0, 0.03), pos[k, 1] + rng.normal(0, 0.03), th[k] + rng.normal(0, 0.01)))
    g.addPrior(0, Pose2(pos[0, 0], pos[0, 1], th[0]), noiseModel.Isotropic.Sigma(3, 0.1))
    pairs = {(k, k + 1) for k in range(nv - 1)}
    for a, b in rng.integers(0, nv, (int(nv * rng.choice([0.3, 1.5])), 2)):
        if a != b:
            pairs.add((int(min(a, b)), int(max(a, b))))
    hub = int(rng.integers(0, nv))
    for b in rng.choice(nv, size=min(nv - 1, int(rng.choice([8, 22, 30]))), replace=False):
        if int(b) != hub:
            pairs.add((min(hub, int(b)), max(hub, int(b))))
    if nv >= 60 and cluster:
        c0 = int(rng.integers(0, nv - 50))
        pairs |= {(a, b) for a in range(c0, c0 + 48) for b in range(a + 1, c0 + 48) if rng.random() < 0.6}
    for a, b in sorted(pairs):
        c, s_ = math.cos(th[a]), math.sin(th[a])      # pa^-1 pb
        dx, dy = pos[b, 0] - pos[a, 0], pos[b, 1] - pos[a, 1]
        zx, zy, zt = c * dx + s_ * dy, -s_ * dx + c * dy, th[b] - th[a]
        g.add(BetweenFactor(a, b, Pose2(zx + rng.normal(0, 0.05), zy + rng.normal(0, 0.05), zt + rng.normal(0, 0.02)),
                            noiseModel.Diagonal.Sigmas(np.array([0.2, 0.2, 0.1]))))
    return g, v


@pytest.mark.parametrize("seed", range(8))
def test_optimizers_on_random_structures(gpu, oracle, seed):
    """LM (legacy and Ceres policies), Gauss-Newton and Dogleg on random Pose2 structures: the oracle's accept/reject
    trace, lambda / trust-region schedule and errors."""
    rng = np.random.default_rng(900 + seed + FUZZ_OFFSET)
    nv = int(rng.choice([15, 80, 200]))
    g, v = _random_pose2_graph(rng, nv, seed % 2 == 0)
    arr = g.to_arrays(v)
    kind = [A.ORDER_ND, A.ORDER_MINDEGREE][seed % 2]
    for which in ("legacy", "ceres", "gn", "dogleg"):
        gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
        ordering = gb.compute_ordering(kind)
        gb.set_ordering(ordering)
        ob.set_ordering(ordering)
        if which in ("legacy", "ceres"):
            p = A.lm_params_legacy() if which == "legacy" else A.lm_params_ceres()
            p.max_iterations = 12
            rg, ro = gb.lm_optimize(p), ob.lm_optimize(p)
            assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"]), (seed, which)
            assert np.allclose(rg["trace_lambda"], ro["trace_lambda"], rtol=1e-8), (seed, which)
        elif which == "gn":
            rg, ro = gb.gn_optimize(8), ob.gn_optimize(8)
        else:
            rg, ro = gb.dogleg_optimize(1.0, 10), ob.dogleg_optimize(1.0, 10)
        assert rg["iterations"] == ro["iterations"], (seed, which)
        assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * max(ro["final_error"], 1e-12), (seed, which)
        assert relerr(gb.get_values(), ob.get_values()) < 1e-7, (seed, which)
        gb.close()
        ob.close()


@pytest.mark.parametrize("seed", range(10))
def test_partial_reelimination_on_random_structures(gpu, seed):
    """Structure fuzz of the filtered launch plans: random Pose2 graphs with chords, hubs and a dense cluster; random sets
    of moved variables; gsx_relinearize_partial + back-substitution must stay bit for bit the full path, and the
    wildfire pass at threshold 0 likewise."""
    rng = np.random.default_rng(500 + seed + FUZZ_OFFSET)
    nv = int(rng.choice([12, 60, 150, 400]))
    g, v = _random_pose2_graph(rng, nv, seed % 2 == 0)
    arr = g.to_arrays(v)
    P, F = gpu.product_backend(arr), gpu.product_backend(arr)
    kind = [A.ORDER_ND, A.ORDER_MINDEGREE][seed % 2]
    ordering = P.compute_ordering(kind)
    for be in (P, F):
        if seed % 3 == 0:
            be.set_amalgamation(0.0, 128)
        be.set_ordering(ordering)
    P.linearize()
    P.solve(0.0, False)
    off = np.concatenate([[0], np.cumsum(arr.state_dims())])
    current = arr.values.copy()
    for round_ in range(3):
        idx = np.sort(rng.choice(nv, size=max(1, int(nv * rng.choice([0.02, 0.1]))), replace=False))
        states = []
        for i in idx:
            current[off[i]:off[i + 1]] += rng.normal(0, 0.02, 3)
            states.append(current[off[i]:off[i + 1]])
        P.relinearize_partial(arr.var_keys[idx], np.concatenate(states))
        dp = P.solve(0.0, False) if round_ % 2 == 0 else P.backsubstitute_wildfire(0.0)[0]
        F.set_values(current)
        F.linearize()
        df = F.solve(0.0, False)
        assert np.array_equal(P.jacobians(), F.jacobians()), (seed, round_)
        assert np.array_equal(dp, df), (seed, round_, float(np.max(np.abs(dp - df))))


# ---- a resident factorization never outlives the tree / linearization it was computed for ----------------------------------
@pytest.mark.parametrize("name", ["bal_small", "pose3"])
def test_reordering_a_live_handle_drops_the_resident_factorization(gpu, oracle, name):
    """solve(lambda = 0) / marginals under ordering A leave the undamped factorization resident; gsx_set_ordering(B)
    (with or without gsx_set_amalgamation) re-allocates the arena for another tree — the fast paths (solve at lambda = 0,
    marginals) must re-factor, not back-substitute through the new arena laid out for the old tree."""
    arr = PROBLEMS[name]
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    kinds = (A.ORDER_SCHUR_ND, A.ORDER_SCHUR) if name.startswith("bal") else (A.ORDER_ND, A.ORDER_MINDEGREE)
    oa, obb = gb.compute_ordering(kinds[0]), gb.compute_ordering(kinds[1])
    gb.set_ordering(oa)
    ob.set_ordering(oa)
    gb.linearize()
    ob.linearize()
    assert relerr(gb.solve(0.0, False), ob.solve(0.0, False)) < 1e-8
    key = oa[len(oa) // 2]
    co = ob.marginal_covariance(key)
    assert np.max(np.abs(gb.marginal_covariance(key) - co)) <= 1e-7 * np.max(np.abs(co))
    # new ordering, NO re-linearization: lambda = 0 again
    gb.set_ordering(obb)
    ob.set_ordering(obb)
    ob.linearize()
    assert relerr(gb.solve(0.0, False), ob.solve(0.0, False)) < 1e-8
    assert np.max(np.abs(gb.marginal_covariance(key) - co)) <= 1e-7 * np.max(np.abs(co))
    # ... and a change of amalgamation followed by the first ordering again, marginals first
    gb.set_amalgamation(0.5, 64)
    gb.set_ordering(oa)
    assert np.max(np.abs(gb.marginal_covariance(key) - co)) <= 1e-7 * np.max(np.abs(co))
    ob.set_ordering(oa)
    assert relerr(gb.solve(0.0, False), ob.solve(0.0, False)) < 1e-8


def test_failed_factorization_is_not_reused(gpu):
    """An indeterminate Gauss-Newton run (lambda = 0 on a gauge-free graph) leaves a FAILED factorization in the arena:
    a marginal query afterwards must report the failure again, not read the wreck; once a prior anchors the gauge the same
    query works."""
    fg = NonlinearFactorGraph()
    for i in range(5):
        fg.add(BetweenFactor(i, i + 1, Pose2(1, 0, 0.1), noiseModel.Isotropic.Sigma(3, 0.1)))
    v = Values()
    for i in range(6):
        v.insert(i, Pose2(1.0 * i + 0.05 * i, 0.02 * i, 0.1 * i))
    be = gpu.product_backend(fg.to_arrays(v))
    be.set_ordering(list(range(6)))
    with pytest.raises(gt.IndeterminantLinearSystemException):
        be.gn_optimize(max_iterations=3)
    with pytest.raises(gt.IndeterminantLinearSystemException):
        be.marginal_covariance(2)
    with pytest.raises(gt.IndeterminantLinearSystemException):
        be.solve(0.0, False)
    # damped: fine; then lambda = 0 again must re-factor (and fail again), not trust the damped factorization
    be.linearize()
    assert np.all(np.isfinite(be.solve(1e-3, False)))
    with pytest.raises(gt.IndeterminantLinearSystemException):
        be.solve(0.0, False)


# ---- round 3: the dependency-driven launches against the level launches they replace ------------------------------------------
@pytest.mark.parametrize("name,maker", [
    ("pose3_1500", lambda: datasets.synth_manhattan_pose3(1500, seed=9)),
    ("pose2_5000", lambda: datasets.synth_manhattan_pose2(5000, seed=9)),
    ("pose3_small", lambda: PROBLEMS["pose3"]),
    ("pose2_small", lambda: PROBLEMS["pose2"]),
])
def test_tree_kernels_against_the_level_launches(gpu, name, maker, monkeypatch):
    """front_tree_kernel (a workgroup climbs from a front to its parent when it was the last child to arrive) and
    backsolve_tree_kernel (ticket list, top-down) run the same front bodies as the level-by-level launches, whatever
    workgroup arrives when: the damped step is the same BITS from run to run, and equals the step with the tree kernels
    switched off (GSX_TREE_TIERS=0: no tree fronts, every LDS front in its level's launch) to rounding — the two
    schedules give a front other workgroup sizes, so the panel updates sum in other shapes (measured: not bit-equal)."""
    arr = maker()
    steps = {}
    for mode in ("tree", "levels"):
        if mode == "levels":
            monkeypatch.setenv("GSX_TREE_TIERS", "0")
        be = gpu.product_backend(arr)
        be.set_amalgamation(0.5, 32)
        be.set_ordering(be.compute_ordering(A.ORDER_ND))
        st = be.stats()
        assert (st["n_tree_fronts"] > 0) == (mode == "tree"), (name, mode, st["n_tree_fronts"])
        # (graphs without blocked fronts: with them, switching the tree fronts off also moves LDS fronts into the blocked
        #  launches of their level — other kernels, other rounding)
        assert st["n_big_fronts"] == 0
        be.linearize()
        d1 = be.solve(1e-3, False)
        d2 = be.solve(1e-3, False)
        assert np.array_equal(d1, d2)
        steps[mode] = (d1, be.solve(1e-2, True), be.hessian_diagonal())
        be.close()
    for a, b in zip(steps["tree"], steps["levels"]):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-11 * np.abs(b).max()), (name, np.abs(a - b).max())
