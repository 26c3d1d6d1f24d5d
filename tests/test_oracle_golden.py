"""Pins the CPU oracle (oracle/) against the reference's own known-answer tests.

Every case below is taken from a test in /root/reference (cited) — inputs and
expected outputs are data, restated here as fixtures; nothing under
/root/reference is read at run time.  These run with -m "not gpu".
"""
import math
import os

import numpy as np
import pytest

import gtsam_petercdev_amd as gt
from gtsam_petercdev_amd import _abi as A
from gtsam_petercdev_amd import datasets
from gtsam_petercdev_amd.graph import (X, L, P, Pose2, Values, NonlinearFactorGraph, GaussianFactorGraph,
                                       JacobianFactor, PriorFactor, BetweenFactor, noiseModel, Ordering,
                                       LevenbergMarquardtOptimizer, LevenbergMarquardtParams,
                                       GaussNewtonOptimizer, Point2)


@pytest.fixture(scope="module")
def orc(oracle):
    return oracle


# ---- gtsam/base/tests/testCholesky.cpp:26-138 ------------------------------------------------
ABC7 = np.array([
    [4.0375, 3.4584, 3.5735, 2.4815, 2.1471, 2.7400, 2.2063],
    [0., 4.7267, 3.8423, 2.3624, 2.8091, 2.9579, 2.5914],
    [0., 0., 5.1600, 2.0797, 3.4690, 3.2419, 2.9992],
    [0., 0., 0., 1.8786, 1.0535, 1.4250, 1.3347],
    [0., 0., 0., 0., 3.0788, 2.6283, 2.3791],
    [0., 0., 0., 0., 0., 2.9227, 2.4056],
    [0., 0., 0., 0., 0., 0., 2.5776]])


def test_choleskyPartial0(orc):
    abc = ABC7[:3, :3].copy()
    rsl, ok = orc.cholesky_partial(abc, 0)
    assert ok and np.allclose(rsl, abc, atol=1e-9)


def test_choleskyPartial(orc):
    rsl, ok = orc.cholesky_partial(ABC7, 3)
    assert ok
    R1 = rsl.T.copy()
    R2 = rsl.copy()
    R1[3:, 3:] = np.eye(4)
    blk = np.triu(R2[3:, 3:])
    R2[3:, 3:] = blk + blk.T - np.diag(np.diagonal(blk))
    # only the upper triangle of rsl is meaningful: R1 must be lower, R2 upper in the frontal rows
    R1 = np.tril(R1[:, :3]).tolist()
    R1 = np.concatenate([np.array(R1), np.concatenate([np.zeros((3, 4)), np.eye(4)], axis=0)], axis=1)
    R2f = np.concatenate([np.triu(rsl)[:3, :], np.concatenate([np.zeros((4, 3)), R2[3:, 3:]], axis=1)], axis=0)
    actual = R1 @ R2f
    expected = np.triu(ABC7) + np.triu(ABC7, 1).T
    assert np.allclose(actual, expected, atol=1e-9)


def test_cholesky_BadScaling(orc):
    Am = np.array([[1e-40, 0.0], [0.0, 1.0]])
    R, _ = orc.cholesky_partial(Am.T @ Am, 2)
    assert abs(R[0, 0] / R[1, 1] - 1e-40) < 1e-41


def test_cholesky_underconstrained(orc):
    Lm = np.array([
        [1, 0, 0, 0, 0, 0],
        [1.11177808157954, 1.06204809504665, 0.507342638873381, 1.34953401829486, 1, 0],
        [0.155864888199928, 1.10933048588373, 0.501255576961674, 1, 0, 0],
        [1.12108665967793, 1.01584408366945, 1, 0, 0, 0],
        [0.776164062474843, 0.117617236580373, -0.0236628691347294, 0.814118199972143, 0.694309975328922, 1],
        [0.1197220685104, 1, 0, 0, 0, 0]])
    d = [0.814723686393179, 0.811780089277421, 1.82596950680844, 0.240287537694585]
    for tail in ([1.34342584865901, 1e-12], [0, 0], [-0.5, -0.6]):
        Am = Lm @ np.diag(d + tail) @ Lm.T
        _, ok = orc.cholesky_partial(Am, 6)
        assert not ok


# ---- tests/smallExample.h:270-290 + tests/testGaussianJunctionTreeB.cpp:127-140 --------------------
def small_gaussian_factor_graph():
    I = np.eye(2)
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(X(1), 10 * I, -1.0 * np.ones(2)))
    fg.add(JacobianFactor(X(1), -10 * I, X(2), 10 * I, [2.0, -1.0]))
    fg.add(JacobianFactor(X(1), -5 * I, L(1), 5 * I, [0.0, 1.0]))
    fg.add(JacobianFactor(X(2), -5 * I, L(1), 5 * I, [-1.0, 1.5]))
    return fg


CORRECT_DELTA = {L(1): [-0.1, 0.1], X(1): [-0.1, -0.1], X(2): [0.1, -0.2]}


@pytest.mark.parametrize("ordering", [[L(1), X(1), X(2)], [X(2), L(1), X(1)], [X(1), X(2), L(1)]])
def test_optimizeMultiFrontal2(orc, ordering):
    actual = small_gaussian_factor_graph().optimize(ordering, backend_factory=orc.oracle_backend)
    for k, v in CORRECT_DELTA.items():
        assert np.allclose(actual[k], v, atol=1e-9)


# ---- gtsam/linear/tests/testGaussianBayesTree.cpp -----------------------------------------------------
def chain_graph():
    """:36-47 — x1 - x2 - x3 - x4 with a prior on x4, Isotropic sigma 0.5."""
    noise = noiseModel.Isotropic.Sigma(1, 0.5)
    one = np.eye(1)
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(2, one, 1, one, [1.0], noise))
    fg.add(JacobianFactor(2, one, 3, one, [1.0], noise))
    fg.add(JacobianFactor(3, one, 4, one, [1.0], noise))
    fg.add(JacobianFactor(4, one, [1.0], noise))
    return fg


def _rows_equal_up_to_sign(a, b, tol=1e-9):
    """GaussianConditional::equals accepts a conditional whose rows differ by a sign (QR vs Cholesky)."""
    return all(np.allclose(x, y, atol=tol) or np.allclose(x, -y, atol=tol) for x, y in zip(a, b))


def test_GaussianBayesTree_eliminate(orc):
    """:84-117 — Bayes tree (x3 x4) <- (x2 x1 : x3) with the expected [R S d] of both cliques."""
    arrays = chain_graph().to_arrays(None)
    arrays.values = np.zeros(4)
    be = orc.oracle_backend(arrays)
    be.set_ordering([2, 1, 3, 4])
    be.linearize()
    be.solve(0.0)
    parent, fronts = be.get_tree()
    keys = arrays.var_keys.tolist()
    cl = {tuple(keys[i] for i in f): c for c, (f, s) in enumerate(fronts)}
    assert set(cl) == {(3, 4), (2, 1)}
    assert parent[cl[(3, 4)]] == -1 and parent[cl[(2, 1)]] == cl[(3, 4)]
    assert tuple(keys[i] for i in fronts[cl[(2, 1)]][1]) == (3,)
    s2 = math.sqrt(2.0)
    gc1 = np.array([[2.0, 2.0, 2.0], [0.0, 2.0, 2.0]])                      # [R(x3 x4) | d]
    gc2 = np.array([[-2 * s2, -s2, -s2, -2 * s2], [0.0, -s2, s2, 0.0]])     # [R(x2 x1) | S(x3) | d]
    assert _rows_equal_up_to_sign(be.conditional(cl[(3, 4)]), gc1)
    assert _rows_equal_up_to_sign(be.conditional(cl[(2, 1)]), gc2)


def test_GaussianBayesTree_optimizeMultiFrontal(orc):
    """:120-129"""
    actual = chain_graph().optimize([2, 1, 3, 4], backend_factory=orc.oracle_backend)
    for k, v in {1: 0.0, 2: 1.0, 3: 0.0, 4: 1.0}.items():
        assert np.allclose(actual[k], [v], atol=1e-9)


# ---- gtsam/linear/tests/testHessianFactor.cpp -----------------------------------------------------------
def test_HessianFactor_eliminate2(orc):
    """:377-444 — EliminateCholesky of one combined factor on (x2 | l1 x1): expected R11, S12, d (1e-4)."""
    sigmas = [0.2, 0.2, 0.1, 0.1]
    Ax2 = np.array([[-1., 0.], [0., -1.], [1., 0.], [0., 1.]])
    Al1x1 = np.array([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., -1., 0.], [0., 0., 0., -1.]])
    b2 = [-0.2, 0.3, 0.2, -0.1]
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(0, Ax2, 1, Al1x1, b2, noiseModel.Diagonal.Sigmas(sigmas)))
    # the test eliminates x2 only; a prior on (l1 x1) makes the full system solvable and does not touch the
    # rows of the conditional on x2 (first two rows of the factor of the merged clique)
    fg.add(JacobianFactor(1, np.eye(4), np.zeros(4)))
    arrays = fg.to_arrays(None)
    arrays.values = np.zeros(6)
    be = orc.oracle_backend(arrays)
    be.set_ordering([0, 1])
    be.linearize()
    be.solve(0.0)
    parent, fronts = be.get_tree()
    assert len(fronts) == 1  # {x2} merges into its parent (JunctionTree-inst.h:120-149)
    cond = be.conditional(0)
    old_sigma = 0.0894427
    expected = np.array([[1.0, 0.0, -0.2, 0.0, -0.8, 0.0, 0.2], [0.0, 1.0, 0.0, -0.2, 0.0, -0.8, -0.14]]) / old_sigma
    assert _rows_equal_up_to_sign(cond[:2], expected, tol=1e-4 / old_sigma * 0.1)


def test_HessianFactor_combine_hessianDiagonal(orc):
    """:447-477, :527-549 — information of a 3-key factor: diagonal of the expected 7x7 matrix."""
    A0 = 11.1803399 * np.eye(2)
    A1 = -2.23606798 * np.eye(2)
    A2 = -8.94427191 * np.eye(2)
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(0, A0, 1, A1, 2, A2, [2.23606798, -1.56524758], noiseModel.Diagonal.Sigmas([1.0, 1.0])))
    arrays = fg.to_arrays(None)
    arrays.values = np.zeros(6)
    be = orc.oracle_backend(arrays)
    be.linearize()
    assert np.allclose(be.hessian_diagonal(), [125.0, 125.0, 5.0, 5.0, 80.0, 80.0], atol=1e-4)


def nonlinear_smoother(T):
    """tests/smallExample.h:434-462 with simulated2D Prior/Odometry == Prior/Between on Point2."""
    g, v = NonlinearFactorGraph(), Values()
    unit = noiseModel.Isotropic.Sigma(2, 1.0)
    g.add(PriorFactor(X(1), Point2(1.0, 0.0), unit))
    v.insert(X(1), Point2(1.0, 0.0))
    for t in range(2, T + 1):
        g.add(BetweenFactor(X(t - 1), X(t), Point2(1.0, 0.0), unit))
        g.add(PriorFactor(X(t), Point2(t, 0), unit))
        v.insert(X(t), Point2(t, 0))
    return g, v


def test_smoother_junction_tree_constructor2(orc):
    """tests/testGaussianJunctionTreeB.cpp:66-111: cliques (x3 x2 x4) <- {(x1), (x5 x6) <- (x7)}."""
    g, v = nonlinear_smoother(7)
    arrays = g.to_arrays(v)
    be = orc.oracle_backend(arrays)
    be.set_ordering([X(1), X(3), X(5), X(7), X(2), X(6), X(4)])
    be.linearize()
    delta = be.solve(0.0)
    parent, fronts = be.get_tree()
    keys = arrays.var_keys.tolist()
    cl = {tuple(keys[i] for i in f): (parent[c], tuple(keys[i] for i in s)) for c, (f, s) in enumerate(fronts)}
    assert set(cl) == {(X(3), X(2), X(4)), (X(5), X(6)), (X(7),), (X(1),)}
    root = [c for c, (f, s) in enumerate(fronts) if tuple(keys[i] for i in f) == (X(3), X(2), X(4))][0]
    assert parent[root] == -1
    idx = {tuple(keys[i] for i in f): c for c, (f, s) in enumerate(fronts)}
    assert parent[idx[(X(1),)]] == root and parent[idx[(X(5), X(6))]] == root
    assert parent[idx[(X(7),)]] == idx[(X(5), X(6))]
    # OptimizeMultiFrontal (:113-125): the smoother is at its optimum -> delta = 0
    assert np.allclose(delta, 0.0, atol=1e-9)


# ---- tests/testNonlinearOptimizer.cpp -----------------------------------------------------------------
def test_Factorization(orc):
    """:185-208 — one LM iteration on a 2-pose graph."""
    config = Values()
    config.insert(X(1), Pose2(0., 0., 0.))
    config.insert(X(2), Pose2(1.5, 0., 0.))
    graph = NonlinearFactorGraph()
    graph.addPrior(X(1), Pose2(0., 0., 0.), noiseModel.Isotropic.Sigma(3, 1e-10))
    graph.add(BetweenFactor(X(1), X(2), Pose2(1., 0., 0.), noiseModel.Isotropic.Sigma(3, 1)))
    ordering = Ordering([X(1), X(2)])
    opt = LevenbergMarquardtOptimizer(graph, config, ordering, LevenbergMarquardtParams.LegacyDefaults(),
                                      backend_factory=orc.oracle_backend)
    opt.iterate()
    res = opt.values()
    assert res.at(X(1)).equals(Pose2(0., 0., 0.), 1e-5)
    assert res.at(X(2)).equals(Pose2(1., 0., 0.), 1e-5)


def more_optimization_graph():
    fg = NonlinearFactorGraph()
    fg.addPrior(0, Pose2(0, 0, 0), noiseModel.Isotropic.Sigma(3, 1))
    fg.add(BetweenFactor(0, 1, Pose2(1, 0, math.pi / 2), noiseModel.Isotropic.Sigma(3, 1)))
    fg.add(BetweenFactor(1, 2, Pose2(1, 0, math.pi / 2), noiseModel.Isotropic.Sigma(3, 1)))
    return fg


def test_MoreOptimization(orc):
    """:248-321 — 3-pose LM converges to the exact poses; gradient at the optimum is zero."""
    fg = more_optimization_graph()
    init = Values()
    init.insert(0, Pose2(3, 4, -math.pi))
    init.insert(1, Pose2(10, 2, -math.pi))
    init.insert(2, Pose2(11, 7, -math.pi))
    expected = {0: Pose2(0, 0, 0), 1: Pose2(1, 0, math.pi / 2), 2: Pose2(1, 1, math.pi)}
    opt = LevenbergMarquardtOptimizer(fg, init, Ordering([0, 1, 2]), LevenbergMarquardtParams.LegacyDefaults(),
                                      backend_factory=orc.oracle_backend)
    actual = opt.optimize()
    for k, e in expected.items():
        assert actual.at(k).equals(e, 1e-6)
    # gradientAtZero = sum A'b = 0
    be = opt.backend
    be.linearize()
    jac = be.jacobians()
    off = opt.arrays.jacobian_offsets()
    grad = np.zeros(9)
    toff = opt.arrays.tangent_offsets()
    for f in range(opt.arrays.n_factors):
        vs = opt.arrays.f_vars[opt.arrays.f_key_ptr[f]:opt.arrays.f_key_ptr[f + 1]]
        Ab = jac[off[f]:off[f + 1]].reshape(-1, 3).T if False else jac[off[f]:off[f + 1]].reshape((3, -1), order="F")
        b = Ab[:, -1]
        for s, vi in enumerate(vs):
            grad[toff[vi]:toff[vi] + 3] += Ab[:, 3 * s:3 * s + 3].T @ b
    assert np.allclose(grad, 0.0, atol=1e-6)


def test_MoreOptimization_diagonal_damping(orc):
    """:283-300 — damped.hessianDiagonal() == d + lambda*d for diagonalDamping."""
    fg = more_optimization_graph()
    initBetter = Values()
    initBetter.insert(0, Pose2(3, 4, 0))
    initBetter.insert(1, Pose2(10, 2, math.pi / 3))
    initBetter.insert(2, Pose2(11, 7, math.pi / 2))
    arrays = fg.to_arrays(initBetter)
    be = orc.oracle_backend(arrays)
    be.set_ordering([0, 1, 2])
    be.linearize()
    d = be.hessian_diagonal()
    lam = 1e-5
    # Build the damped system explicitly as the reference does and compare its Hessian diagonal:
    # solving with (H + lam*diag(d)) must equal solving the explicit damped normal equations.
    jac = be.jacobians()
    off = arrays.jacobian_offsets()
    toff = arrays.tangent_offsets()
    H = np.zeros((9, 9))
    g = np.zeros(9)
    for f in range(arrays.n_factors):
        vs = arrays.f_vars[arrays.f_key_ptr[f]:arrays.f_key_ptr[f + 1]]
        Ab = jac[off[f]:off[f + 1]].reshape((3, -1), order="F")
        J = np.zeros((3, 9))
        for s, vi in enumerate(vs):
            J[:, toff[vi]:toff[vi] + 3] = Ab[:, 3 * s:3 * s + 3]
        H += J.T @ J
        g += J.T @ Ab[:, -1]
    assert np.allclose(np.diagonal(H), d, rtol=1e-12)
    expected = np.linalg.solve(H + lam * np.diag(d), g)
    delta = be.solve(lam, diagonal_damping=True, min_diagonal=0.0, max_diagonal=1e300)
    assert np.allclose(delta, expected, rtol=1e-8, atol=1e-10)


def disconnected_graph():
    """tests/testNonlinearOptimizer.cpp:485-503 — two components: (x1 - x2) and x3 alone."""
    graph = NonlinearFactorGraph()
    graph.addPrior(X(1), Pose2(0., 0., 0.), noiseModel.Isotropic.Sigma(3, 1))
    graph.add(BetweenFactor(X(1), X(2), Pose2(1.5, 0., 0.), noiseModel.Isotropic.Sigma(3, 1)))
    graph.addPrior(X(3), Pose2(3., 0., 0.), noiseModel.Isotropic.Sigma(3, 1))
    init = Values()
    for k in (1, 2, 3):
        init.insert(X(k), Pose2(0., 0., 0.))
    expected = {X(1): Pose2(0., 0., 0.), X(2): Pose2(1.5, 0., 0.), X(3): Pose2(3.0, 0., 0.)}
    return graph, init, expected


def test_disconnected_graph(orc):
    """:485-503 — a forest (two Bayes-tree roots) optimizes like any other graph."""
    graph, init, expected = disconnected_graph()
    opt = LevenbergMarquardtOptimizer(graph, init, Ordering([X(1), X(2), X(3)]), LevenbergMarquardtParams(),
                                      backend_factory=orc.oracle_backend)
    actual = opt.optimize()
    for k, e in expected.items():
        assert actual.at(k).equals(e, 1e-9)


def test_lm_lambda0_equals_gauss_newton(orc):
    """:60-82 — with lambda = 0 one LM iteration equals one Gauss-Newton iteration."""
    fg = more_optimization_graph()
    init = Values()
    init.insert(0, Pose2(0.2, -0.1, 0.1))
    init.insert(1, Pose2(1.3, 0.2, 1.3))
    init.insert(2, Pose2(0.8, 1.2, 3.0))
    p = LevenbergMarquardtParams.LegacyDefaults()
    p.lambdaInitial = 0.0
    lm = LevenbergMarquardtOptimizer(fg, init, Ordering([0, 1, 2]), p, backend_factory=orc.oracle_backend)
    lm.iterate()
    gn = GaussNewtonOptimizer(fg, init, Ordering([0, 1, 2]), maxIterations=1, backend_factory=orc.oracle_backend)
    gn.optimize()
    for k in (0, 1, 2):
        assert lm.values().at(k).equals(gn.values().at(k), 1e-9)


def test_indeterminate_system_is_reported(orc):
    """A graph without a prior is gauge-free: EliminateCholesky must fail like the reference
    (HessianFactor.cpp:475-482 -> IndeterminantLinearSystemException)."""
    fg = NonlinearFactorGraph()
    fg.add(BetweenFactor(0, 1, Pose2(1, 0, 0), noiseModel.Isotropic.Sigma(3, 1)))
    v = Values()
    v.insert(0, Pose2(0, 0, 0))
    v.insert(1, Pose2(1, 0, 0))
    be = orc.oracle_backend(fg.to_arrays(v))
    be.set_ordering([0, 1])
    be.linearize()
    with pytest.raises(gt.IndeterminantLinearSystemException):
        be.solve(0.0)


# ---- gtsam/slam/tests/testGeneralSFMFactor_Cal3Bundler.cpp:100-113 -----------------------------------------
def test_sfm_unwhitened_error(orc):
    """camera at (0,0,-6) looking down +z, point at origin, z=(3,0): error = h(x)-z = (-3, 0)."""
    from gtsam_petercdev_amd.graph import (GeneralSFMFactor, Pose3, Rot3, Cal3Bundler,
                                           PinholeCameraCal3Bundler, Point3)
    g, v = NonlinearFactorGraph(), Values()
    g.add(GeneralSFMFactor(Point2(3., 0.), noiseModel.Unit.Create(2), X(1), L(1)))
    v.insert(X(1), PinholeCameraCal3Bundler(Pose3(Rot3(), Point3(0, 0, -6)), Cal3Bundler(1.0, 0.0, 0.0)))
    v.insert(L(1), Point3(0, 0, 0))
    be = orc.oracle_backend(g.to_arrays(v))
    be.linearize()
    jac = be.jacobians().reshape((2, 13), order="F")
    assert np.allclose(jac[:, -1], [3.0, 0.0])  # b = z - h(x) = -(error)
    assert abs(be.error() - 4.5) < 1e-12


# ---- tests/testGeneralSFMFactorB.cpp:44-63 ---------------------------------------------------------------------
def test_PinholeCamera_BAL(orc, golden_dir):
    """dubrovnik-3-7-pre, unit noise, default LM, reference COLAMD ordering: final error 0.0199833."""
    sfm = datasets.read_bal(golden_dir + "/dubrovnik-3-7-pre.txt")
    assert (sfm.numberCameras(), sfm.numberTracks(), sfm.cam_idx.size) == (3, 7, 19)
    arrays = datasets.bal_arrays(sfm, priors=False)
    be = orc.oracle_backend(arrays)
    if orc.have_ref_colamd():
        ordering = orc.colamd_ordering(arrays)
    else:  # the golden ordering the reference's CCOLAMD produces for this graph (generated by the line above)
        ordering = np.load(golden_dir + "/dubrovnik_colamd_ordering.npy")
    be.set_ordering(ordering)
    res = be.lm_optimize(A.lm_params_legacy())
    assert abs(res["final_error"] - 0.0199833) < 1e-5
    assert abs(be.error() - 0.0199833) < 1e-5


# ---- gtsam/inference/tests/testOrdering.cpp:40-107 (the reference's own CCOLAMD, built from its C source) ----
def _chain():
    fg = GaussianFactorGraph()
    for i in range(5):
        fg.add(JacobianFactor(i, np.eye(1), i + 1, np.eye(1), [0.0]))
    return fg.to_arrays(None)


def test_colamd_chain(orc):
    if not orc.have_ref_colamd():
        pytest.skip("oracle/_ref/libccolamd_ref.so not built (reference tree absent)")
    arrays = _chain()
    assert orc.colamd_ordering(arrays).tolist() == [0, 1, 2, 3, 4, 5]
    # ColamdConstrainedLast({2,4}) -> cmember 1 for those (Ordering.cpp:128-151)
    cm = np.zeros(6, np.int32)
    cm[[2, 4]] = 1
    assert orc.colamd_ordering(arrays, cm).tolist() == [0, 1, 5, 3, 4, 2]
    # ColamdConstrainedFirst({2,4}) -> group 0 for those, 1 for the rest (Ordering.cpp:154-183)
    cm = np.ones(6, np.int32)
    cm[[2, 4]] = 0
    assert orc.colamd_ordering(arrays, cm).tolist() == [2, 4, 0, 1, 3, 5]
    # grouped: {2:1, 4:1, 5:2} (testOrdering.cpp:89-103)
    cm = np.zeros(6, np.int32)
    cm[[2, 4]] = 1
    cm[5] = 2
    assert orc.colamd_ordering(arrays, cm).tolist() == [0, 1, 3, 2, 4, 5]


# ---- gtsam/inference/tests/testOrdering.cpp:240-393 (the reference's own METIS 5, built from its C sources) ----
def _symbolic(factors):
    """A SymbolicFactorGraph as a linear graph of 1-dim variables: only the key lists matter."""
    fg = GaussianFactorGraph()
    for keys in factors:
        args = []
        for k in keys:
            args += [k, np.eye(1)]
        fg.add(JacobianFactor(*args, [0.0]))
    return fg.to_arrays(None)


def test_MetisIndex_csr_format_4(orc):
    """:240-275 — CSR of the x1..x6 chain with a unary factor on x1, then a landmark seen from x1..x4."""
    chain = [[X(1)], [X(1), X(2)], [X(2), X(3)], [X(3), X(4)], [X(4), X(5)], [X(5), X(6)]]
    xadj, adj, _ = orc.metis_index(_symbolic(chain))
    assert xadj.tolist() == [0, 1, 3, 5, 7, 9, 10]
    assert adj.tolist() == [1, 0, 2, 1, 3, 2, 4, 3, 5, 4]
    more = chain + [[L(1)], [X(1), L(1)], [X(2), L(1)], [X(3), L(1)], [X(4), L(1)]]
    xadj, adj, i2v = orc.metis_index(_symbolic(more))
    # the landmark is vertex 6 (first seen last) although its key sorts before the x's
    assert xadj.tolist() == [0, 2, 5, 8, 11, 13, 14, 18]
    assert adj[-4:].tolist() == [0, 1, 2, 3]
    if orc.have_ref_metis():
        assert sorted(orc.metis_ordering(_symbolic(more)).tolist()) == sorted([L(1)] + [X(i) for i in range(1, 7)])


def test_MetisIndex_metis(orc):
    """:278-297"""
    xadj, adj, _ = orc.metis_index(_symbolic([[0], [0, 1], [1, 2]]))
    assert xadj.tolist() == [0, 1, 3, 4]
    assert adj.tolist() == [1, 0, 2, 1]


def test_metis_chain_and_loop(orc):
    """:300-336 (the Linux branch) and :375-383 — METIS_NodeND of the reference's own METIS."""
    if not orc.have_ref_metis():
        pytest.skip("oracle/_ref/libmetis_ref.so not built (reference tree absent)")
    chain = [[i, i + 1] for i in range(5)]
    assert orc.metis_ordering(_symbolic(chain)).tolist() == [5, 3, 4, 1, 0, 2]
    assert orc.metis_ordering(_symbolic(chain + [[0, 5]])).tolist() == [3, 2, 5, 0, 4, 1]


def test_metis_empty_and_single(orc):
    """:339-360 — an empty graph gives an empty ordering, a single node itself (no METIS call)."""
    assert orc.metis_ordering(_symbolic([[7]])).tolist() == [7]
    assert orc.metis_ordering(GaussianFactorGraph().to_arrays(None)).size == 0


# ---- tests/testDoglegOptimizer.cpp ---------------------------------------------------------------------
def test_Dogleg_ComputeBlendEdgeCases(orc):
    """:76-92 (issue #1861) — a trust region equal to |n| returns n, equal to |u| returns u."""
    n = np.array([0.3233546123, -0.2133456123, 0.3664345632])
    u = np.array([0.0023456342, -0.04535687, 0.087345661212])
    assert np.allclose(orc.dogleg_point(np.linalg.norm(n), u, n, blend_only=True), n, atol=1e-9)
    assert np.allclose(orc.dogleg_point(np.linalg.norm(u), u, n, blend_only=True), u, atol=1e-9)


def _dogleg_bayes_net_points(orc):
    """The Bayes net of :42-62 / :97-113 as a linear graph of unit-noise rows [R S | d]: its steepest-descent point
    (GaussianFactorGraph::optimizeGradientSearch) and Newton point, in key order 0..4 (2 dims each)."""
    rows = [  # (keys, blocks, d)
        ((0, 3, 4), ([[3, 4], [0, 6]], [[7, 8], [9, 10]], [[11, 12], [13, 14]]), (1, 2)),
        ((1, 2, 4), ([[17, 18], [0, 20]], [[21, 22], [23, 24]], [[25, 26], [27, 28]]), (15, 16)),
        ((2, 3), ([[31, 32], [0, 34]], [[35, 36], [37, 38]]), (29, 30)),
        ((3, 4), ([[41, 42], [0, 44]], [[45, 46], [47, 48]]), (39, 40)),
        ((4,), ([[51, 52], [0, 54]],), (49, 50)),
    ]
    A_ = np.zeros((10, 10))
    b = np.zeros(10)
    for i, (keys, blocks, d) in enumerate(rows):
        for k, blk in zip(keys, blocks):
            A_[2 * i:2 * i + 2, 2 * k:2 * k + 2] = blk
        b[2 * i:2 * i + 2] = d
    xn = np.linalg.solve(A_, b)
    grad = -A_.T @ b
    xu = -(grad @ grad) / np.sum((A_ @ grad) ** 2) * grad
    return xu, xn


def test_Dogleg_ComputeBlend_and_DoglegPoint(orc):
    """:40-73, :95-130 — the blend has norm Delta; the dog-leg point is the scaled steepest-descent point, the blend,
    or the Newton point depending on Delta."""
    xu, xn = _dogleg_bayes_net_points(orc)
    assert np.linalg.norm(xu) < np.linalg.norm(xn)
    assert abs(np.linalg.norm(orc.dogleg_point(1.5, xu, xn, blend_only=True)) - 1.5) < 1e-10
    assert abs(np.linalg.norm(orc.dogleg_point(0.5, xu, xn)) - 0.5) < 1e-5
    assert np.allclose(orc.dogleg_point(1.5, xu, xn), orc.dogleg_point(1.5, xu, xn, blend_only=True), atol=1e-12)
    assert np.allclose(orc.dogleg_point(5.0, xu, xn), xn, atol=1e-12)


def test_Dogleg_converges_on_the_pose2_examples(orc):
    """Dogleg on the graphs of tests/testNonlinearOptimizer.cpp:248-321 and :485-503 reaches the same optimum as LM."""
    from gtsam_petercdev_amd.graph import DoglegOptimizer
    fg = more_optimization_graph()
    init = Values()
    init.insert(0, Pose2(3, 4, -math.pi))
    init.insert(1, Pose2(10, 2, -math.pi))
    init.insert(2, Pose2(11, 7, -math.pi))
    opt = DoglegOptimizer(fg, init, Ordering([0, 1, 2]), backend_factory=orc.oracle_backend)
    actual = opt.optimize()
    for k, e in {0: Pose2(0, 0, 0), 1: Pose2(1, 0, math.pi / 2), 2: Pose2(1, 1, math.pi)}.items():
        assert actual.at(k).equals(e, 1e-5)
    graph, init, expected = disconnected_graph()
    actual = DoglegOptimizer(graph, init, Ordering([X(1), X(2), X(3)]), backend_factory=orc.oracle_backend).optimize()
    for k, e in expected.items():
        assert actual.at(k).equals(e, 1e-6)


# ---- robust noise models: gtsam/linear/tests/testNoiseModel.cpp, tests/testNonlinearOptimizer.cpp ------------------
ROBUST_FUNCTIONS = [  # (mEstimator, k, {error: (weight, loss)}) — testNoiseModel.cpp:476-569
    ("Huber", 5.0, {1.0: (1.0, 0.5), 10.0: (0.5, 37.5), -10.0: (0.5, 37.5), -1.0: (1.0, 0.5)}),
    ("Cauchy", 5.0, {1.0: (0.961538461538461, 0.490258914416017), 10.0: (0.2, 20.117973905426254),
                     -10.0: (0.2, 20.117973905426254), -1.0: (0.961538461538461, 0.490258914416017)}),
    ("Tukey", 5.0, {1.0: (0.9216, 0.480266666666667), 10.0: (0.0, 4.166666666666667),
                    -10.0: (0.0, 4.166666666666667), -1.0: (0.9216, 0.480266666666667)}),
]


def robust_function_check(backend_factory, name, k, table):
    """A 1-D prior with Robust(mEstimator(k), Unit): error() is loss(e), the linearized [A b] is sqrt(weight(e)) [1 -e]."""
    for e, (weight, loss) in table.items():
        g = NonlinearFactorGraph()
        m = getattr(noiseModel.mEstimator, name).Create(k)
        g.add(PriorFactor(0, np.array([0.0]), noiseModel.Robust.Create(m, noiseModel.Unit.Create(1))))
        v = Values()
        v.insert(0, np.array([e]))
        be = backend_factory(g.to_arrays(v))
        assert abs(be.error() - loss) < 1e-8, (name, e)
        be.linearize()
        J = be.jacobians()
        assert abs(J[0] - math.sqrt(weight)) < 1e-8 and abs(J[1] + math.sqrt(weight) * e) < 1e-8, (name, e, J)


@pytest.mark.parametrize("name,k,table", ROBUST_FUNCTIONS)
def test_robustFunction(orc, name, k, table):
    robust_function_check(orc.oracle_backend, name, k, table)


def huber(k, base):
    return noiseModel.Robust.Create(noiseModel.mEstimator.Huber.Create(k), base)


def robust_optimization_cases():
    """tests/testNonlinearOptimizer.cpp:351-482: (graph, initial, expected, tolerance, ordering)."""
    iso = noiseModel.Isotropic.Sigma
    fg = NonlinearFactorGraph()
    fg.addPrior(0, Pose2(0, 0, 0), iso(3, 1))
    fg.add(BetweenFactor(0, 1, Pose2(1, 1.1, math.pi / 4), huber(2.0, iso(3, 1))))
    fg.add(BetweenFactor(0, 1, Pose2(1, 0.9, math.pi / 2), huber(3.0, iso(3, 1))))
    init = Values()
    init.insert(0, Pose2(0, 0, 0))
    init.insert(1, Pose2(0.961187, 0.99965, 1.1781))
    yield "Pose2OptimizationWithHuberNoOutlier", fg, init, {0: Pose2(0, 0, 0), 1: Pose2(0.961187, 0.99965, 1.1781)}, 3e-2

    fg = NonlinearFactorGraph()
    fg.addPrior(0, Point2(0, 0), iso(2, 0.01))
    for z in ((1, 1.8), (1, 0.9), (1, 90)):
        fg.add(BetweenFactor(0, 1, Point2(*z), huber(1.0, iso(2, 1))))
    init = Values()
    init.insert(0, Point2(1, 1))
    init.insert(1, Point2(1, 0))
    yield "Point2LinearOptimizationWithHuber", fg, init, {0: Point2(0, 0), 1: Point2(1, 1.85)}, 1e-4

    fg = NonlinearFactorGraph()
    fg.addPrior(0, Pose2(0, 0, 0), iso(3, 0.1))
    for z in ((0, 9, math.pi / 2), (0, 11, math.pi / 2), (0, 10, math.pi / 2), (0, 9, 0)):
        fg.add(BetweenFactor(0, 1, Pose2(*z), huber(0.2, iso(3, 1))))
    init = Values()
    init.insert(0, Pose2(0, 0, 0))
    init.insert(1, Pose2(0, 10, math.pi / 4))
    yield "Pose2OptimizationWithHuber", fg, init, {0: Pose2(0, 0, 0), 1: Pose2(0, 10, 1.45212)}, 1e-1

    fg = NonlinearFactorGraph()
    for pt in (-10, -3, -1, 1, 3, 10, 1000):
        fg.add(PriorFactor(0, np.array([float(pt)]), huber(20, iso(1, 1))))
    init = Values()
    init.insert(0, np.array([100.0]))
    yield "RobustMeanCalculation", fg, init, {0: np.array([3.33333333])}, 1e-5


def robust_optimization_check(backend_factory):
    from gtsam_petercdev_amd.graph import DoglegOptimizer
    for name, fg, init, expected, tol in robust_optimization_cases():
        order = Ordering(sorted(init.keys()))
        results = [GaussNewtonOptimizer(fg, init, order, backend_factory=backend_factory).optimize(),
                   LevenbergMarquardtOptimizer(fg, init, order, LevenbergMarquardtParams(),
                                               backend_factory=backend_factory).optimize(),
                   DoglegOptimizer(fg, init, order, relativeErrorTol=1e-10 if name == "RobustMeanCalculation" else 1e-5,
                                   backend_factory=backend_factory).optimize()]
        for res in results:
            for k, e in expected.items():
                a = res.at(k)
                if isinstance(e, np.ndarray):
                    assert np.allclose(a, e, atol=tol), (name, a, e)
                else:
                    assert a.equals(e, tol), (name, k)


def test_optimization_with_Huber(orc):
    """tests/testNonlinearOptimizer.cpp:351-482 — GN, LM and Dogleg with Huber-robust factors reach the expected values."""
    robust_optimization_check(orc.oracle_backend)


# ---- gtsam/slam/tests/testProjectionFactor.cpp ------------------------------------------------------------------
def projection_factor_check(backend_factory):
    """:96-115 Error = (-3, 0); :141-163 Jacobians H1 / H2 (1e-3); cheirality: zero Jacobians, error (2 fx, 2 fx)."""
    from gtsam_petercdev_amd.graph import Cal3_S2, GenericProjectionFactor, Pose3, Rot3, Point3
    K = Cal3_S2(60, 640, 480)
    assert abs(K.fx() - 554.256) < 1e-3
    g = NonlinearFactorGraph()
    g.add(GenericProjectionFactor([323.0, 240.0], noiseModel.Unit.Create(2), X(1), L(1), K))
    v = Values()
    v.insert(X(1), Pose3(Rot3(), Point3(0, 0, -6)))
    v.insert(L(1), Point3(0.0, 0.0, 0.0))
    be = backend_factory(g.to_arrays(v))
    assert abs(be.error() - 0.5 * 9.0) < 1e-9                      # |(-3, 0)|^2 / 2
    be.linearize()
    J = be.jacobians().reshape(10, 2).T                              # 2 x (6 pose | 3 point | b): the factor's key order
    H1, H2, b = J[:, :6], J[:, 6:9], J[:, 9]
    assert np.allclose(b, [3.0, 0.0], atol=1e-9)                     # b = -(h(x) - z)
    assert np.allclose(H1, [[0., -554.256, 0., -92.376, 0., 0.], [554.256, 0., 0., 0., -92.376, 0.]], atol=1e-3)
    assert np.allclose(H2, [[92.376, 0., 0.], [0., 92.376, 0.]], atol=1e-3)
    # point behind the camera (ProjectionFactor.h:154-165)
    v2 = Values()
    v2.insert(X(1), Pose3(Rot3(), Point3(0, 0, 6)))
    v2.insert(L(1), Point3(0.0, 0.0, 0.0))
    be2 = backend_factory(g.to_arrays(v2))
    assert abs(be2.error() - 0.5 * 2 * (2 * K.fx()) ** 2) < 1e-6
    be2.linearize()
    J2 = be2.jacobians().reshape(10, 2).T
    assert np.all(J2[:, :9] == 0.0) and np.allclose(J2[:, 9], -2 * K.fx())


def test_ProjectionFactor(orc):
    projection_factor_check(orc.oracle_backend)


# ---- tests/testMarginals.cpp ----------------------------------------------------------------------------------------
def planar_slam_linear_graph():
    """tests/testMarginals.cpp:40-107 (PlanarSLAMSelfContained_advanced) linearized at its solution: prior + two odometry
    BetweenFactor<Pose2> + three BearingRangeFactor<Pose2, Point2>.  BearingRange is not a factor type of this path, so the
    whole graph is given as JacobianFactors (its Jacobians at the linearization point, in the Pose2 tangent (v_x, v_y,
    omega) of the body frame / Point2), which is all Marginals sees.  Keys: x1..x3 = 1..3, l1, l2 = 11, 12."""
    poses = {1: (0.0, 0.0, 0.0), 2: (2.0, 0.0, 0.0), 3: (4.0, 0.0, 0.0)}
    lms = {11: (2.0, 2.0), 12: (4.0, 2.0)}
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(1, np.eye(3), np.zeros(3), noiseModel.Diagonal.Sigmas([0.3, 0.3, 0.1])))
    odo = noiseModel.Diagonal.Sigmas([0.2, 0.2, 0.1])
    for a, b in ((1, 2), (2, 3)):
        # BetweenFactor<Pose2> at zero error: H1 = -Ad(h^-1) with h = (2, 0, 0), H2 = I  (BetweenFactor.h:111-124)
        Ad_inv = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 2.0], [0.0, 0.0, 1.0]])  # Ad of (-2, 0, 0): [[c,-s,y],[s,c,-x],[0,0,1]]
        fg.add(JacobianFactor(a, -Ad_inv, b, np.eye(3), np.zeros(3), odo))
    meas = noiseModel.Diagonal.Sigmas([0.1, 0.2])
    for xk, lk in ((1, 11), (2, 11), (3, 12)):
        x, y, th = poses[xk]
        c, s = math.cos(th), math.sin(th)
        d = np.array(lms[lk]) - np.array([x, y])
        q = np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1]])     # point in the body frame
        r2 = q @ q
        r = math.sqrt(r2)
        dq_dpose = np.array([[-1.0, 0.0, q[1]], [0.0, -1.0, -q[0]]])  # Pose2::transformTo, Pose2.cpp
        dq_dpoint = np.array([[c, s], [-s, c]])
        db_dq = np.array([-q[1], q[0]]) / r2                           # bearing = atan2(q_y, q_x)
        dr_dq = q / r
        Hpose = np.stack([db_dq @ dq_dpose, dr_dq @ dq_dpose])
        Hpoint = np.stack([db_dq @ dq_dpoint, dr_dq @ dq_dpoint])
        fg.add(JacobianFactor(xk, Hpose, lk, Hpoint, np.zeros(2), meas))
    return fg


PLANAR_SLAM_MARGINALS = {  # tests/testMarginals.cpp:76-100, tolerance 1e-8
    1: [[0.09, 0, 0], [0, 0.09, 0], [0, 0, 0.01]],
    2: [[0.120967742, -0.00129032258, 0.00451612903], [-0.00129032258, 0.158387097, 0.0206451613],
        [0.00451612903, 0.0206451613, 0.0177419355]],
    3: [[0.160967742, 0.00774193548, 0.00451612903], [0.00774193548, 0.351935484, 0.0561290323],
        [0.00451612903, 0.0561290323, 0.0277419355]],
    11: [[0.168709677, -0.0477419355], [-0.0477419355, 0.163548387]],
    12: [[0.293870968, -0.104516129], [-0.104516129, 0.391935484]],
}


def planar_slam_marginals_check(backend_factory, orderings):
    arrays = planar_slam_linear_graph().to_arrays(None)
    arrays.values = np.zeros(int(arrays.var_dims.sum()))
    for ordering in orderings:
        be = backend_factory(arrays)
        be.set_ordering(ordering)
        be.linearize()
        for key, expected in PLANAR_SLAM_MARGINALS.items():
            assert np.allclose(be.marginal_covariance(key), expected, atol=1e-8), (ordering, key)


def planar_slam_nonlinear_graph():
    """tests/testMarginals.cpp:40-75 with its own factor types: prior + BetweenFactor<Pose2> + BearingRangeFactor<Pose2,
    Point2>, and the linearization point `soln` (lines 78-83)."""
    from gtsam_petercdev_amd.graph import BearingRangeFactor
    g = NonlinearFactorGraph()
    g.add(PriorFactor(1, Pose2(0.0, 0.0, 0.0), noiseModel.Diagonal.Sigmas([0.3, 0.3, 0.1])))
    odo = noiseModel.Diagonal.Sigmas([0.2, 0.2, 0.1])
    g.add(BetweenFactor(1, 2, Pose2(2.0, 0.0, 0.0), odo))
    g.add(BetweenFactor(2, 3, Pose2(2.0, 0.0, 0.0), odo))
    meas = noiseModel.Diagonal.Sigmas([0.1, 0.2])
    g.add(BearingRangeFactor(1, 11, math.radians(45), math.sqrt(8.0), meas))
    g.add(BearingRangeFactor(2, 11, math.radians(90), 2.0, meas))
    g.add(BearingRangeFactor(3, 12, math.radians(90), 2.0, meas))
    v = Values()
    v.insert(1, Pose2(0.0, 0.0, 0.0))
    v.insert(2, Pose2(2.0, 0.0, 0.0))
    v.insert(3, Pose2(4.0, 0.0, 0.0))
    v.insert(11, Point2(2.0, 2.0))
    v.insert(12, Point2(4.0, 2.0))
    return g, v


def bearing_range_check(backend_factory, orderings):
    """BearingRangeFactor<Pose2, Point2>: (1) the reference's planar-SLAM marginals with the nonlinear factors
    (tests/testMarginals.cpp:40-107, tolerance 1e-8), (2) zero error at that solution, (3) the factor's Jacobians against
    central differences of its error, as gtsam/sam/tests/testBearingRangeFactor.cpp:45-57 does (1e-5 there),
    (4) the bearing wraps (Rot2 local coordinates), (5) LM from a disturbed estimate returns to the solution."""
    from gtsam_petercdev_amd.graph import BearingRangeFactor
    g, v = planar_slam_nonlinear_graph()
    arrays = g.to_arrays(v)
    for ordering in orderings:
        be = backend_factory(arrays)
        be.set_ordering(ordering)
        assert abs(be.error()) < 1e-18
        be.linearize()
        for key, expected in PLANAR_SLAM_MARGINALS.items():
            assert np.allclose(be.marginal_covariance(key), expected, atol=1e-8), (ordering, key)
    # (3) testBearingRangeFactor.cpp: factor2D(poseKey, pointKey, 1, 2, Isotropic(2, 0.5)), pose (1, 2, 0.3)?  The test
    # uses its own values; any generic point will do for a derivative check.
    g1 = NonlinearFactorGraph()
    g1.add(BearingRangeFactor(1, 2, 1.0, 2.0, noiseModel.Isotropic.Sigma(2, 0.5)))
    x0 = np.array([1.0, 2.0, 0.3, -4.0, 11.0])

    def whitened_error(x):
        vv = Values()
        vv.insert(1, Pose2(*x[:3]))
        vv.insert(2, Point2(*x[3:]))
        b = backend_factory(g1.to_arrays(vv))
        b.linearize()
        return -b.jacobians().reshape(6, 2).T[:, 5].copy(), b

    e0, b0 = whitened_error(x0)
    J = b0.jacobians().reshape(6, 2).T[:, :5]
    c, s = math.cos(x0[2]), math.sin(x0[2])
    h = 1e-6
    for k in range(5):
        step = np.zeros(5)
        if k < 2:  # Pose2 retract: translation moves in the body frame (Pose2::ChartAtOrigin::Retract, Pose2.cpp)
            step[:2] = np.array([[c, -s], [s, c]])[:, k] * h
        else:
            step[k] = h
        ep, _ = whitened_error(x0 + step)
        em, _ = whitened_error(x0 - step)
        assert np.allclose((ep - em) / (2 * h), J[:, k], atol=1e-5), k
    # (4) measured bearing near +pi, predicted near -pi: the error is the short way round
    g2 = NonlinearFactorGraph()
    g2.add(BearingRangeFactor(1, 2, math.pi - 0.01, 1.0, noiseModel.Unit.Create(2)))
    v2 = Values()
    v2.insert(1, Pose2(0.0, 0.0, 0.0))
    v2.insert(2, Point2(-1.0, -0.02))
    be = backend_factory(g2.to_arrays(v2))
    eb = math.atan2(-0.02, -1.0) - (math.pi - 0.01) + 2 * math.pi
    er = math.hypot(1.0, 0.02) - 1.0
    assert abs(be.error() - 0.5 * (eb * eb + er * er)) < 1e-12 and abs(eb) < 0.05
    # (5)
    vi = Values()
    vi.insert(1, Pose2(0.3, -0.2, 0.15))
    vi.insert(2, Pose2(2.4, 0.3, -0.2))
    vi.insert(3, Pose2(3.7, -0.3, 0.1))
    vi.insert(11, Point2(1.6, 2.5))
    vi.insert(12, Point2(4.4, 1.7))
    params = LevenbergMarquardtParams()
    params.ordering = Ordering([11, 12, 1, 2, 3])
    opt = LevenbergMarquardtOptimizer(g, vi, params, backend_factory=backend_factory)
    opt.optimize()
    assert opt.error() < 1e-10
    assert np.allclose(opt.backend.get_values(), arrays.values, atol=1e-5)


def test_BearingRangeFactor2D(orc):
    bearing_range_check(orc.oracle_backend, [[1, 2, 3, 11, 12]])


PLANAR_SLAM_JOINT_L2X1X3 = np.array([  # tests/testMarginals.cpp:111-120, order (l2, x1, x3), tolerance 1e-6
    [0.293871159514111, -0.104516127560770, 0.090000180000270, -0.000000000000000, -0.020000000000000, 0.151935669757191, -0.104516127560770, -0.050967744878460],
    [-0.104516127560770, 0.391935664055174, 0.000000000000000, 0.090000180000270, 0.040000000000000, 0.007741936219615, 0.351935664055174, 0.056129031890193],
    [0.090000180000270, 0.000000000000000, 0.090000180000270, -0.000000000000000, 0.000000000000000, 0.090000180000270, 0.000000000000000, 0.000000000000000],
    [-0.000000000000000, 0.090000180000270, -0.000000000000000, 0.090000180000270, 0.000000000000000, -0.000000000000000, 0.090000180000270, 0.000000000000000],
    [-0.020000000000000, 0.040000000000000, 0.000000000000000, 0.000000000000000, 0.010000000000000, 0.000000000000000, 0.040000000000000, 0.010000000000000],
    [0.151935669757191, 0.007741936219615, 0.090000180000270, -0.000000000000000, 0.000000000000000, 0.160967924878730, 0.007741936219615, 0.004516127560770],
    [-0.104516127560770, 0.351935664055174, 0.000000000000000, 0.090000180000270, 0.040000000000000, 0.007741936219615, 0.351935664055174, 0.056129031890193],
    [-0.050967744878460, 0.056129031890193, 0.000000000000000, 0.000000000000000, 0.010000000000000, 0.004516127560770, 0.056129031890193, 0.027741936219615]])


def planar_slam_joint_marginals_check(backend_factory, orderings):
    """Marginals::jointMarginalCovariance on the planar-SLAM example: 3 variables, 2 variables, 1 variable
    (tests/testMarginals.cpp:109-157; the expected blocks to 1e-6)."""
    g, v = planar_slam_nonlinear_graph()
    arrays = g.to_arrays(v)
    for ordering in orderings:
        be = backend_factory(arrays)
        be.set_ordering(ordering)
        be.linearize()
        J = be.joint_marginal_covariance([12, 1, 3])
        assert J.shape == (8, 8) and np.allclose(J, PLANAR_SLAM_JOINT_L2X1X3, atol=1e-6), ordering
        assert np.allclose(J, J.T, atol=1e-15)
        J2 = be.joint_marginal_covariance([12, 1])
        assert np.allclose(J2, PLANAR_SLAM_JOINT_L2X1X3[:5, :5], atol=1e-6), ordering
        J1 = be.joint_marginal_covariance([1])
        assert np.allclose(J1, PLANAR_SLAM_MARGINALS[1], atol=1e-6), ordering
        # the diagonal blocks are the single-variable marginals, whatever the order of the request
        Jr = be.joint_marginal_covariance([3, 12, 1])
        assert np.allclose(Jr[:3, :3], be.marginal_covariance(3), atol=1e-12)
        assert np.allclose(Jr[3:5, 3:5], be.marginal_covariance(12), atol=1e-12)
        assert np.allclose(Jr[:3, 3:5], J[5:8, 0:2], atol=1e-12)


def test_planarSLAMjointMarginals(orc):
    planar_slam_joint_marginals_check(orc.oracle_backend, [[1, 2, 3, 11, 12]])
    # the class mirror, with the reference's own symbols (blocks come back sorted by key: l2 < x1 < x3)
    from gtsam_petercdev_amd.graph import Marginals, BearingRangeFactor
    g = NonlinearFactorGraph()
    x1, x2, x3, l1, l2 = X(1), X(2), X(3), L(1), L(2)
    g.add(PriorFactor(x1, Pose2(0.0, 0.0, 0.0), noiseModel.Diagonal.Sigmas([0.3, 0.3, 0.1])))
    odo = noiseModel.Diagonal.Sigmas([0.2, 0.2, 0.1])
    g.add(BetweenFactor(x1, x2, Pose2(2.0, 0.0, 0.0), odo))
    g.add(BetweenFactor(x2, x3, Pose2(2.0, 0.0, 0.0), odo))
    meas = noiseModel.Diagonal.Sigmas([0.1, 0.2])
    g.add(BearingRangeFactor(x1, l1, math.radians(45), math.sqrt(8.0), meas))
    g.add(BearingRangeFactor(x2, l1, math.radians(90), 2.0, meas))
    g.add(BearingRangeFactor(x3, l2, math.radians(90), 2.0, meas))
    v = Values()
    for k, p in ((x1, Pose2(0.0, 0.0, 0.0)), (x2, Pose2(2.0, 0.0, 0.0)), (x3, Pose2(4.0, 0.0, 0.0))):
        v.insert(k, p)
    v.insert(l1, Point2(2.0, 2.0))
    v.insert(l2, Point2(4.0, 2.0))
    m = Marginals(g, v, ordering=[x1, x2, x3, l1, l2], backend_factory=orc.oracle_backend)
    assert np.allclose(m.marginalCovariance(x2), PLANAR_SLAM_MARGINALS[2], atol=1e-8)
    joint = m.jointMarginalCovariance([x1, l2, x3])
    assert joint.keys() == [l2, x1, x3]
    E = PLANAR_SLAM_JOINT_L2X1X3
    assert np.allclose(joint(l2, l2), E[0:2, 0:2], atol=1e-6) and np.allclose(joint(x1, l2), E[2:5, 0:2], atol=1e-6)
    assert np.allclose(joint(x3, l2), E[5:8, 0:2], atol=1e-6) and np.allclose(joint(x1, x3), E[2:5, 5:8], atol=1e-6)
    assert np.allclose(joint.fullMatrix(), E, atol=1e-6)


def test_planarSLAMmarginals(orc):
    """The reference's expected marginal covariances (tests/testMarginals.cpp:76-107)."""
    planar_slam_marginals_check(orc.oracle_backend, [[1, 2, 3, 11, 12]])


# ---- the reference's drivers on its TORO / "graph" data files -------------------------------------------------------
def toro_example_problems(golden_dir):
    """(name, arrays, ordering kind) of three of the reference's own runs:
    examples/Pose2SLAMExample_graph.cpp:33-51 — w100.graph with Diagonal(0.05, 0.05, 5 deg), prior (0.01)^3 on pose 0;
    matlab/gtsam_examples/PlanarSLAMExample_graph.m:17-26 — example.graph (odometry + bearing-range) with
    Diagonal(0.05, 0.05, 2 deg), prior (0.1, 0.1, 2 deg) on pose 40 at its initial value;
    matlab/gtsam_examples/Pose3SLAMExample_graph.m:17-40 — sphere2500.txt with Diagonal(5 deg x3, 0.05 x3), pose 0
    pinned (NonlinearEqualityPose3 there; a 1e-6 prior here) — first 400 poses, to keep the oracle quick.
    None of these runs has a published number in the reference: their results are parity unpinned; the tests check
    convergence and, on the GPU, agreement with the oracle."""
    from gtsam_petercdev_amd import _lib
    a = _lib.load2d(os.path.join(golden_dir, "w100.graph"), model_sigmas=[0.05, 0.05, 5.0 * math.pi / 180.0])
    a = a.with_factor(A.F_PRIOR, [0], 3, [0.0, 0.0, 0.0], A.NOISE_DIAGONAL, [0.01, 0.01, 0.01])
    yield "w100", a, A.ORDER_MINDEGREE
    b = _lib.load2d(os.path.join(golden_dir, "example.graph"), model_sigmas=[0.05, 0.05, 2.0 * math.pi / 180.0])
    i40 = int(np.nonzero(b.var_keys == 40)[0][0])
    off = int(np.sum(np.where(b.var_types[:i40] == A.VAR_POSE2, 3, 2)))
    b = b.with_factor(A.F_PRIOR, [i40], 3, b.values[off:off + 3], A.NOISE_DIAGONAL, [0.1, 0.1, 2.0 * math.pi / 180.0])
    yield "planar_example_graph", b, A.ORDER_MINDEGREE
    c = _lib.read_g2o(os.path.join(golden_dir, "sphere2500.txt"), is3D=True)
    yield "sphere2500", c, A.ORDER_ND


def test_toro_examples_converge(orc, golden_dir):
    from gtsam_petercdev_amd import _lib
    for name, arr, kind in toro_example_problems(golden_dir):
        if name == "sphere2500":
            continue  # (2500 poses: the GPU test runs it against the oracle)
        ob = orc.oracle_backend(arr)
        ob.set_ordering(_lib.ProductBackend(arr, host_only=True).compute_ordering(kind))
        r = ob.lm_optimize(A.lm_params_legacy())
        assert r["final_error"] < 0.25 * r["initial_error"], name   # (the floor is the measurement noise)
        assert r["iterations"] < 100, name


# ---- the oracle's threaded loops (the reference's TBB loops) change nothing but the wall time -------------------------------
@pytest.mark.parametrize("kind", ["bal", "pose3"])
def test_oracle_threads_are_bitwise_identical_to_serial(oracle, kind):
    """orc_set_threads: factors in linearize (NonlinearFactorGraph.cpp:214-261) and independent subtrees in elimination /
    back-substitution (parallelTraversalTasks.h:35-156) on std::thread — same Jacobians, same Bayes tree in the same
    clique order, same conditionals, same solution, bit for bit, as the serial traversal."""
    from gtsam_petercdev_amd import datasets, _lib
    if kind == "bal":
        arr, okind = datasets.synth_bal_arrays(30, 3000, 12000, seed=3, long_range=0.3), A.ORDER_SCHUR_ND
    else:
        arr, okind = datasets.synth_manhattan_pose3(6000, seed=5), A.ORDER_ND
    ordering = _lib.ProductBackend(arr, host_only=True).compute_ordering(okind)
    runs = []
    for threads in (1, 4):
        ob = oracle.oracle_backend(arr)
        ob.set_threads(threads)
        ob.set_ordering(ordering)
        ob.linearize()
        jac = ob.jacobians()
        delta = ob.solve(1e-4, False)
        parent, fronts = ob.get_tree()
        conds = [ob.conditional(c) for c in (0, len(fronts) // 2, len(fronts) - 1)]
        _, tree = ob.timing()
        runs.append((jac, delta, list(parent), fronts, conds, tree))
    a, b = runs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a[3], b[3]))
    assert all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))
    assert a[5] == b[5]


# ---- hard constraints: noiseModel::Constrained and the QR elimination of a clique that holds one -------------------------------
def _linear_dependent_rows(expected, actual, tol):
    """assert the rows of the two matrices are pairwise parallel (linear_dependent of gtsam/base/Matrix.cpp: what the
    reference's tests compare the constrained QR with — a constraint row keeps its own scale)."""
    expected, actual = np.asarray(expected, float), np.asarray(actual, float)
    assert expected.shape == actual.shape
    for e, a in zip(expected, actual):
        if np.allclose(e, 0, atol=tol):
            assert np.allclose(a, 0, atol=tol)
            continue
        k = np.argmax(np.abs(e))
        assert abs(a[k]) > tol, (e, a)
        assert np.allclose(a * (e[k] / a[k]), e, atol=10 * tol), (e, a)


def test_NoiseModel_constrained_QR(orc):
    """gtsam/linear/tests/testNoiseModel.cpp:225-248 (the Constrained half of QR on exampleQR::Ab, :205-222)."""
    Ab = np.array([[-1., 0., 1., 0., 0., 0., -0.2],
                   [0., -1., 0., 1., 0., 0., 0.3],
                   [1., 0., 0., 0., -1., 0., 0.2],
                   [0., 1., 0., 0., 0., -1., -0.1]])
    Rd, sigmas, _ = orc.constrained_qr(Ab, [0.2, 0.2, 0.1, 0.1])
    assert np.allclose(sigmas, [0.0894427, 0.0894427, 0.223607, 0.223607], atol=1e-6)
    expected = np.array([[1., 0., -0.2, 0., -0.8, 0., 0.2],
                         [0., 1., 0., -0.2, 0., -0.8, -0.14],
                         [0., 0., 1., 0., -1., 0., 0.0],
                         [0., 0., 0., 1., 0., -1., 0.2]])
    _linear_dependent_rows(expected, Rd, 1e-6)


def test_NoiseModel_OverdeterminedQR(orc):
    """testNoiseModel.cpp:251-282 (Constrained version on unit sigmas: rows not divided by sigma)."""
    Ab = np.vstack([[0, 1, 0, 0], [0, 0, 1, 0], np.ones((7, 4))]).astype(float)
    Rd, sigmas, _ = orc.constrained_qr(Ab, np.ones(9))
    assert np.allclose(sigmas, [0.377964473, 1, 1], atol=1e-6)
    expected = np.zeros((9, 4))
    expected[0] = np.array([2.64575131] * 4) * 0.377964473
    expected[1, 1] = 1
    expected[2, 2] = 1
    assert np.allclose(Rd, expected, atol=1e-6)


def test_NoiseModel_MixedQR(orc):
    """testNoiseModel.cpp:285-315."""
    Ab = np.array([[1, 0, 0, 0, 0, 1, 0],
                   [0, 0, 0, 0, 1, 0, 0],
                   [0, 0, 1, 1, 0, 0, 0],
                   [0, 1, 0, 0, 0, 0, 0],
                   [0, 0, 0, 0, 0, 1, 0]], float)
    Rd, sigmas, _ = orc.constrained_qr(Ab, [0, 1, 0, 1, 1])
    assert np.allclose(sigmas, [0, 1, 0, 1, 1], atol=1e-6)
    expected = np.array([[1, 0, 0, 0, 0, 1, 0],
                         [0, 1, 0, 0, 0, 0, 0],
                         [0, 0, 1, 1, 0, 0, 0],
                         [0, 0, 0, 0, 1, 0, 0],
                         [0, 0, 0, 0, 0, 1, 0]], float)
    _linear_dependent_rows(expected, Rd, 1e-6)


def test_NoiseModel_MixedQR2(orc):
    """testNoiseModel.cpp:318-354: x = z and y = z — all the measurements are measurements of z."""
    rows = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [-1, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0],
            [0, -1, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]
    sg = np.ones(11)
    sg[3] = sg[7] = 0
    Rd, sigmas, _ = orc.constrained_qr(np.array(rows, float), sg)
    assert np.allclose(sigmas, [0, 0, 1.0 / 3], atol=1e-6)
    expected = np.zeros((11, 4))
    expected[0] = [-1, 0, 1, 0]
    expected[1] = [0, -1, 1, 0]
    expected[2] = [0, 0, 1, 0]
    assert np.allclose(Rd, expected, atol=1e-6)


def test_NoiseModel_FullyConstrained_and_QRNan(orc):
    """testNoiseModel.cpp:357-390."""
    Ab = np.array([[1, 0, 0, 0, 0, 1, 2], [0, 0, 1, 1, 0, 0, 4], [0, 1, 0, 1, 1, 1, 8]], float)
    Rd, sigmas, _ = orc.constrained_qr(Ab, np.zeros(3))
    assert np.array_equal(sigmas, [0, 0, 0])
    _linear_dependent_rows([[1, 0, 0, 0, 0, 1, 2], [0, 1, 0, 1, 1, 1, 8], [0, 0, 1, 1, 0, 0, 4]], Rd, 1e-6)
    Rd, sigmas, _ = orc.constrained_qr(np.array([[2, 4, 2, 4, 6], [2, 1, 2, 4, 4]], float), np.zeros(2))
    assert np.array_equal(sigmas, [0, 0])
    _linear_dependent_rows([[1, 2, 1, 2, 3], [0, 1, 0, 0, 2.0 / 3]], Rd, 1e-9)


def test_JacobianFactor_constraint_eliminate(orc):
    """gtsam/linear/tests/testJacobianFactor.cpp:588-641: constraint_eliminate1 (x = v), constraint_eliminate2 (R = [1 2; 0 1],
    S = [1 2; 0 0] up to the scale of a constraint row)."""
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(1, np.eye(2), [1.2, 3.4], noiseModel.Constrained.All(2)))
    be = orc.oracle_backend(fg.to_arrays())
    be.set_ordering([1])
    be.linearize()
    assert np.allclose(be.solve(0.0, False), [1.2, 3.4], atol=1e-12)
    _linear_dependent_rows([[1, 0, 1.2], [0, 1, 3.4]], be.conditional(0), 1e-12)
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(1, [[2, 4], [2, 1]], 2, [[2, 4], [2, 4]], [3.0, 4.0], noiseModel.Constrained.All(2)))
    fg.add(JacobianFactor(2, np.eye(2), [0.0, 0.0], noiseModel.Unit.Create(2)))   # (so that the graph can be solved)
    be = orc.oracle_backend(fg.to_arrays())
    be.set_ordering([1, 2])
    be.linearize()
    be.solve(0.0, False)
    # (the unit prior joins the clique: the first two rows of its conditional are the constraint's.  GaussianConditional::
    #  equals compares the rows of [R S] only, up to scale — gtsam/linear/GaussianConditional.cpp:132-169 — so d is not
    #  part of the golden; NoiseModel QRNan above pins the same elimination with its right-hand side.)
    _linear_dependent_rows([[1, 2, 1, 2], [0, 1, 0, 0]], be.conditional(0)[:2, :4], 1e-9)


_X_, _Y_, _Z_ = 0, 1, 2   # tests/smallExample.h: _x_ = 0, _y_ = 1, _z_ = 2


def constrained_linear_graphs():
    """createSimpleConstraintGraph / createSingleConstraintGraph / createMultiConstraintGraph with their solutions
    (tests/smallExample.h:471-607), solved by tests/testGaussianFactorGraphB.cpp:300-340."""
    s01 = noiseModel.Isotropic.Sigma(2, 0.1)
    con = noiseModel.Constrained.All(2)
    I = np.eye(2)
    simple = GaussianFactorGraph()
    simple.add(JacobianFactor(_X_, I, [1.0, -1.0], s01))
    simple.add(JacobianFactor(_X_, I, _Y_, -I, [0.0, 0.0], con))
    single = GaussianFactorGraph()
    single.add(JacobianFactor(_X_, I, [1.0, -1.0], s01))
    single.add(JacobianFactor(_X_, [[1, 2], [2, 1]], _Y_, 10 * I, [1.0, 2.0], con))
    multi = GaussianFactorGraph()
    multi.add(JacobianFactor(_X_, I, [-2.0, 2.0], s01))
    multi.add(JacobianFactor(_X_, [[1, 2], [2, 1]], _Y_, 10 * I, [1.0, 2.0], con))
    multi.add(JacobianFactor(_X_, [[3, 4], [-1, -2]], _Z_, [[1, 1], [1, 2]], [3.0, 4.0], con))
    return [("simple", simple, {_X_: [1.0, -1.0], _Y_: [1.0, -1.0]}),
            ("single", single, {_X_: [1.0, -1.0], _Y_: [0.2, 0.1]}),
            ("multi", multi, {_X_: [-2.0, 2.0], _Y_: [-0.1, 0.4], _Z_: [-4.0, 5.0]})]


def constrained_linear_check(backend_factory):
    import itertools
    for name, fg, expected in constrained_linear_graphs():
        for ordering in itertools.permutations(sorted(expected)):
            actual = fg.optimize(list(ordering), backend_factory=backend_factory)
            for k, v in expected.items():
                assert np.allclose(actual[k], v, atol=1e-9), (name, ordering, k, actual[k], v)


def test_GaussianFactorGraph_constrained(orc):
    constrained_linear_check(orc.oracle_backend)
