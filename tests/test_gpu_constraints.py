"""Hard constraints (noiseModel::Constrained, zero-sigma rows) through the HIP path — SURVEY 8(f) f2.

The reference eliminates a clique that holds such a row with EliminateQR / Constrained::QR (HessianFactor.cpp:538-551,
NoiseModel.cpp:503-620); the product rewrites the assembled front (constraint.hip) and factors it with the blocked Cholesky.
Checked against: the reference's known answers (tests/smallExample.h constrained graphs), the oracle's restatement of the
QR path (pinned by testNoiseModel / testJacobianFactor goldens in tests/test_oracle_golden.py), and a dense KKT solve."""
from __future__ import annotations

import numpy as np
import pytest

from gtsam_petercdev_amd import _abi as A
from gtsam_petercdev_amd import datasets
from gtsam_petercdev_amd.graph import GaussianFactorGraph, JacobianFactor, noiseModel

pytestmark = pytest.mark.gpu
# GSX_FUZZ_OFFSET=<n> moves every structure fuzz below to other seeds (an occasional wider sweep; the default is what CI runs)
FUZZ_OFFSET = int(__import__("os").environ.get("GSX_FUZZ_OFFSET", "0"))


@pytest.fixture(scope="module")
def gpu():
    from gtsam_petercdev_amd import _lib
    assert _lib.device_count() > 0, "no GPU visible: the HIP path has no fallback"
    return _lib


def linear_arrays(fg: GaussianFactorGraph) -> A.ProblemArrays:
    arrays = fg.to_arrays(None)
    arrays.values = np.zeros(int(arrays.var_dims.sum()))
    return arrays


def with_constraints(arr: A.ProblemArrays, models: dict) -> A.ProblemArrays:
    """A copy of the problem where factor f gets the model models[f] = sigmas (zeros = hard constraints; a DIAGONAL model,
    mu = 1000) or (sigmas, mu) (GSX_NOISE_CONSTRAINED)."""
    kinds = arr.f_noise_kind.copy()
    ptr, noise = [0], []
    for f in range(arr.n_factors):
        if f in models:
            m = models[f]
            if isinstance(m, tuple):
                sig, mu = np.asarray(m[0], float), np.broadcast_to(np.asarray(m[1], float), np.shape(m[0]))
                par = np.concatenate([sig, mu])
                kinds[f] = A.NOISE_CONSTRAINED
            else:
                par = np.asarray(m, float)
                kinds[f] = A.NOISE_DIAGONAL
            assert par.size in (arr.f_rows[f], 2 * arr.f_rows[f])
        else:
            par = arr.noise[arr.f_noise_ptr[f]:arr.f_noise_ptr[f + 1]]
        noise.append(par)
        ptr.append(ptr[-1] + par.size)
    return A.ProblemArrays(arr.var_keys, arr.var_types, arr.var_dims, arr.f_type, arr.f_rows, arr.f_key_ptr, arr.f_vars,
                           arr.f_meas_ptr, arr.meas, kinds, ptr, np.concatenate(noise), arr.values.copy(), dict(arr.meta))


def constraint_rows(arr: A.ProblemArrays):
    """(factor, row, mu) of every hard-constraint row."""
    out = []
    for f in range(arr.n_factors):
        k = arr.f_noise_kind[f]
        if k not in (A.NOISE_DIAGONAL, A.NOISE_CONSTRAINED):
            continue
        par = arr.noise[arr.f_noise_ptr[f]:arr.f_noise_ptr[f + 1]]
        m = arr.f_rows[f]
        for r in range(m):
            if par[r] == 0.0:
                out.append((f, r, par[m + r] if k == A.NOISE_CONSTRAINED else 1000.0))
    return out


def dense_kkt_step(arr: A.ProblemArrays, jac: np.ndarray, con_scale, lam: float, damp: np.ndarray):
    """min 1/2 |A x - b|^2 + 1/2 lam x' diag(damp) x over the unconstrained rows, subject to the constraint rows — solved
    as one dense KKT system (an independent statement of what the elimination must return).  jac: the ORACLE's [A b] (the
    constraint rows unwhitened)."""
    n = int(arr.var_dims.sum())
    toff = arr.tangent_offsets()
    joff = arr.jacobian_offsets()
    con = {(f, r) for f, r, _ in con_scale}
    rows_A, rows_C = [], []
    for f in range(arr.n_factors):
        m = arr.f_rows[f]
        vs = arr.f_vars[arr.f_key_ptr[f]:arr.f_key_ptr[f + 1]]
        cols = int(arr.var_dims[vs].sum()) + 1
        M = jac[joff[f]:joff[f] + m * cols].reshape(cols, m).T
        for r in range(m):
            row = np.zeros(n + 1)
            c = 0
            for v in vs:
                d = arr.var_dims[v]
                row[toff[v]:toff[v] + d] = M[r, c:c + d]
                c += d
            row[n] = M[r, c]
            (rows_C if (f, r) in con else rows_A).append(row)
    Aa, Cc = np.array(rows_A), np.array(rows_C)
    H = Aa[:, :n].T @ Aa[:, :n] + lam * np.diag(damp)
    g = Aa[:, :n].T @ Aa[:, n]
    k = Cc.shape[0]
    KKT = np.block([[H, Cc[:, :n].T], [Cc[:, :n], np.zeros((k, k))]])
    sol = np.linalg.solve(KKT, np.concatenate([g, Cc[:, n]]))
    return sol[:n]


# ---- the reference's known answers ---------------------------------------------------------------------------------------------
def test_GaussianFactorGraph_constrained_known_answers(gpu):
    """tests/testGaussianFactorGraphB.cpp:300-340 (constrained_simple / constrained_single / constrained_multi1) with the
    graphs and solutions of tests/smallExample.h:471-607, every elimination order."""
    from tests.test_oracle_golden import constrained_linear_check
    constrained_linear_check(gpu.product_backend)


def test_JacobianFactor_constraint_eliminate1(gpu):
    """gtsam/linear/tests/testJacobianFactor.cpp:588-605: x = v exactly."""
    fg = GaussianFactorGraph()
    fg.add(JacobianFactor(1, np.eye(2), [1.2, 3.4], noiseModel.Constrained.All(2)))
    be = gpu.product_backend(linear_arrays(fg))
    be.set_ordering([1])
    be.linearize()
    assert np.allclose(be.solve(0.0, False), [1.2, 3.4], atol=1e-13)
    st = be.stats()
    assert st["n_constraint_rows"] == 2 and st["n_constrained_fronts"] == 1


# ---- nonlinear graphs with constraint rows, against the oracle's QR path and the KKT system ----------------------------------------
def _pose_problem(kind, n, seed):
    arr = datasets.synth_manhattan_pose2(n, seed=seed) if kind == "pose2" else datasets.synth_manhattan_pose3(n, seed=seed)
    d = 3 if kind == "pose2" else 6
    prior = int(np.flatnonzero(arr.f_type == A.F_PRIOR)[0])
    between = np.flatnonzero(arr.f_type == A.F_BETWEEN)
    models = {prior: np.zeros(d)}                                   # PriorFactor with Constrained::All(d)
    loop = [int(f) for f in between
            if abs(int(arr.f_vars[arr.f_key_ptr[f]]) - int(arr.f_vars[arr.f_key_ptr[f] + 1])) > 1]
    if loop:  # a loop closure known exactly in some directions (Constrained::MixedSigmas(mu, sigmas))
        sig = np.full(d, 0.1)
        sig[[0, d - 1]] = 0.0
        models[loop[len(loop) // 2]] = (sig, 50.0)
    chain = int(between[len(between) // 3])
    models[chain] = np.zeros(d)                                     # a rigid link: all d rows are constraints
    return with_constraints(arr, models)


@pytest.mark.parametrize("kind,n,order,relax", [("pose2", 60, A.ORDER_MINDEGREE, 0.0), ("pose2", 400, A.ORDER_ND, 0.5),
                                               ("pose3", 40, A.ORDER_MINDEGREE, 0.0), ("pose3", 300, A.ORDER_ND, 0.5),
                                               ("pose3", 1500, A.ORDER_ND, 0.5)])
def test_constrained_pose_graph(gpu, oracle, kind, n, order, relax):
    arr = _pose_problem(kind, n, seed=5)
    rows = constraint_rows(arr)
    assert len(rows) >= 2 * (3 if kind == "pose2" else 6)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(order)
    gb.set_amalgamation(relax, 64)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    st = gb.stats()
    assert st["n_constraint_rows"] == len(rows) and st["n_constrained_fronts"] >= 1
    # graph error: a violated constraint row weighs mu e^2 / 2 (Constrained::squaredMahalanobisDistance)
    assert abs(gb.error() - ob.error()) <= 1e-11 * abs(ob.error())
    gb.linearize()
    ob.linearize()
    # [A b]: the product keeps a constraint row scaled by sqrt(mu) (the weight the error functions give it)
    jg, jo = gb.jacobians(), ob.jacobians()
    joff = arr.jacobian_offsets()
    scale = np.ones_like(jo)
    for f, r, mu in rows:
        m = arr.f_rows[f]
        scale[joff[f] + r:joff[f + 1]:m] = np.sqrt(mu)
    assert np.allclose(jg, jo * scale, rtol=1e-10, atol=1e-12)
    hd = ob.hessian_diagonal()
    # (diag(J'J) takes a constraint row UNWHITENED, JacobianFactor::hessianDiagonalAdd — not with the sqrt(mu) it is stored with)
    assert np.allclose(gb.hessian_diagonal(), hd, rtol=1e-10, atol=1e-12)
    for lam, diag in [(0.0, False), (1e-3, False), (10.0, False), (1e-2, True)]:
        dg, do = gb.solve(lam, diag), ob.solve(lam, diag)
        # (the undamped 9000-dimensional system is the ill-conditioned one: 4e-8 measured, Cholesky of J'J against the
        #  oracle's QR of the constrained cliques)
        tol = 1e-6 if (n >= 1000 and lam == 0.0) else 1e-8
        assert np.linalg.norm(dg - do) <= tol * np.linalg.norm(do), (lam, diag, np.linalg.norm(dg - do))
        eg, eo = gb.linear_error(), ob.linear_error()
        assert np.allclose(eg, eo, rtol=100 * tol, atol=1e-10 * abs(eo[0])), (lam, diag, eg, eo)
        if n <= 400 and not diag:
            dk = dense_kkt_step(arr, jo, rows, lam, np.ones_like(hd))
            assert np.linalg.norm(dg - dk) <= 1e-8 * np.linalg.norm(dk)
        # the constraint rows hold exactly at the step
        toff = arr.tangent_offsets()
        for f, r, mu in rows:
            m = arr.f_rows[f]
            vs = arr.f_vars[arr.f_key_ptr[f]:arr.f_key_ptr[f + 1]]
            cols = int(arr.var_dims[vs].sum()) + 1
            M = jo[joff[f]:joff[f] + m * cols].reshape(cols, m).T
            x = np.concatenate([dg[toff[v]:toff[v] + arr.var_dims[v]] for v in vs])
            assert abs(M[r, :-1] @ x - M[r, -1]) <= 1e-9 * (1 + abs(M[r, -1]) + np.abs(M[r, :-1]).sum() * np.abs(x).max())
    p = A.lm_params_legacy()
    p.max_iterations = 10
    gb2, ob2 = gpu.product_backend(arr), oracle.oracle_backend(arr)
    gb2.set_amalgamation(relax, 64)
    gb2.set_ordering(ordering)
    ob2.set_ordering(ordering)
    rg, ro = gb2.lm_optimize(p), ob2.lm_optimize(p)
    assert rg["iterations"] == ro["iterations"]
    assert np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    fin = np.isfinite(ro["trace_error"])
    assert np.allclose(rg["trace_error"][fin], ro["trace_error"][fin], rtol=1e-6)
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * max(ro["final_error"], 1e-12)
    assert np.allclose(gb2.get_values(), ob2.get_values(), rtol=1e-5, atol=1e-6 if n >= 1000 else 1e-8)


def _scalar_chain_with_wide_constraints(n, seed):
    """n scalar variables in a chain of unit 'between' rows plus priors, and 2-row / 3-row constraint factors that span
    variables far apart: a front with ONE frontal scalar meets several constraint rows, so rows must be handed to the
    parent clique (the staggered rows of Constrained::QR, NoiseModel.cpp:540-563)."""
    rng = np.random.default_rng(seed)
    fg = GaussianFactorGraph()
    one = np.eye(1)
    for i in range(n):
        fg.add(JacobianFactor(i, one * rng.uniform(0.5, 1.5), [rng.normal()], noiseModel.Unit.Create(1)))
    for i in range(n - 1):
        fg.add(JacobianFactor(i, -one, i + 1, one, [rng.normal()], noiseModel.Isotropic.Sigma(1, 0.5)))
    n_con = 0
    for s in range(0, n - 12, 9):
        a, b, c = s, s + 5, s + 11
        m = 2 if (s // 9) % 2 == 0 else 3
        keys = [a, b, c] if m == 2 else [a, b, c, min(s + 12, n - 1)]
        blocks = [rng.normal(size=(m, 1)) for _ in keys]
        sig = np.zeros(m)
        args = []
        for k, B in zip(keys, blocks):
            args += [k, B]
        fg.add(JacobianFactor(*args, rng.normal(size=m), noiseModel.Constrained.MixedSigmas(sig)))
        n_con += m
    return fg, n_con


@pytest.mark.parametrize("n,seed", [(40, 1), (120, 2), (400, 3)])
@pytest.mark.parametrize("order", ["natural", "reverse", "nd"])
def test_constraint_rows_handed_to_the_parent_clique(gpu, oracle, n, seed, order):
    fg, n_con = _scalar_chain_with_wide_constraints(n, seed)
    arr = linear_arrays(fg)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    if order == "nd":
        ordering = gb.compute_ordering(A.ORDER_ND)
    else:
        ordering = np.arange(n, dtype=np.uint64) if order == "natural" else np.arange(n, dtype=np.uint64)[::-1].copy()
    gb.set_amalgamation(0.0, 16)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    st = gb.stats()
    assert st["n_constraint_rows"] == n_con
    # at least one clique takes in more rows than it has frontal scalars (so the hand-over path runs)
    parent, fronts = gb.get_tree()
    pos = {int(k): i for i, k in enumerate(ordering)}
    first_front = {}
    for c, (fv, _) in enumerate(fronts):
        for v in fv:
            first_front[v] = c
    own = np.zeros(len(fronts), int)
    for f, r, _ in constraint_rows(arr):
        vs = arr.f_vars[arr.f_key_ptr[f]:arr.f_key_ptr[f + 1]]
        own[first_front[min(vs, key=lambda v: pos[int(arr.var_keys[v])])]] += 1
    assert any(own[c] > len(fronts[c][0]) for c in range(len(fronts)))
    gb.linearize()
    ob.linearize()
    rows = constraint_rows(arr)
    for lam in (0.0, 0.3):
        dg, do = gb.solve(lam, False), ob.solve(lam, False)
        dk = dense_kkt_step(arr, ob.jacobians(), rows, lam, np.ones(n))
        assert np.linalg.norm(do - dk) <= 1e-9 * np.linalg.norm(dk)
        assert np.linalg.norm(dg - dk) <= 1e-9 * np.linalg.norm(dk), (lam, np.linalg.norm(dg - dk))


def test_redundant_and_unsupported(gpu):
    """The same hard prior twice: the second copy's rows reduce to 0 = 0 and are dropped (Constrained::QR leaves them without
    a pivot).  Marginals of a constrained problem are refused, not wrong."""
    fg = GaussianFactorGraph()
    I = np.eye(2)
    fg.add(JacobianFactor(0, I, [1.0, -1.0], noiseModel.Constrained.All(2)))
    fg.add(JacobianFactor(0, I, [1.0, -1.0], noiseModel.Constrained.All(2)))
    fg.add(JacobianFactor(0, -I, 1, I, [0.5, 0.5], noiseModel.Unit.Create(2)))
    be = gpu.product_backend(linear_arrays(fg))
    be.set_ordering([0, 1])
    be.linearize()
    assert np.allclose(be.solve(0.0, False), [1.0, -1.0, 1.5, -0.5], atol=1e-12)
    with pytest.raises(A.GsxError):
        be.marginal_covariance(0)
    with pytest.raises(A.GsxError):
        be.dogleg_optimize()


@pytest.mark.parametrize("kind,n", [("pose2", 500), ("pose3", 400)])
def test_partial_reelimination_with_constraints_is_bit_identical(gpu, kind, n):
    """gsx_relinearize_partial on a graph with constraint rows: the cliques that are redone include constrained ones (their
    rows re-read from the new [A b], a clean descendant's leftover rows still in place) — bit for bit the full path."""
    arr = _pose_problem(kind, n, seed=11)
    P, F = gpu.product_backend(arr), gpu.product_backend(arr)
    ordering = P.compute_ordering(A.ORDER_ND)
    for be in (P, F):
        be.set_amalgamation(0.5, 32)
        be.set_ordering(ordering)
    assert P.stats()["n_constrained_fronts"] >= 1
    P.linearize()
    P.solve(0.0, False)
    F.linearize()
    F.solve(0.0, False)
    F.retract(None, commit=True)
    x0, x1 = arr.values.copy(), F.get_values()
    off = np.concatenate([[0], np.cumsum(arr.state_dims())])
    current = x0.copy()
    rng = np.random.default_rng(3)
    con_vars = sorted({int(v) for f, _, _ in constraint_rows(arr) for v in arr.f_vars[arr.f_key_ptr[f]:arr.f_key_ptr[f + 1]]})
    for round_, idx in enumerate([np.array(con_vars[:2]), np.sort(rng.choice(arr.n_vars, arr.n_vars // 30, replace=False)),
                                  np.arange(arr.n_vars - 5, arr.n_vars)]):
        states = np.concatenate([x1[off[i]:off[i + 1]] for i in idx])
        for i in idx:
            current[off[i]:off[i + 1]] = x1[off[i]:off[i + 1]]
        stats = P.relinearize_partial(arr.var_keys[idx], states)
        assert 0 < stats["n_fronts_reeliminated"] <= stats["n_fronts"]
        dp = P.solve(0.0, False)
        F.set_values(current)
        F.linearize()
        df = F.solve(0.0, False)
        assert np.array_equal(P.jacobians(), F.jacobians()), round_
        assert np.array_equal(dp, df), (round_, float(np.max(np.abs(dp - df))))


def test_linear_seam_on_a_kept_handle_with_constraints(gpu, oracle):
    """gsx_solve_gfg_h (the NonlinearOptimizer::solve seam) on a graph with constraint rows: the structure — including which
    rows are constraints — is analysed once, every call hands over new [A b] numbers."""
    from tests.test_oracle_golden import constrained_linear_graphs
    name, fg, expected = constrained_linear_graphs()[2]   # createMultiConstraintGraph
    arrays = linear_arrays(fg)
    be = gpu.product_backend(arrays)
    be.set_ordering([0, 1, 2])
    x = be.solve_gfg_h(None)
    off = arrays.tangent_offsets()
    for k, v in expected.items():
        assert np.allclose(x[off[k]:off[k + 1]], v, atol=1e-9)
    rng = np.random.default_rng(4)
    for trial in range(3):
        fg2 = GaussianFactorGraph()
        blocks = []
        for f in fg.factors:
            m = f.rows
            Ab = f.meas.reshape(-1, m).T + 0.3 * rng.normal(size=(m, f.meas.size // m))
            blocks.append(Ab)
            args = []
            c = 0
            for k, d in zip(f.keys_, f.block_dims):
                args += [k, Ab[:, c:c + d]]
                c += d
            fg2.add(JacobianFactor(*args, Ab[:, -1], f.noise) if f.noise is not None else JacobianFactor(*args, Ab[:, -1]))
        want = fg2.optimize([0, 1, 2], backend_factory=oracle.oracle_backend)
        got = be.solve_gfg_h(blocks)
        for k, v in want.items():
            assert np.allclose(got[off[k]:off[k + 1]], v, rtol=1e-7, atol=1e-9), (trial, k)


@pytest.mark.parametrize("seed", range(96))
def test_constraint_fuzz(gpu, oracle, seed):
    """Random pose graphs, random factors made constraints (all rows or a random subset, random mu), random ordering and
    amalgamation: the damped and undamped steps against the oracle's QR path."""
    rng = np.random.default_rng(1000 + seed + FUZZ_OFFSET)
    kind = "pose2" if seed % 2 == 0 else "pose3"
    d = 3 if kind == "pose2" else 6
    n = int(rng.integers(15, 160))
    arr = (datasets.synth_manhattan_pose2 if kind == "pose2" else datasets.synth_manhattan_pose3)(n, seed=int(rng.integers(1, 1000)))
    cand = np.flatnonzero((arr.f_type == A.F_PRIOR) | (arr.f_type == A.F_BETWEEN))
    chosen = rng.choice(cand, size=int(rng.integers(1, 5)), replace=False)
    models = {}
    for f in chosen:
        sig = rng.uniform(0.05, 0.5, d)
        rows = rng.random(d) < 0.5
        if not rows.any() or rng.random() < 0.4:
            rows[:] = True
        sig[rows] = 0.0
        models[int(f)] = (sig, float(rng.choice([1.0, 50.0, 1000.0]))) if rng.random() < 0.5 else sig
    arr = with_constraints(arr, models)
    rows = constraint_rows(arr)
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    which = int(rng.integers(0, 4))
    if which == 0:
        ordering = arr.var_keys.copy()
    elif which == 1:
        ordering = arr.var_keys[::-1].copy()
    else:
        ordering = gb.compute_ordering(A.ORDER_MINDEGREE if which == 2 else A.ORDER_ND)
    gb.set_amalgamation(float(rng.choice([0.0, 0.5, 2.0])), int(rng.choice([16, 64, 128])))
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    assert gb.stats()["n_constraint_rows"] == len(rows)
    gb.linearize()
    ob.linearize()
    for lam, diag in [(0.0, False), (1e-2, False), (1e-1, True)]:
        dg, do = gb.solve(lam, diag), ob.solve(lam, diag)
        assert np.linalg.norm(dg - do) <= 1e-7 * np.linalg.norm(do), (seed, lam, diag, np.linalg.norm(dg - do) / np.linalg.norm(do))
        assert np.allclose(gb.linear_error(), ob.linear_error(), rtol=1e-6), (seed, lam)


@pytest.mark.parametrize("order", [A.ORDER_SCHUR, A.ORDER_SCHUR_ND, A.ORDER_MINDEGREE])
@pytest.mark.parametrize("relax", [0.0, 0.25])
def test_bundle_adjustment_with_a_hard_gauge(gpu, oracle, order, relax):
    """BAL-shaped problem whose gauge is fixed by HARD priors (Constrained::All) on the first camera and the first point
    instead of soft ones (SFMExample_bal.cpp:61-67 adds soft ones): the constrained cliques are a camera front with lean
    landmark children and a landmark clique that would otherwise be a lean leaf."""
    arr = datasets.synth_bal_arrays(12, 150, 700, seed=3, long_range=0.3, priors=True)
    pri = np.flatnonzero(arr.f_type == A.F_PRIOR)
    assert len(pri) == 2
    arr = with_constraints(arr, {int(pri[0]): np.zeros(9), int(pri[1]): (np.zeros(3), 10.0)})
    rows = constraint_rows(arr)
    assert len(rows) == 12
    gb, ob = gpu.product_backend(arr), oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(order)
    gb.set_amalgamation(relax, 128)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    assert gb.stats()["n_constrained_fronts"] == 2
    assert abs(gb.error() - ob.error()) <= 1e-11 * abs(ob.error())
    gb.linearize()
    ob.linearize()
    for lam, diag in [(1e-4, False), (1.0, False), (1e-3, True)]:
        dg, do = gb.solve(lam, diag), ob.solve(lam, diag)
        # (a bundle adjustment damped by 1e-4 has a condition number near 1e9: both eliminations are compared with the
        #  dense KKT solution, each within what that conditioning allows, and with each other)
        tol = 1e-8 if lam >= 1.0 else 2e-6
        if not diag:
            dk = dense_kkt_step(arr, ob.jacobians(), rows, lam, np.ones(int(arr.var_dims.sum())))
            assert np.linalg.norm(dg - dk) <= tol * np.linalg.norm(dk), (lam, np.linalg.norm(dg - dk) / np.linalg.norm(dk))
            assert np.linalg.norm(do - dk) <= tol * np.linalg.norm(dk), (lam, np.linalg.norm(do - dk) / np.linalg.norm(dk))
        assert np.linalg.norm(dg - do) <= tol * np.linalg.norm(do), (lam, diag, np.linalg.norm(dg - do) / np.linalg.norm(do))
        toff = arr.tangent_offsets()
        for f, r, mu in rows:   # the hard-prior variables do not move
            v = int(arr.f_vars[arr.f_key_ptr[f]])
            assert abs(dg[toff[v] + r] - do[toff[v] + r]) <= 1e-10
    p = A.lm_params_legacy()
    p.max_iterations = 8
    gb2, ob2 = gpu.product_backend(arr), oracle.oracle_backend(arr)
    gb2.set_amalgamation(relax, 128)
    gb2.set_ordering(ordering)
    ob2.set_ordering(ordering)
    rg, ro = gb2.lm_optimize(p), ob2.lm_optimize(p)
    assert rg["iterations"] == ro["iterations"] and np.array_equal(rg["trace_accepted"], ro["trace_accepted"])
    assert abs(rg["final_error"] - ro["final_error"]) <= 1e-6 * max(ro["final_error"], 1e-12)
