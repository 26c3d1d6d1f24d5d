"""The boundary pieces a GTSAM-side shim needs beyond the whole-graph entry points:
  * gsx_get_conditional — per-clique [R S d] read-back, against the oracle's GaussianConditionals
    (gtsam/linear/GaussianConditional.h:243-252);
  * gsx_set_block_jacobians — the S5 fallback (gtsam/nonlinear/NonlinearFactor.h:145-146): factor types the backend
    does not know are linearized on the CPU and uploaded as [A b] blocks at every linearization point;
  * gsx_solve_gfg_h — the NonlinearOptimizer::solve seam (gtsam/nonlinear/NonlinearOptimizer.h:129-130) on a kept handle."""
import ctypes as C

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


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


CASES = {
    "bal": (lambda: datasets.synth_bal_arrays(25, 1500, 7000, seed=11, long_range=0.3), A.ORDER_SCHUR_ND),     # lean leaves + blocked fronts
    "pose3": (lambda: datasets.synth_manhattan_pose3(2500, seed=12), A.ORDER_ND),                              # LDS fronts
    "pose2_mindeg": (lambda: datasets.synth_manhattan_pose2(1500, seed=13), A.ORDER_MINDEGREE),
}


@pytest.mark.parametrize("name", list(CASES))
def test_conditionals_match_the_oracle_clique_by_clique(gpu, oracle, name):
    build, kind = CASES[name]
    arr = build()
    gb = gpu.product_backend(arr)
    gb.set_amalgamation(0.0, 128)      # the reference's cliques
    ob = oracle.oracle_backend(arr)
    ordering = gb.compute_ordering(kind)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    gb.linearize()
    ob.linearize()
    lam = 1e-4
    gb.solve(lam, False)
    ob.solve(lam, False)
    pg, fg = gb.get_tree()
    po, fo = ob.get_tree()
    by_frontals = {tuple(f): c for c, (f, s) in enumerate(fo)}
    dims = arr.var_dims
    rng = np.random.default_rng(3)
    # the root, the largest cliques, and a random sample
    sizes = np.array([dims[f].sum() + (dims[s].sum() if len(s) else 0) for f, s in fg])
    pick = set(np.argsort(-sizes)[:6].tolist()) | set(rng.integers(0, len(fg), 40).tolist()) | {int(np.where(np.asarray(pg) < 0)[0][0])}
    checked = 0
    for c in sorted(pick):
        f, s = fg[c]
        co = by_frontals[tuple(f)]          # same clique in the oracle's (post-order) numbering
        assert sorted(fo[co][1]) == sorted(s)
        Rg, Ro = gb.conditional(c), ob.conditional(co)
        assert Rg.shape == Ro.shape
        # columns: frontals (same order: elimination order), separator variables in each side's own order, rhs
        nf = int(dims[f].sum())
        def sep_cols(sep):
            out, col = {}, nf
            for v in sep:
                out[v] = (col, col + int(dims[v]))
                col += int(dims[v])
            return out
        cg, co_cols = sep_cols(s), sep_cols(fo[co][1])
        scale = max(np.abs(Ro).max(), 1e-300)
        assert np.abs(Rg[:, :nf] - Ro[:, :nf]).max() <= 1e-9 * scale, (name, c)
        for v in s:
            assert np.abs(Rg[:, cg[v][0]:cg[v][1]] - Ro[:, co_cols[v][0]:co_cols[v][1]]).max() <= 1e-9 * scale, (name, c, v)
        assert np.abs(Rg[:, -1] - Ro[:, -1]).max() <= 1e-9 * scale, (name, c)
        assert np.all(np.tril(Rg[:, :nf], -1) == 0)   # R is upper triangular
        checked += 1
    assert checked >= 20
    with pytest.raises(A.GsxError):
        gb.conditional(len(fg))


def _split_fallback(arr, every=3):
    """The Between factors with index % every == 0 become 'unknown to the backend': GSX_F_LINEAR slots at the end of the
    product's factor list, and a graph of their own for the CPU side."""
    nf = arr.n_factors
    is_fb = np.array([arr.f_type[f] == A.F_BETWEEN and f % every == 0 for f in range(nf)])

    def subset(mask):
        keep = np.where(mask)[0]
        kp, mp, npx = [0], [0], [0]
        fv, meas, noise = [], [], []
        for f in keep:
            fv.extend(arr.f_vars[arr.f_key_ptr[f]:arr.f_key_ptr[f + 1]].tolist())
            kp.append(len(fv))
            meas.append(arr.meas[arr.f_meas_ptr[f]:arr.f_meas_ptr[f + 1]])
            mp.append(mp[-1] + meas[-1].size)
            noise.append(arr.noise[arr.f_noise_ptr[f]:arr.f_noise_ptr[f + 1]])
            npx.append(npx[-1] + noise[-1].size)
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0)
        return A.ProblemArrays(arr.var_keys, arr.var_types, arr.var_dims, arr.f_type[keep], arr.f_rows[keep],
                               np.array(kp, np.int32), np.array(fv, np.int32), np.array(mp, np.int64), cat(meas),
                               arr.f_noise_kind[keep], np.array(npx, np.int64), cat(noise), arr.values.copy(), dict(arr.meta))
    native, fallback = subset(~is_fb), subset(is_fb)
    mixed = native
    first_slot = native.n_factors
    for f in range(fallback.n_factors):
        vs = fallback.f_vars[fallback.f_key_ptr[f]:fallback.f_key_ptr[f + 1]]
        m = int(fallback.f_rows[f])
        ncols = int(arr.var_dims[vs].sum()) + 1
        mixed = mixed.with_factor(A.F_LINEAR, vs.tolist(), m, np.zeros(m * ncols), A.NOISE_UNIT)
    return mixed, fallback, first_slot


def _fallback_blocks(of, fallback, arr):
    """CPU linearization of the fallback factors at the oracle handle's values: their whitened [A b] blocks."""
    of.linearize()
    jac = of.jacobians()
    blocks, off = [], 0
    for f in range(fallback.n_factors):
        vs = fallback.f_vars[fallback.f_key_ptr[f]:fallback.f_key_ptr[f + 1]]
        m = int(fallback.f_rows[f])
        ncols = int(arr.var_dims[vs].sum()) + 1
        blocks.append(jac[off:off + m * ncols].reshape(ncols, m).T)
        off += m * ncols
    return blocks


class _LmState(C.Structure):
    _fields_ = [("lam", C.c_double), ("factor", C.c_double), ("cost", C.c_double), ("outer", C.c_int32), ("inner", C.c_int32)]


class _LmDecision(C.Structure):
    _fields_ = [("verdict", C.c_int32), ("solved", C.c_int32), ("gain_ratio", C.c_double), ("cost_change", C.c_double),
                ("trial_cost", C.c_double), ("lambda_tried", C.c_double)]


@pytest.mark.parametrize("kind", ["pose2", "pose3"])
def test_mixed_native_and_cpu_linearized_graph_through_lm(gpu, oracle, kind):
    """A third of the Between factors are 'unknown' to the backend: linearized on the CPU (here: by the oracle, standing in
    for factor->linearize(values)) and uploaded with gsx_set_block_jacobians at every linearization point, their nonlinear
    error added by the caller.  The LM run — the product's kernels and its decision function — must be the run the oracle
    makes on the all-native graph."""
    arr = datasets.synth_manhattan_pose2(400, seed=21) if kind == "pose2" else datasets.synth_manhattan_pose3(300, seed=22)
    mixed, fallback, first_slot = _split_fallback(arr)
    assert fallback.n_factors > 50
    gb = gpu.product_backend(mixed)
    of = oracle.oracle_backend(fallback)            # the CPU side: only the fallback factors
    ob = oracle.oracle_backend(arr)                 # the expected run: everything native
    ordering = gb.compute_ordering(A.ORDER_ND)
    gb.set_ordering(ordering)
    ob.set_ordering(ordering)
    lib = gpu.load()
    lib.gsx_lm_decide.restype = C.c_int32
    p = A.lm_params_legacy()
    p.max_iterations = 8
    expect = ob.lm_optimize(p)
    X = gb.get_values()
    of.set_values(X)
    st = _LmState(p.lambda_initial, p.lambda_factor, gb.error() + of.error(), 0, 0)
    assert abs(st.cost - expect["initial_error"]) <= 1e-10 * expect["initial_error"]
    trace = []
    while st.outer < p.max_iterations:
        before = st.cost
        of.set_values(X)
        gb.set_block_jacobians(first_slot, _fallback_blocks(of, fallback, arr))
        gb.linearize()
        while True:
            lam = st.lam
            delta = gb.solve(lam, bool(p.diagonal_damping), p.min_diagonal, p.max_diagonal)
            lin0, lind = gb.linear_error()
            trial = gb.retract(None, commit=False)          # native factors at the trial point (device)
            of.set_values(X)
            trial += of.retract(delta, commit=False)        # + the fallback factors at the same point (CPU)
            d = _LmDecision()
            lib.gsx_lm_decide(C.byref(p), C.byref(st), C.c_int32(1), C.c_double(lin0), C.c_double(lind), C.c_double(trial),
                              C.byref(d))
            trace.append((d.trial_cost, lam, int(d.verdict == 1)))
            if d.verdict == 1:
                gb.retract(None, commit=True, want_error=False)
                X = gb.get_values()
            if d.verdict != 0:
                break
        dec = before - st.cost
        if d.verdict == 3 or (p.relative_error_tol and dec / before <= p.relative_error_tol) or dec <= p.absolute_error_tol:
            break
    n = len(expect["trace_accepted"])
    assert len(trace) == n and n >= 3
    assert [t[2] for t in trace] == expect["trace_accepted"].tolist()
    assert np.allclose([t[1] for t in trace], expect["trace_lambda"], rtol=1e-9)
    assert np.allclose([t[0] for t in trace], expect["trace_error"], rtol=1e-8)
    assert relerr(X, ob.get_values()) < 1e-8
    assert st.cost < 0.5 * expect["initial_error"]
    # argument checks: a native factor is not a slot; sizes must match
    with pytest.raises(A.GsxError):
        gb.set_block_jacobians(0, [np.zeros((3, 7))])
    with pytest.raises(A.GsxError):
        gb.set_block_jacobians(first_slot, [np.zeros((2, 2))])


def test_linear_seam_on_a_kept_handle(gpu, oracle):
    """gsx_solve_gfg_h: the structure (keys, dims, ordering) is analysed once; every call hands over new numbers only —
    what NonlinearOptimizer::solve sees across the trials of an LM run (same graph, other lambda / linearization)."""
    from tests.test_oracle_golden import small_gaussian_factor_graph, CORRECT_DELTA, L, X
    g = small_gaussian_factor_graph()
    arrays = g.to_arrays(None)
    arrays.values = np.zeros(int(arrays.var_dims.sum()))
    be = gpu.product_backend(arrays)
    be.set_ordering([L(1), X(1), X(2)])
    # the factors' own numbers (tests/testGaussianJunctionTreeB.cpp:113-140)
    x = be.solve_gfg_h(None)
    keys = arrays.var_keys.tolist()
    off = np.concatenate([[0], np.cumsum(arrays.var_dims)])
    for k, v in CORRECT_DELTA.items():
        i = keys.index(k)
        assert np.allclose(x[off[i]:off[i + 1]], v, atol=1e-9)
    # other numbers on the same structure: random well-posed [A b] blocks, against a dense normal-equation solve
    rng = np.random.default_rng(0)
    ntot = int(arrays.var_dims.sum())
    for trial in range(4):
        blocks, Afull, bfull = [], [], []
        for f in range(arrays.n_factors):
            vs = arrays.f_vars[arrays.f_key_ptr[f]:arrays.f_key_ptr[f + 1]]
            m = int(arrays.f_rows[f])
            blk = rng.normal(size=(m, int(arrays.var_dims[vs].sum()) + 1))
            if len(vs) == 1:
                blk[:, :m] += 4 * np.eye(m)        # keeps the system positive definite
            blocks.append(blk)
            row = np.zeros((m, ntot))
            col = 0
            for v in vs:
                row[:, off[v]:off[v + 1]] = blk[:, col:col + arrays.var_dims[v]]
                col += arrays.var_dims[v]
            Afull.append(row)
            bfull.append(blk[:, -1])
        Afull, bfull = np.vstack(Afull), np.concatenate(bfull)
        # noise models of the slots are folded in by the library: this graph's are unit after to_arrays' whitening? use the
        # library itself as the reference for the first call, numpy for the numbers
        got = be.solve_gfg_h(blocks)
        # expected with each factor's noise model folded in (sigma per row)
        W = []
        for f in range(arrays.n_factors):
            m = int(arrays.f_rows[f])
            kind_ = int(arrays.f_noise_kind[f])
            npar = arrays.noise[arrays.f_noise_ptr[f]:arrays.f_noise_ptr[f + 1]]
            if kind_ == A.NOISE_UNIT:
                W.append(np.ones(m))
            elif kind_ == A.NOISE_ISOTROPIC:
                W.append(np.full(m, 1.0 / npar[0]))
            elif kind_ == A.NOISE_DIAGONAL:
                W.append(1.0 / npar)
            else:
                pytest.skip("Gaussian noise in the example graph")
        W = np.concatenate(W)
        exp = np.linalg.lstsq(Afull * W[:, None], bfull * W, rcond=None)[0]
        assert relerr(got, exp) < 1e-9, trial
    st = be.stats()
    assert st["n_fronts"] >= 1
