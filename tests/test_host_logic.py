"""CPU-only checks of the product's host side: the C-ABI library loads and exports exactly what
include/gsx.h declares, the host algorithms (orderings, symbolic analysis = Bayes tree) agree with
the oracle, and the numeric entry points fail loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import gtsam_petercdev_amd as gt
from gtsam_petercdev_amd import _abi as A
from gtsam_petercdev_amd import datasets, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gtsam_petercdev_amd import build
    build.build_lib()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "gsx.h")).read()
    declared = set(re.findall(r"\b(gsx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gsx_status", "gsx_handle"}
    assert len(declared) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (gsx_[a-z0-9_]+)", out))
    assert declared <= exported, sorted(declared - exported)
    assert exported <= declared, f"exported but not declared in include/gsx.h: {sorted(exported - declared)}"


def test_product_does_not_link_or_import_the_oracle():
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    pkg = os.path.join(ROOT, "gtsam_petercdev_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f
                assert "orc_" not in src or f == "_abi.py", f


def _tree_set(be):
    parent, fronts = be.get_tree()
    out = {}
    for c, (f, s) in enumerate(fronts):
        p = parent[c]
        out[tuple(sorted(f))] = (tuple(sorted(s)), tuple(sorted(fronts[p][0])) if p >= 0 else None)
    return out


PROBLEMS = {
    "bal": lambda: datasets.synth_bal_arrays(8, 60, 200, seed=1, long_range=0.3),
    "pose2": lambda: datasets.synth_manhattan_pose2(300, seed=3),
    "pose3": lambda: datasets.synth_manhattan_pose3(200, seed=4),
}


@pytest.mark.parametrize("name", list(PROBLEMS))
def test_bayes_tree_identical_to_reference_construction(lib, oracle, name):
    """The product's symbolic analysis must give the cliques, separators and parents that the
    reference's EliminationTree + JunctionTree constructors give (restated literally in the oracle),
    for its own orderings and for the reference's CCOLAMD ordering."""
    arr = PROBLEMS[name]()
    pb = _lib.ProductBackend(arr, host_only=True)
    pb.set_amalgamation(0.0, 128)   # the reference's cliques (a new handle's default is the library's own amalgamation)
    ob = oracle.oracle_backend(arr)
    orderings = [pb.compute_ordering(k) for k in (A.ORDER_NATURAL, A.ORDER_MINDEGREE, A.ORDER_ND, A.ORDER_SCHUR, A.ORDER_SCHUR_ND)]
    if oracle.have_ref_colamd():
        orderings.append(oracle.colamd_ordering(arr))
    for ordering in orderings:
        assert sorted(ordering.tolist()) == sorted(arr.var_keys.tolist())  # a permutation
        pb.set_ordering(ordering)
        ob.set_ordering(ordering)
        ob.linearize()
        ob.solve(1e-3)
        assert _tree_set(pb) == _tree_set(ob)
        st = pb.stats()
        _, tree = ob.timing()
        assert st["n_fronts"] == tree["cliques"]
        assert abs(st["factor_flops"] - tree["flops"]) <= 1e-9 * tree["flops"]
        assert abs(st["front_bytes"] - tree["bytes"]) <= 1e-9 * tree["bytes"]
        assert st["max_front_dim"] == tree["max_f"]


@pytest.mark.parametrize("name", list(PROBLEMS))
@pytest.mark.parametrize("relax,maxf", [(0.25, 128), (1.0, 64), (50.0, 4096), (A.AMALGAMATION_AUTO, 0)])
def test_relaxed_amalgamation_is_a_coarsening_of_the_reference_tree(lib, name, relax, maxf):
    """gsx_set_amalgamation(relax > 0): every clique is a union of CONNECTED reference cliques, its separator is the
    separator of its topmost member, the parent relation is the quotient of the reference tree; relax = 0 restores
    the reference tree exactly."""
    arr = PROBLEMS[name]()
    pb = _lib.ProductBackend(arr, host_only=True)
    ordering = pb.compute_ordering(A.ORDER_ND)
    pb.set_amalgamation(0.0, 128)
    pb.set_ordering(ordering)
    ref_parent, ref_fronts = pb.get_tree()
    ref_flops = pb.stats()["factor_flops"]
    pb.set_amalgamation(relax, maxf)
    pb.set_ordering(ordering)
    parent, fronts = pb.get_tree()
    assert len(fronts) <= len(ref_fronts) and pb.stats()["factor_flops"] >= ref_flops * (1 - 1e-12)
    var_front = {}
    for c, (f, s) in enumerate(fronts):
        for v in f:
            assert v not in var_front  # the frontal sets partition the variables
            var_front[int(v)] = c
    assert len(var_front) == arr.n_vars
    members = {}
    for rc, (rf, rs) in enumerate(ref_fronts):
        owners = {var_front[int(v)] for v in rf}
        assert len(owners) == 1  # a reference clique is never split
        members.setdefault(owners.pop(), []).append(rc)
    for c, rcs in members.items():
        inside = set(rcs)
        tops = [rc for rc in rcs if ref_parent[rc] not in inside]
        assert len(tops) == 1  # connected: exactly one member whose reference parent lies outside
        top = tops[0]
        assert sorted(map(int, fronts[c][1])) == sorted(map(int, ref_fronts[top][1]))
        rp = ref_parent[top]
        assert parent[c] == (-1 if rp < 0 else var_front[int(ref_fronts[rp][0][0])])
    pb.set_amalgamation(0.0, 128)
    pb.set_ordering(ordering)
    p0, f0 = pb.get_tree()
    assert [sorted(map(int, f)) for f, _ in f0] == [sorted(map(int, f)) for f, _ in ref_fronts]
    assert list(p0) == list(ref_parent)


def test_ordering_errors(lib):
    arr = PROBLEMS["pose2"]()
    pb = _lib.ProductBackend(arr, host_only=True)
    with pytest.raises(gt.GsxError) as ei:
        pb.set_ordering(arr.var_keys[:-1])
    assert ei.value.status == A.GSX_E_BAD_ORDERING
    bad = arr.var_keys.copy()
    bad[0] = bad[1]
    with pytest.raises(gt.GsxError) as ei:
        pb.set_ordering(bad)
    assert ei.value.status == A.GSX_E_BAD_ORDERING
    bad = arr.var_keys.copy()
    bad[0] = 10 ** 9
    with pytest.raises(gt.GsxError):
        pb.set_ordering(bad)


def test_malformed_description_is_rejected(lib):
    arr = PROBLEMS["pose2"]()
    arr.f_rows[0] = 5  # Pose2 between factor must have 3 rows
    with pytest.raises(gt.GsxError) as ei:
        _lib.ProductBackend(arr, host_only=True)
    assert ei.value.status == A.GSX_E_INVALID


def test_zero_sigmas_are_hard_constraints_or_refused(lib):
    """A zero sigma in a DIAGONAL model is the reference's noiseModel::Constrained (gtsam/linear/NoiseModel.h:389-500,
    Diagonal::Sigmas :292-309): accepted, counted, and its clique scheduled as a constrained (blocked) front.  A zero or
    negative sigma anywhere else — an Isotropic model, a negative entry, a non-positive mu — is refused at gsx_create
    instead of turning into infinities in the whitening."""
    arr = PROBLEMS["pose2"]()
    prior = int(np.flatnonzero(arr.f_type == A.F_PRIOR)[0])
    assert (int(arr.f_noise_kind[prior]) & A.NOISE_BASE_MASK) == A.NOISE_DIAGONAL
    arr.noise[int(arr.f_noise_ptr[prior]):int(arr.f_noise_ptr[prior + 1])] = 0.0   # Constrained::All(3) on the first pose
    pb = _lib.ProductBackend(arr, host_only=True)
    pb.set_amalgamation(0.0, 128)
    for kind in (A.ORDER_NATURAL, A.ORDER_MINDEGREE, A.ORDER_ND):
        pb.set_ordering(pb.compute_ordering(kind))
        st = pb.stats()
        # three rows, all absorbed in the clique of the first pose: no other front is touched
        assert st["n_constraint_rows"] == 3 and st["n_constrained_fronts"] == 1
        _, fronts = pb.get_tree()
        v0 = int(arr.f_vars[arr.f_key_ptr[prior]])
        c = next(i for i, (fv, _) in enumerate(fronts) if v0 in fv)
        assert pb.front_classes()[c] == 2
    arr = PROBLEMS["bal"]() if "bal" in PROBLEMS else PROBLEMS[next(iter(PROBLEMS))]()
    iso = [i for i in range(arr.n_factors) if (int(arr.f_noise_kind[i]) & A.NOISE_BASE_MASK) == A.NOISE_ISOTROPIC]
    assert iso
    arr.noise[int(arr.f_noise_ptr[iso[0]])] = 0.0
    with pytest.raises(gt.GsxError) as ei:
        _lib.ProductBackend(arr, host_only=True)
    assert ei.value.status == A.GSX_E_INVALID
    arr = PROBLEMS["pose2"]()
    arr.noise[int(arr.f_noise_ptr[prior])] = -1.0
    with pytest.raises(gt.GsxError):
        _lib.ProductBackend(arr, host_only=True)


def test_leftover_constraint_rows_go_to_the_front_of_their_first_variable(lib):
    """Scalar chain; a 2-row constraint on (x0, x5, x11): the clique of x0 has one frontal scalar, so one row is a pivot
    there and the other waits for x5 — in the clique where x5 is frontal, not in every clique in between."""
    from gtsam_petercdev_amd.graph import GaussianFactorGraph, JacobianFactor, noiseModel
    n = 16
    fg = GaussianFactorGraph()
    one = np.eye(1)
    for i in range(n):
        fg.add(JacobianFactor(i, one, [0.0], noiseModel.Unit.Create(1)))
    for i in range(n - 1):
        fg.add(JacobianFactor(i, -one, i + 1, one, [1.0], noiseModel.Unit.Create(1)))
    fg.add(JacobianFactor(0, [[1.0], [2.0]], 5, [[1.0], [-1.0]], 11, [[3.0], [1.0]], [1.0, 2.0], noiseModel.Constrained.All(2)))
    arr = fg.to_arrays(None)
    arr.values = np.zeros(n)
    pb = _lib.ProductBackend(arr, host_only=True)
    pb.set_amalgamation(0.0, 128)
    pb.set_ordering(np.arange(n, dtype=np.uint64))
    st = pb.stats()
    assert st["n_constraint_rows"] == 2 and st["n_constrained_fronts"] == 2
    _, fronts = pb.get_tree()
    cls = pb.front_classes()
    con = [i for i in range(len(fronts)) if cls[i] == 2]
    assert len(con) == 2 and 0 in fronts[con[0]][0] and 5 in fronts[con[1]][0]


def test_schur_ordering_puts_landmarks_first(lib):
    arr = PROBLEMS["bal"]()
    pb = _lib.ProductBackend(arr, host_only=True)
    o = pb.compute_ordering(A.ORDER_SCHUR)
    idx = {int(k): i for i, k in enumerate(arr.var_keys)}
    types = [int(arr.var_types[idx[int(k)]]) for k in o]
    n_pts = arr.meta["n_points"]
    assert all(t == A.VAR_VECTOR for t in types[:n_pts]) and all(t == A.VAR_CAMERA for t in types[n_pts:])


def test_schur_orderings_know_pose_landmark_graphs(lib):
    """Visual SLAM with a fixed calibration (GenericProjectionFactor<Pose3, Point3>: SURVEY §8 config 5) has the same
    bipartite shape as SFM: its landmarks go first too, and the nested dissection of the reduced pose graph keeps the
    tree of a 600-keyframe street shallow (minimum degree over everything gives a chain as long as the street)."""
    arr, ids = datasets.synth_visual_slam(600, 20000, 90000, seed=4)
    assert len(set(ids.tolist())) == arr.n_factors
    pb = _lib.ProductBackend(arr, host_only=True)
    idx = {int(k): i for i, k in enumerate(arr.var_keys)}
    for kind in (A.ORDER_SCHUR, A.ORDER_SCHUR_ND):
        o = pb.compute_ordering(kind)
        types = [int(arr.var_types[idx[int(k)]]) for k in o]
        n_l = arr.meta["n_landmarks"]
        assert all(t == A.VAR_VECTOR for t in types[:n_l]) and all(t == A.VAR_POSE3 for t in types[n_l:])
    pb.set_ordering(pb.compute_ordering(A.ORDER_SCHUR_ND))
    assert pb.stats()["n_levels"] <= 12
    # growth by one keyframe: the smaller graph's factors keep their ids
    a1, id1 = datasets.synth_visual_slam(600, 20000, 90000, seed=4, upto=599)
    assert set(id1.tolist()) <= set(ids.tolist()) and a1.n_vars < arr.n_vars


@pytest.mark.skipif(_lib.device_count() > 0, reason="only meaningful without a GPU")
def test_no_gpu_means_loud_failure_not_fallback(lib):
    arr = PROBLEMS["pose2"]()
    with pytest.raises(gt.GsxError) as ei:
        _lib.ProductBackend(arr)  # uploads the initial values -> needs a device
    assert ei.value.status == A.GSX_E_NO_DEVICE
    pb = _lib.ProductBackend(arr, host_only=True)
    for call in (pb.error, pb.linearize, lambda: pb.solve(0.0), pb.get_values):
        with pytest.raises(gt.GsxError) as ei:
            call()
        assert ei.value.status in (A.GSX_E_NO_DEVICE, A.GSX_E_STATE)
    ok = C.c_int32()
    m = np.eye(3)
    st = lib.gsx_cholesky_partial(m.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(3), C.c_int32(2), C.c_int32(0),
                                  C.byref(ok))
    assert st == A.GSX_E_NO_DEVICE


def test_datasets_roundtrip_g2o(tmp_path, golden_dir):
    """g2o reader/writer (gtsam/slam/dataset.cpp): read the reference's example files, write, re-read."""
    for fname, is3d in (("noisyToyGraph.txt", False), ("pose3example.txt", True)):
        arr = datasets.read_g2o(os.path.join(golden_dir, fname), is3D=is3d)
        assert arr.n_vars > 0 and arr.n_factors > arr.n_vars - 1
        p = tmp_path / ("rt_" + fname)
        datasets.write_g2o(str(p), arr, arr.values)
        arr2 = datasets.read_g2o(str(p), is3D=is3d)
        assert np.array_equal(arr.var_keys, arr2.var_keys)
        assert np.allclose(arr.values, arr2.values, atol=1e-12)
        assert np.allclose(arr.meas, arr2.meas, atol=1e-9)


@pytest.mark.parametrize("reader", ["python", "native"])
def test_pose3example_oracle_gn_converges(oracle, golden_dir, reader):
    """examples/Pose3SLAMExample_g2o.cpp on examples/Data/pose3example.txt (oracle, CPU), read by the Python reader
    and by the native one (gsx_read_g2o): the reference's printed errors."""
    path = os.path.join(golden_dir, "pose3example.txt")
    arr = datasets.read_g2o(path, is3D=True) if reader == "python" else _lib.read_g2o(path, True)
    ob = oracle.oracle_backend(arr)
    ob.set_ordering(arr.var_keys if not oracle.have_ref_colamd() else oracle.colamd_ordering(arr))
    r = ob.gn_optimize(100)
    # SURVEY.md §6.2 records what the reference's own example binary prints for this file:
    # "Pose3SLAMExample_g2o pose3example.txt  64 941.32 -> 19 130.66" (Gauss-Newton, default params)
    assert abs(r["initial_error"] - 64941.32) < 0.01
    assert abs(r["final_error"] - 19130.66) < 0.01


# ---- native readers (csrc/io.cpp) against the Python readers on the reference's own data files -------------------
def _same_arrays(a, b, noise_tol=1e-12):
    for n in ("var_keys", "var_types", "var_dims", "f_type", "f_rows", "f_key_ptr", "f_vars", "f_meas_ptr",
              "f_noise_kind", "f_noise_ptr"):
        assert np.array_equal(getattr(a, n), getattr(b, n)), n
    assert np.allclose(a.meas, b.meas, rtol=0, atol=1e-15)
    assert np.allclose(a.noise, b.noise, rtol=noise_tol, atol=1e-15)
    assert np.allclose(a.values, b.values, rtol=0, atol=1e-15)


@pytest.mark.parametrize("name,is3d", [("pose2example.txt", False), ("noisyToyGraph.txt", False),
                                       ("pose3example.txt", True)])
def test_native_g2o_reader_matches_python_reader(lib, golden_dir, name, is3d):
    path = os.path.join(golden_dir, name)
    _same_arrays(_lib.read_g2o(path, is3d), datasets.read_g2o(path, is3D=is3d))


@pytest.mark.parametrize("priors", [False, True])
def test_native_bal_reader_matches_python_reader(lib, golden_dir, priors):
    path = os.path.join(golden_dir, "dubrovnik-3-7-pre.txt")
    py = datasets.bal_arrays(datasets.read_bal(path), priors=priors)
    _same_arrays(_lib.read_bal(path, priors), py)


def test_native_readers_reject_missing_and_malformed_files(lib, tmp_path):
    with pytest.raises(A.GsxError):
        _lib.read_g2o(str(tmp_path / "nope.g2o"))
    bad = tmp_path / "bad.txt"
    bad.write_text("3 7 12\n0 0 1.0\n")  # truncated BAL
    with pytest.raises(A.GsxError):
        _lib.read_bal(str(bad))
    bad.write_text("EDGE_SE2 0 1 1.0 0.0\n")  # truncated edge
    with pytest.raises(A.GsxError):
        _lib.read_g2o(str(bad))


@pytest.mark.parametrize("name,is3d", [("pose2example.txt", False), ("pose3example.txt", True)])
def test_native_g2o_writer_round_trips(lib, golden_dir, tmp_path, name, is3d):
    """gsx_write_g2o -> gsx_read_g2o gives back the same problem (states to 1e-15, information factors to 1e-9), and
    the file parses to the same problem as the one the Python writer produces."""
    arr = _lib.read_g2o(os.path.join(golden_dir, name), is3d)
    out = tmp_path / "native.g2o"
    _lib.write_g2o(str(out), arr, arr.values)
    back = _lib.read_g2o(str(out), is3d)
    for n in ("var_keys", "var_types", "var_dims", "f_type", "f_rows", "f_key_ptr", "f_vars", "f_noise_kind"):
        assert np.array_equal(getattr(arr, n), getattr(back, n)), n
    assert np.allclose(arr.values, back.values, rtol=0, atol=1e-14)
    assert np.allclose(arr.meas, back.meas, rtol=0, atol=1e-14)
    assert np.allclose(arr.noise, back.noise, rtol=1e-9, atol=1e-12)
    py = tmp_path / "python.g2o"
    datasets.write_g2o(str(py), arr, arr.values)
    _same_arrays(back, _lib.read_g2o(str(py), is3d), noise_tol=1e-9)


# ---- TORO / "graph" formats (gsx_load2d, VERTEX3 / EDGE3) -------------------------------------------------------------
def test_load2D_w100(golden_dir):
    """gtsam/slam/tests/testDataset.cpp:90-101 (dataSet, load2D): 300 factors, 100 values, the first factor
    BetweenFactor<Pose2>(1, 0, Pose2(-0.99879, 0.0417574, -0.00818381), Unit(3)) — the file's covariance in TORO layout
    is the identity, which the smart model turns into Unit."""
    arr = _lib.load2d(os.path.join(golden_dir, "w100.graph"))
    assert arr.n_factors == 300 and arr.n_vars == 100
    assert np.all(arr.f_type == A.F_BETWEEN) and np.all(arr.var_types == A.VAR_POSE2)
    assert arr.var_keys[arr.f_vars[:2]].tolist() == [1, 0]
    assert np.allclose(arr.meas[:3], [-0.99879, 0.0417574, -0.00818381], atol=0)
    assert arr.f_noise_kind[0] == A.NOISE_UNIT and arr.f_noise_ptr[1] == 0
    # parseVariables<Pose2>: the poses of the file (VERTEX2 1 0.995595 0.0837204 0.0146728 is the second line)
    assert np.allclose(arr.values[3:6], [0.995595, 0.0837204, 0.0146728], atol=0)
    # the model of examples/Pose2SLAMExample_graph.cpp:38-40 replaces the file's
    sig = [0.05, 0.05, 5.0 * np.pi / 180.0]
    arr2 = _lib.load2d(os.path.join(golden_dir, "w100.graph"), model_sigmas=sig)
    assert np.all(arr2.f_noise_kind == A.NOISE_DIAGONAL) and np.allclose(arr2.noise[:3], sig, atol=0)
    # maxIndex (dataset.cpp:364-366): edges touching a pose above it are dropped
    arr3 = _lib.load2d(os.path.join(golden_dir, "w100.graph"), max_index=5)
    assert arr3.n_vars == 6 and np.all(arr3.var_keys[arr3.f_vars] <= 5) and 5 <= arr3.n_factors < 300


def test_load2D_noise_formats(tmp_path):
    """createNoiseModel (dataset.cpp:216-296): the four layouts, the AUTO guess, smart models, robust kernels."""
    f = tmp_path / "t.graph"
    v = "4 0 9 16 0 0"   # TORO layout: ff fs ss rr fr sr -> diag(4, 9, 16)
    f.write_text(f"VERTEX2 0 0 0 0\nVERTEX2 1 1 0 0\nEDGE2 0 1 1 0 0 {v}\n")
    a = _lib.load2d(str(f))                                   # AUTO -> GRAPH: a covariance
    assert a.f_noise_kind[0] == A.NOISE_DIAGONAL and np.allclose(a.noise[:3], [2, 3, 4])
    a = _lib.load2d(str(f), noise_format=A.NOISE_FORMAT_TORO)  # the same numbers as information
    assert a.f_noise_kind[0] == A.NOISE_DIAGONAL and np.allclose(a.noise[:3], [1 / 2, 1 / 3, 1 / 4])
    a = _lib.load2d(str(f), noise_format=A.NOISE_FORMAT_TORO, smart=False)
    assert a.f_noise_kind[0] == A.NOISE_GAUSSIAN and np.allclose(a.noise[:9].reshape(3, 3), np.diag([2.0, 3, 4]))
    a = _lib.load2d(str(f), kernel=1)
    assert a.f_noise_kind[0] == (A.NOISE_DIAGONAL | A.NOISE_ROBUST_HUBER) and np.allclose(a.noise[:4], [2, 3, 4, 1.345])
    f.write_text("EDGE2 0 1 1 0 0 2 0.5 0 3 0 4\n")          # G2O layout I11 I12 I13 I22 I23 I33, not diagonal
    with pytest.raises(A.GsxError):
        _lib.load2d(str(f))                                   # AUTO cannot tell
    a = _lib.load2d(str(f), noise_format=A.NOISE_FORMAT_G2O)
    R = a.noise[:9].reshape(3, 3)
    assert a.f_noise_kind[0] == A.NOISE_GAUSSIAN and np.allclose(R.T @ R, [[2, 0.5, 0], [0.5, 3, 0], [0, 0, 4]])
    assert a.n_vars == 2 and np.allclose(a.values, [0, 0, 0, 1, 0, 0])   # both poses from the odometry
    a = _lib.load2d(str(f), noise_format=A.NOISE_FORMAT_COV)
    R = a.noise[:9].reshape(3, 3)
    assert np.allclose(np.linalg.inv(R.T @ R), [[2, 0.5, 0], [0.5, 3, 0], [0, 0, 4]])
    f.write_text("EDGE2 0 1 1 0 0 1 0 0 1 0 1\n")            # COV layout, identity
    a = _lib.load2d(str(f))
    assert a.f_noise_kind[0] == A.NOISE_UNIT


def test_load2D_bearing_range(golden_dir):
    """examples/Data/example.graph (matlab/gtsam_examples/PlanarSLAMExample_graph.m): 95 poses, 94 odometry edges, 422 BR
    lines -> BearingRangeFactor<Pose2, Point2> with Diagonal(bearing_std, range_std), landmarks keyed L(j) and created
    from their first sighting (dataset.cpp:452-496, 547-563)."""
    arr = _lib.load2d(os.path.join(golden_dir, "example.graph"))
    nb, nbr = int((arr.f_type == A.F_BETWEEN).sum()), int((arr.f_type == A.F_BEARINGRANGE).sum())
    assert (nb, nbr) == (94, 422) and int((arr.var_types == A.VAR_POSE2).sum()) == 95
    lm = arr.var_types == A.VAR_VECTOR
    assert lm.sum() > 0 and np.all(arr.var_keys[lm] >> np.uint64(56) == ord("l")) and np.all(arr.var_dims[lm] == 2)
    k = int(np.argmax(arr.f_type == A.F_BEARINGRANGE))        # "BR 0 144 0.185182458717 9.13212526936 0.0349 0.1"
    vs = arr.f_vars[arr.f_key_ptr[k]:arr.f_key_ptr[k + 1]]
    assert int(arr.var_keys[vs[0]]) == 0 and int(arr.var_keys[vs[1]]) == (ord("l") << 56) + 144
    assert np.allclose(arr.meas[arr.f_meas_ptr[k]:arr.f_meas_ptr[k + 1]], [0.185182458717, 9.13212526936], atol=0)
    assert arr.f_noise_kind[k] == A.NOISE_DIAGONAL
    assert np.allclose(arr.noise[arr.f_noise_ptr[k]:arr.f_noise_ptr[k + 1]], [0.0349, 0.1], atol=0)
    # landmark 144 sits where pose 0 saw it
    x, y, th = arr.values[:3]
    off = int(np.sum(np.where(arr.var_types[:vs[1]] == A.VAR_POSE2, 3, 2)))
    assert np.allclose(arr.values[off:off + 2], [x + 9.13212526936 * np.cos(th + 0.185182458717),
                                                 y + 9.13212526936 * np.sin(th + 0.185182458717)])


def test_read_toro_3d_sphere2500(golden_dir):
    """examples/Data/sphere2500.txt: 4949 EDGE3 lines and no vertices (process_shonan_timing_results.py:179 gives the
    counts); measurement = Pose3(Rot3::Ypr(yaw, pitch, roll), t), information as written (dataset.cpp:829-840); the
    initial estimate chains the successive odometry from the origin (matlab/+gtsam/load3D.m:21-53)."""
    arr = _lib.read_g2o(os.path.join(golden_dir, "sphere2500.txt"), is3D=True)
    assert arr.n_vars == 2500 and arr.n_factors == 4949 + 1     # + the anchoring prior of the g2o examples
    # EDGE3 0 1 0.341895 -0.0416997 0.0330394 -0.00305942 0.00822248 0.1802  10 0 0 0 0 0 10 ...
    R = arr.meas[:9].reshape(3, 3)
    r, p, y = -0.00305942, 0.00822248, 0.1802
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    assert np.allclose(R, Rz @ Ry @ Rx, atol=1e-15) and np.allclose(arr.meas[9:12], [0.341895, -0.0416997, 0.0330394])
    Rn = arr.noise[:36].reshape(6, 6)
    assert np.allclose(Rn.T @ Rn, np.diag([10.0, 10, 10, 100, 100, 25]))
    # pose 1 = origin * first measurement; every pose got a value with a proper rotation
    assert np.allclose(arr.values[:12], [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]) and np.allclose(arr.values[12:24], arr.meas[:12])
    Rs = arr.values.reshape(2500, 12)[:, :9].reshape(2500, 3, 3)
    assert np.allclose(Rs @ Rs.transpose(0, 2, 1), np.eye(3), atol=1e-9)


def test_save2D_round_trip(golden_dir, tmp_path):
    """save2D (dataset.cpp:587-617): VERTEX2 + EDGE2 with swapped keys, inverted measurement and the one model's
    information in TORO order; read back with load2D(TORO) every edge is the inverse of the original."""
    arr = _lib.load2d(os.path.join(golden_dir, "w100.graph"))
    sig = [0.05, 0.1, 0.02]
    out = tmp_path / "w100_saved.graph"
    _lib.save2d(str(out), arr, arr.values, sig)
    lines = out.read_text().splitlines()
    assert sum(l.startswith("VERTEX2 ") for l in lines) == 100 and sum(l.startswith("EDGE2 ") for l in lines) == 300
    back = _lib.load2d(str(out), noise_format=A.NOISE_FORMAT_TORO)
    assert back.n_vars == 100 and back.n_factors == 300 and np.allclose(back.values, arr.values, atol=1e-15)
    assert np.all(back.f_noise_kind == A.NOISE_DIAGONAL) and np.allclose(back.noise[:3], sig, rtol=1e-14)
    for f in (0, 7, 299):
        a, b = arr.f_vars[2 * f:2 * f + 2]
        assert back.f_vars[2 * f:2 * f + 2].tolist() == [b, a]
        x, y, th = arr.meas[3 * f:3 * f + 3]
        c, s = np.cos(th), np.sin(th)
        assert np.allclose(back.meas[3 * f:3 * f + 3], [-(c * x + s * y), -(-s * x + c * y), -th], atol=1e-15)


def test_writeBAL_round_trip(golden_dir, tmp_path):
    """writeBAL (SfmData.cpp:249-326): write the dubrovnik fixture from its lowered arrays, read it back: same cameras,
    points and measurements (to the `float` the reader keeps: 1e-6), observations grouped by point."""
    src = os.path.join(golden_dir, "dubrovnik-3-7-pre.txt")
    arr = _lib.read_bal(src)
    out = tmp_path / "dub.txt"
    _lib.write_bal(str(out), arr, arr.values)
    head = out.read_text().split()[:3]
    assert [int(x) for x in head] == [3, 7, int((arr.f_type == A.F_SFM).sum())]
    back = _lib.read_bal(str(out))
    assert back.n_vars == arr.n_vars and back.n_factors == arr.n_factors
    assert np.array_equal(back.var_keys, arr.var_keys) and np.array_equal(back.f_vars, arr.f_vars)
    assert np.allclose(back.values, arr.values, rtol=0, atol=2e-6 * max(1.0, np.max(np.abs(arr.values))))
    assert np.allclose(back.meas, arr.meas, rtol=1e-6)
    # the optimisation problem is the same one: SFMExample_bal's initial error (examples/SFMExample_bal.cpp)
    cams = back.values[:17 * 3].reshape(3, 17)
    R = cams[:, :9].reshape(3, 3, 3)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-12)


def test_host_code_under_address_sanitizer(golden_dir, tmp_path):
    """The host-side sources of the product (lowering, orderings, symbolic analysis with the shard partition, readers and
    writers) compiled with g++ -fsanitize=address,undefined and run over every golden file, all ordering kinds, relaxed
    amalgamation and world sizes 1/2/3/8 — the sanitizer aborts on any out-of-bounds access, leak or undefined behaviour.
    (GPU sanitizers are not available on this pool; the device side is covered by the parity tests.)"""
    src = os.path.join(ROOT, "gtsam_petercdev_amd", "csrc")
    exe = tmp_path / "host_sanitize"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(ROOT, "tests", "native", "host_sanitize.cpp")] + \
          [os.path.join(src, f) for f in ("problem.cpp", "ordering.cpp", "nd.cpp", "lm_policy.cpp", "symbolic.cpp", "io.cpp")] + ["-o", str(exe)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([str(exe), golden_dir, str(tmp_path)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    assert run.stdout.count(" ok") == 7


# ---- the GTSAM-side shim is real code: a compiler sees it against the reference's own headers ----------------------------
def test_integration_shim_parses_against_the_reference_headers(tmp_path):
    """integration/gsx_shim.h (the binding a GTSAM maintainer adds: GsxLevenbergMarquardtOptimizer overriding solve /
    iterate, the lowering table) goes through `g++ -fsyntax-only` against the headers of /root/reference and include/gsx.h.
    Syntax and types only — nothing of the reference is built or linked; the two headers its cmake would generate
    (gtsam/config.h, gtsam/dllexport.h) are replaced by a dozen defines written here into a temporary directory.  Skipped
    where the reference tree is absent (the GPU box)."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "gtsam", "nonlinear")):
        pytest.skip("reference tree absent")
    inc = tmp_path / "gtsam"
    inc.mkdir()
    (inc / "config.h").write_text(
        "#pragma once\n#define GTSAM_VERSION_MAJOR 4\n#define GTSAM_VERSION_MINOR 3\n#define GTSAM_VERSION_PATCH 0\n"
        "#define GTSAM_VERSION_NUMERIC 40300\n#define GTSAM_VERSION_STRING \"4.3.0\"\n#define GTSAM_ALLOCATOR_STL\n"
        "#define GTSAM_THROW_CHEIRALITY_EXCEPTION\n#define GTSAM_ROT3_EXPMAP\n#define GTSAM_POSE3_EXPMAP\n"
        "#define GTSAM_EIGEN_VERSION_WORLD 3\n#define GTSAM_EIGEN_VERSION_MAJOR 4\n#define GTSAM_USE_EIGEN_MKL 0\n")
    (inc / "dllexport.h").write_text("#pragma once\n#define GTSAM_EXPORT\n#define GTSAM_EXTERN_EXPORT extern\n")
    src = tmp_path / "shim_tu.cpp"
    src.write_text('#include "gsx_shim.h"\nint main() { return 0; }\n')
    cmd = ["g++", "-std=c++17", "-fsyntax-only", f"-I{tmp_path}", f"-I{ref}", f"-I{ref}/gtsam/3rdparty/Eigen",
           f"-I{ROOT}/include", f"-I{ROOT}/integration", str(src)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]


# ---- the library's own nested dissection against the reference's CCOLAMD ------------------------------------
def _tree_cost(be, dims):
    """(factor flops sum f^3/3 + f^2 (s+1) + f (s+1)^2, height of the Bayes tree in cliques) of the handle's tree."""
    parent, fronts = be.get_tree()
    F = np.array([float(dims[fv].sum()) for fv, _ in fronts])
    S1 = np.array([float(dims[sv].sum()) + 1.0 for _, sv in fronts])
    children = [[] for _ in fronts]
    roots = []
    for i, p in enumerate(parent):
        (children[p] if p >= 0 else roots).append(i)
    depth, stack = 0, [(r, 1) for r in roots]
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        stack.extend((c, d + 1) for c in children[i])
    return float((F ** 3 / 3 + F * F * S1 + F * S1 * S1).sum()), depth


ND_BOUNDS = {  # workload -> (max flops of ORDER_ND, max ratio to the reference's CCOLAMD flops, max tree height)
    "pose3_100k": (1.5e9, 1.5, 40),
    "pose2_100k": (3.0e8, 1.5, 40),
    "bal1723": (2.6e9, 1.3, 24),
}


@pytest.mark.parametrize("workload", list(ND_BOUNDS))
def test_nested_dissection_against_reference_colamd(lib, oracle, workload):
    """ORDER_ND / ORDER_SCHUR_ND (csrc/nd.cpp: multilevel nested dissection) on the seeded full-size bench problems:
    factor flops of the reference's cliques (no amalgamation) within a small factor of the reference's own CCOLAMD
    ordering (compiled from its C sources, oracle/_ref), at a tree height an order of magnitude lower — the height is
    what a level-scheduled GPU factorization pays for."""
    import bench
    arr, kind = bench.make_problem(workload, 42)
    okind = {"schur_nd": A.ORDER_SCHUR_ND, "nd": A.ORDER_ND}[kind]
    be = _lib.ProductBackend(arr, host_only=True)
    be.set_amalgamation(0.0, 128)
    be.set_ordering(be.compute_ordering(okind))
    flops, depth = _tree_cost(be, arr.var_dims)
    be.close()
    max_flops, max_ratio, max_depth = ND_BOUNDS[workload]
    line = f"{workload}: ORDER_{kind.upper()} flops {flops:.4g} height {depth}"
    if oracle.have_ref_colamd():
        ref = _lib.ProductBackend(arr, host_only=True)
        ref.set_amalgamation(0.0, 128)
        ref.set_ordering(oracle.colamd_ordering(arr))
        rflops, rdepth = _tree_cost(ref, arr.var_dims)
        ref.close()
        line += f" | reference CCOLAMD flops {rflops:.4g} height {rdepth} | ratio {flops / rflops:.2f}"
        assert flops <= max_ratio * rflops, line
        assert depth * 4 <= rdepth, line
    print(line)
    assert flops <= max_flops and depth <= max_depth, line


METIS_BOUNDS = {  # workload -> max ratio of ORDER_ND / ORDER_SCHUR_ND flops to the reference's METIS flops (reference cliques)
    "pose3_100k": 1.45, "pose2_100k": 1.45, "bal1723": 1.1,
}


@pytest.mark.parametrize("workload", list(METIS_BOUNDS))
def test_metis_fixture_and_nested_dissection_against_reference_metis(lib, oracle, golden_dir, workload):
    """The committed METIS orderings (tests/golden/metis_perm_*_seed42.npy) ARE what the reference's own METIS returns
    through Ordering::Metis for the seeded bench problems (regenerated here when oracle/_ref/libmetis_ref.so exists:
    variable ordering bit-identical), and the library's nested dissection is printed and bounded against them: flops of
    the reference's cliques and tree height."""
    import bench
    arr, kind = bench.make_problem(workload, 42)
    keys = bench.load_ordering(arr, workload)
    assert np.array_equal(np.sort(keys), arr.var_keys)
    if oracle.have_ref_metis():
        assert np.array_equal(oracle.metis_ordering(arr), keys), "fixture is not the reference METIS result any more"
    costs = {}
    for name in ("metis", kind):
        be = _lib.ProductBackend(arr, host_only=True)
        be.set_amalgamation(0.0, 128)
        be.set_ordering(keys if name == "metis" else be.compute_ordering({"schur_nd": A.ORDER_SCHUR_ND, "nd": A.ORDER_ND}[kind]))
        costs[name] = _tree_cost(be, arr.var_dims)
        be.close()
    line = (f"{workload}: ORDER_{kind.upper()} flops {costs[kind][0]:.4g} height {costs[kind][1]} | reference METIS flops "
            f"{costs['metis'][0]:.4g} height {costs['metis'][1]} | ratio {costs[kind][0] / costs['metis'][0]:.2f}")
    print(line)
    assert costs[kind][0] <= METIS_BOUNDS[workload] * costs["metis"][0], line


def test_nested_dissection_is_a_permutation_on_awkward_graphs(lib):
    """Disconnected graphs, stars (matching stalls), a clique (no separator), a path, single vertices."""
    rng = np.random.default_rng(5)
    cases = []
    # 3 disconnected Pose2 chains + isolated prior-only variables; a star; a clique; a long path
    def graph(n, edges):
        keys = np.arange(n, dtype=np.uint64)
        arr = A.ProblemArrays(var_keys=keys, var_types=np.full(n, A.VAR_VECTOR, np.int32), var_dims=np.full(n, 2, np.int32),
                              f_type=np.zeros(0, np.int32), f_rows=np.zeros(0, np.int32), f_key_ptr=np.zeros(1, np.int32),
                              f_vars=np.zeros(0, np.int32), f_meas_ptr=np.zeros(1, np.int64), meas=np.zeros(0),
                              f_noise_kind=np.zeros(0, np.int32), f_noise_ptr=np.zeros(1, np.int64), noise=np.zeros(0),
                              values=np.zeros(2 * n), meta={})
        for v in range(n):
            arr = arr.with_factor(A.F_PRIOR, [v], 2, np.zeros(2), A.NOISE_UNIT)
        for a, b in edges:
            arr = arr.with_factor(A.F_BETWEEN, [a, b], 2, np.zeros(2), A.NOISE_UNIT)
        return arr
    cases.append(graph(400, [(i, i + 1) for i in range(399) if i % 100 != 99]))
    cases.append(graph(300, [(0, i) for i in range(1, 300)]))
    cases.append(graph(60, [(i, j) for i in range(60) for j in range(i + 1, 60)]))
    cases.append(graph(1, []))
    cases.append(graph(500, [(i, i + 1) for i in range(499)] + [(int(a), int(b)) for a, b in rng.integers(0, 500, (200, 2)) if a != b]))
    for arr in cases:
        be = _lib.ProductBackend(arr, host_only=True)
        order = be.compute_ordering(A.ORDER_ND)
        assert sorted(order.tolist()) == sorted(arr.var_keys.tolist())
        be.set_ordering(order)
        be.close()


# ---- the Levenberg-Marquardt trust policy as a pure function (csrc/lm_policy.cpp) --------------------------------------------
class _LmState(C.Structure):
    _fields_ = [("lam", C.c_double), ("factor", C.c_double), ("cost", C.c_double), ("outer", C.c_int32), ("inner", C.c_int32)]


class _LmDecision(C.Structure):
    _fields_ = [("verdict", C.c_int32), ("solved", C.c_int32), ("gain_ratio", C.c_double), ("cost_change", C.c_double),
                ("trial_cost", C.c_double), ("lambda_tried", C.c_double)]


def _reference_try_lambda(p, lam, factor, error, solved, lin0, lind, trial):
    """LevenbergMarquardtOptimizer::tryLambda + increaseLambda / decreaseLambda restated literally
    (gtsam/nonlinear/LevenbergMarquardtOptimizer.cpp:141-270, internal/LevenbergMarquardtState.h:70-94): returns
    (verdict, lambda, factor, error) with verdict 1 accepted / 0 retry / 2 stop searching / 3 lambda over its bound."""
    step_ok, stop = False, False
    fidelity, new_error, cost_change = 0.0, float("inf"), 0.0
    if solved:
        lin_change = lin0 - lind
        if lin_change >= 0:
            new_error = trial
            cost_change = error - new_error
            if lin_change > np.finfo(float).eps * lin0:
                fidelity = cost_change / lin_change
                step_ok = fidelity > p.min_model_fidelity
            if abs(cost_change) < p.relative_error_tol * error:
                stop = True
    if step_ok:
        if p.use_fixed_lambda_factor:
            lam /= factor
        else:
            lam *= max(1.0 / 3.0, 1.0 - (2.0 * fidelity - 1.0) ** 3)
            factor *= 2.0
        return 1, max(p.lambda_lower_bound, lam), factor, new_error
    if not stop:
        lam *= factor
        if not p.use_fixed_lambda_factor:
            factor *= 2.0
        return (3 if lam >= p.lambda_upper_bound else 0), lam, factor, error
    return 2, lam, factor, error


@pytest.mark.parametrize("preset", ["legacy", "ceres"])
def test_lm_decide_equals_the_reference_policy(lib, preset):
    lib.gsx_lm_decide.restype = C.c_int32
    p = A.lm_params_legacy() if preset == "legacy" else A.lm_params_ceres()
    rng = np.random.default_rng(11)
    n_verdicts = [0, 0, 0, 0]
    for _ in range(4000):
        lam = 10.0 ** rng.uniform(-8, 6)
        factor = float(rng.choice([2.0, 4.0, 10.0, 64.0]))
        error = 10.0 ** rng.uniform(-3, 6)
        solved = int(rng.random() > 0.1)
        lin0 = error * (1 + 0.1 * rng.normal())
        kind = rng.integers(0, 5)
        lind = lin0 * {0: rng.uniform(0, 1), 1: 1 + 1e-3 * rng.random(), 2: 1.0, 3: 1 - 1e-17, 4: rng.uniform(0.9, 1)}[int(kind)]
        trial = error * {0: rng.uniform(0, 1.5), 1: 1 - 1e-7 * rng.random(), 2: 1.0, 3: rng.uniform(0.5, 1), 4: 1 + 1e-9}[int(rng.integers(0, 5))]
        st = _LmState(lam, factor, error, 3, 7)
        d = _LmDecision()
        assert lib.gsx_lm_decide(C.byref(p), C.byref(st), C.c_int32(solved), C.c_double(lin0), C.c_double(lind), C.c_double(trial),
                                 C.byref(d)) == 0
        v, l2, f2, e2 = _reference_try_lambda(p, lam, factor, error, solved, lin0, lind, trial)
        assert d.verdict == v
        assert st.lam == pytest.approx(l2, rel=1e-15) and st.factor == f2 and st.cost == e2
        assert st.outer == 3 + (v == 1) and st.inner == 7 + (v in (0, 1, 3))
        n_verdicts[v] += 1
    assert min(n_verdicts[:3]) > 20, n_verdicts   # every branch was exercised


def test_lm_decide_reproduces_the_oracle_trace(lib, oracle):
    """Drive an LM run with the ORACLE's numerics and the PRODUCT's decision function: the accept / reject trace and the
    lambdas must be the ones the oracle's own optimizer (a restatement of the reference's) produces."""
    lib.gsx_lm_decide.restype = C.c_int32
    arr = datasets.synth_manhattan_pose2(120, seed=9)
    ordering = _lib.ProductBackend(arr, host_only=True).compute_ordering(A.ORDER_MINDEGREE)
    for p in (A.lm_params_legacy(), A.lm_params_ceres()):
        p.max_iterations = 12
        ref = oracle.oracle_backend(arr)
        ref.set_ordering(ordering)
        expect = ref.lm_optimize(p)
        ob = oracle.oracle_backend(arr)
        ob.set_ordering(ordering)
        st = _LmState(p.lambda_initial, p.lambda_factor, ob.error(), 0, 0)
        trace = []
        while st.outer < p.max_iterations:
            before = st.cost
            ob.linearize()
            while True:
                lam = st.lam
                solved = True
                try:
                    ob.solve(lam, bool(p.diagonal_damping), p.min_diagonal, p.max_diagonal, want_delta=False)
                    lin0, lind = ob.linear_error()
                    trial = ob.retract(None, commit=False)
                except A.IndeterminantLinearSystemException:
                    solved, lin0, lind, trial = False, 0.0, 0.0, 0.0
                d = _LmDecision()
                lib.gsx_lm_decide(C.byref(p), C.byref(st), C.c_int32(int(solved)), C.c_double(lin0), C.c_double(lind),
                                  C.c_double(trial), C.byref(d))
                trace.append((d.trial_cost, lam, 1 if d.verdict == 1 else (0 if solved else -1)))
                if d.verdict == 1:
                    ob.retract(None, commit=True, want_error=False)
                if d.verdict != 0:
                    break
            if d.verdict == 3 or st.cost <= p.error_tol:
                break
            dec = before - st.cost
            if (p.relative_error_tol and dec / before <= p.relative_error_tol) or dec <= p.absolute_error_tol:
                break
        n = len(expect["trace_accepted"])
        assert n >= 3 and len(trace) == n, (len(trace), n)
        assert [t[2] for t in trace] == list(expect["trace_accepted"][:n])
        assert np.allclose([t[1] for t in trace], expect["trace_lambda"][:n], rtol=1e-12)
        assert np.allclose([t[0] for t in trace], expect["trace_error"][:n], rtol=1e-9)
