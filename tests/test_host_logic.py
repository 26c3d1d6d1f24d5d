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
@pytest.mark.parametrize("relax,maxf", [(0.25, 128), (1.0, 64), (50.0, 4096)])
def test_relaxed_amalgamation_is_a_coarsening_of_the_reference_tree(lib, name, relax, maxf):
    """gsx_set_amalgamation(relax > 0): every clique is a union of CONNECTED reference cliques, its separator is the
    separator of its topmost member, the parent relation is the quotient of the reference tree; relax = 0 restores
    the reference tree exactly."""
    arr = PROBLEMS[name]()
    pb = _lib.ProductBackend(arr, host_only=True)
    ordering = pb.compute_ordering(A.ORDER_ND)
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


def test_schur_ordering_puts_landmarks_first(lib):
    arr = PROBLEMS["bal"]()
    pb = _lib.ProductBackend(arr, host_only=True)
    o = pb.compute_ordering(A.ORDER_SCHUR)
    idx = {int(k): i for i, k in enumerate(arr.var_keys)}
    types = [int(arr.var_types[idx[int(k)]]) for k in o]
    n_pts = arr.meta["n_points"]
    assert all(t == A.VAR_VECTOR for t in types[:n_pts]) and all(t == A.VAR_CAMERA for t in types[n_pts:])


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
