"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE — see oracle/oracle.cpp).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from gtsam_petercdev_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_COLAMD_PATH = os.path.join(_HERE, "_ref", "libccolamd_ref.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


class OracleBackend(A.Backend):
    def __init__(self, arrays: A.ProblemArrays, device: int = 0):
        super().__init__(load(), "orc_", arrays, device)

    def compute_ordering(self, kind):  # the oracle has no ordering code of its own
        raise NotImplementedError("the oracle takes the ordering as an input")

    def timing(self):
        t = np.zeros(8)
        tree = np.zeros(6)
        self._fn("get_timing")(self._h, t.ctypes.data_as(C.POINTER(C.c_double)),
                               tree.ctypes.data_as(C.POINTER(C.c_double)))
        names = ["linearize", "damp", "eliminate", "backsub", "linear_error", "retract", "error", "symbolic"]
        tn = ["flops", "bytes", "cliques", "depth", "max_f", "max_s"]
        return dict(zip(names, t.tolist())), dict(zip(tn, tree.tolist()))

    def reset_timing(self):
        self._fn("reset_timing")(self._h)

    def set_threads(self, n: int):
        """Host threads for the two loops the reference runs on TBB (factors in linearize, independent subtrees in
        elimination / back-substitution); 1 = serial."""
        self._check(self._fn("set_threads")(self._h, C.c_int32(n)), "set_threads")

    def conditional(self, c):
        nf, nc = C.c_int32(), C.c_int32()
        self._check(self._fn("get_conditional")(self._h, C.c_int32(c), C.byref(nf), C.byref(nc), None), "get_conditional")
        out = np.zeros(nf.value * nc.value)
        self._fn("get_conditional")(self._h, C.c_int32(c), C.byref(nf), C.byref(nc),
                                    out.ctypes.data_as(C.POINTER(C.c_double)))
        return out.reshape(nc.value, nf.value).T  # nf x ncols


def oracle_backend(arrays: A.ProblemArrays, device: int = 0) -> OracleBackend:
    return OracleBackend(arrays, device)


def dogleg_point(delta: float, dx_u, dx_n, blend_only: bool = False) -> np.ndarray:
    """DoglegOptimizerImpl::ComputeDoglegPoint / ComputeBlend restatement on plain vectors."""
    u = np.ascontiguousarray(dx_u, dtype=np.float64)
    n = np.ascontiguousarray(dx_n, dtype=np.float64)
    out = np.zeros_like(u)
    f = load().orc_dogleg_blend if blend_only else load().orc_dogleg_point
    f(C.c_double(delta), u.ctypes.data_as(C.POINTER(C.c_double)), n.ctypes.data_as(C.POINTER(C.c_double)),
      C.c_int64(u.size), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def cholesky_partial(abc: np.ndarray, nfrontal: int):
    """gtsam::choleskyPartial restatement on a (n,n) array (upper triangle significant)."""
    n = abc.shape[0]
    m = np.asfortranarray(abc, dtype=np.float64).copy(order="F")
    ok = C.c_int32()
    load().orc_cholesky_partial(m.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(n), C.c_int32(nfrontal),
                                C.byref(ok))
    return m, bool(ok.value)


def constrained_qr(Ab: np.ndarray, sigmas):
    """noiseModel::Constrained::QR restatement (gtsam/linear/NoiseModel.cpp:478-620) on an (m, n+1) array: returns
    (Rd with rank rows — zeros below, sigmas of the rows: 0 = still a hard constraint, leading columns)."""
    M = np.ascontiguousarray(Ab, dtype=np.float64).copy()
    m, n1 = M.shape
    sg = np.ascontiguousarray(sigmas, dtype=np.float64)
    k = min(m, n1 - 1)
    lead = np.zeros(max(k, 1), np.int32)
    prec = np.zeros(max(k, 1))
    rank = C.c_int32()
    load().orc_constrained_qr(M.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(m), C.c_int32(n1 - 1),
                              sg.ctypes.data_as(C.POINTER(C.c_double)), lead.ctypes.data_as(C.POINTER(C.c_int32)),
                              prec.ctypes.data_as(C.POINTER(C.c_double)), C.byref(rank))
    r = rank.value
    with np.errstate(divide="ignore"):
        out_sigmas = np.where(np.isinf(prec[:r]), 0.0, 1.0 / np.sqrt(prec[:r]))
    return M, out_sigmas, lead[:r].copy()


# ---- the reference's own CCOLAMD, compiled from its C sources (oracle/Makefile) ----------------
def have_ref_colamd() -> bool:
    return os.path.exists(REF_COLAMD_PATH)


def colamd_ordering(arrays: A.ProblemArrays, cmember=None) -> np.ndarray:
    """Ordering::Colamd / ColamdConstrained restated (gtsam/inference/Ordering.cpp:43-125):
    columns = variables in ascending key order, rows = factor indices in graph order,
    knobs dense_row = dense_col = -1, cmember all 0 unless given.  Calls the reference's
    ccolamd() from oracle/_ref/libccolamd_ref.so.  Returns keys in elimination order."""
    lib = C.CDLL(REF_COLAMD_PATH)
    nvars, nfac = arrays.n_vars, arrays.n_factors
    if nvars == 0:
        return np.zeros(0, np.uint64)
    if nvars == 1:
        return arrays.var_keys.copy()
    # VariableIndex: per variable the factor indices in graph order
    fidx = np.repeat(np.arange(nfac, dtype=np.int32), np.diff(arrays.f_key_ptr))
    order = np.lexsort((fidx, arrays.f_vars))  # stable by variable, then factor index
    col_rows = fidx[order]
    counts = np.bincount(arrays.f_vars, minlength=nvars)
    p = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    nentries = int(col_rows.size)
    lib.ccolamd_recommended.restype = C.c_size_t
    alen = int(lib.ccolamd_recommended(C.c_int(nentries), C.c_int(nfac), C.c_int(nvars)))
    Aarr = np.zeros(alen, np.int32)
    Aarr[:nentries] = col_rows
    knobs = (C.c_double * 20)()
    lib.ccolamd_set_defaults(knobs)
    knobs[0] = -1.0  # CCOLAMD_DENSE_ROW
    knobs[1] = -1.0  # CCOLAMD_DENSE_COL
    stats = (C.c_int * 20)()
    cm = np.zeros(nvars, np.int32) if cmember is None else np.ascontiguousarray(cmember, dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    rv = lib.ccolamd(C.c_int(nfac), C.c_int(nvars), C.c_int(alen), ip(Aarr), ip(p), knobs, stats, ip(cm))
    if rv != 1:
        raise RuntimeError(f"ccolamd failed with return value {rv}")
    return arrays.var_keys[p[:nvars]]


# ---- the reference's own METIS 5, compiled from its C sources (oracle/Makefile) ----------------
REF_METIS_PATH = os.path.join(_HERE, "_ref", "libmetis_ref.so")


def have_ref_metis() -> bool:
    return os.path.exists(REF_METIS_PATH)


def metis_index(arrays: A.ProblemArrays):
    """MetisIndex::augment restated (gtsam/inference/MetisIndex-inl.h:27-82).

    Returns (xadj, adj, int_to_var): vertices are numbered in FIRST-SEEN order while walking the
    factors in graph order and each factor's keys in the factor's own order (the bimap of
    :47-57); a vertex's neighbours are the other keys of its factors as a sorted set of those
    integers (:60-70, std::set<int32_t>); xadj/adj are the CSR arrays METIS takes (:75-81).
    Like the reference, only vertices that HAVE a neighbour get an xadj entry (the std::map of
    :29 holds no entry for an isolated key); the reference's own callers never have isolated
    keys next to others, and neither do ours — that case raises here instead of handing METIS a
    short array."""
    nvars = arrays.n_vars
    fv = np.asarray(arrays.f_vars, dtype=np.int64)
    ptr = np.asarray(arrays.f_key_ptr, dtype=np.int64)
    # first-seen numbering
    _, first_pos = np.unique(fv, return_index=True)          # per variable (ascending index) its first slot
    seen_vars = fv[np.sort(first_pos)]                       # variables in first-seen order
    var_to_int = np.full(nvars, -1, np.int64)
    var_to_int[seen_vars] = np.arange(seen_vars.size)
    n = int(seen_vars.size)
    # all ordered pairs (k1, k2), k1 != k2, of every factor
    sizes = np.diff(ptr)
    rows, cols = [], []
    for k in np.unique(sizes):
        if k < 2:
            continue
        sel = np.nonzero(sizes == k)[0]
        keys = var_to_int[fv[(ptr[sel, None] + np.arange(k)[None, :])]]  # (nsel, k)
        a = np.repeat(keys, k, axis=1)
        b = np.tile(keys, (1, k))
        m = a != b
        rows.append(a[m])
        cols.append(b[m])
    if rows:
        r = np.concatenate(rows)
        c = np.concatenate(cols)
        pairs = np.unique(r * n + c)                          # set semantics, sorted by (vertex, neighbour)
        r, c = pairs // n, pairs % n
    else:
        r = c = np.zeros(0, np.int64)
    has = np.zeros(n, bool)
    has[r] = True
    if n > 1 and not has.all():
        raise ValueError("MetisIndex: a key without neighbours next to other keys (the reference's CSR is short there)")
    counts = np.bincount(r, minlength=n)[has] if n else np.zeros(0, np.int64)
    xadj = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return xadj, c.astype(np.int32), seen_vars.astype(np.int64)


def metis_ordering(arrays: A.ProblemArrays) -> np.ndarray:
    """Ordering::Metis restated (gtsam/inference/Ordering.cpp:211-251): METIS_NodeND on the
    MetisIndex CSR with vwgt = options = NULL; result[j] = intToKey(perm[j]).  Calls the
    reference's METIS from oracle/_ref/libmetis_ref.so.  Returns keys in elimination order."""
    xadj, adj, int_to_var = metis_index(arrays)
    n = int(int_to_var.size)
    if n == 0:
        return np.zeros(0, np.uint64)
    if n == 1:
        return arrays.var_keys[int_to_var[:1]].copy()
    lib = C.CDLL(REF_METIS_PATH)
    perm = np.zeros(n, np.int32)
    iperm = np.zeros(n, np.int32)
    nv = C.c_int32(n)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    xadj = np.ascontiguousarray(xadj)
    adj = np.ascontiguousarray(adj)
    rv = lib.METIS_NodeND(C.byref(nv), ip(xadj), ip(adj), None, None, ip(perm), ip(iperm))
    if rv != 1:  # METIS_OK
        raise RuntimeError(f"METIS_NodeND failed with return value {rv}")
    return arrays.var_keys[int_to_var[perm]]
