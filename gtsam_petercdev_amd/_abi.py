"""ctypes mirror of include/gsx.h and a thin handle wrapper.

The wrapper is parametrised by (library, symbol prefix) so that the very same
Python code can drive the product (``libgsx.so``, prefix ``gsx_``) and — from
tests/ only — the CPU checker, which exports the same entry points under the
prefix ``orc_`` (this package never loads it).
Nothing in this module computes anything: it marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# ---- enums (include/gsx.h) -------------------------------------------------
GSX_OK, GSX_E_INVALID, GSX_E_NO_DEVICE, GSX_E_BAD_ORDERING, GSX_E_INDETERMINATE, GSX_E_STATE, GSX_E_NOMEM = range(7)
VAR_VECTOR, VAR_POSE2, VAR_POSE3, VAR_CAMERA = range(4)
F_LINEAR, F_PRIOR, F_BETWEEN, F_SFM, F_PROJECTION, F_BEARINGRANGE = range(6)
NOISE_FORMAT_G2O, NOISE_FORMAT_TORO, NOISE_FORMAT_GRAPH, NOISE_FORMAT_COV, NOISE_FORMAT_AUTO = range(5)
NOISE_UNIT, NOISE_ISOTROPIC, NOISE_DIAGONAL, NOISE_GAUSSIAN, NOISE_CONSTRAINED = range(5)
NOISE_ROBUST_HUBER, NOISE_ROBUST_TUKEY, NOISE_ROBUST_CAUCHY, NOISE_BASE_MASK = 1 << 4, 2 << 4, 3 << 4, 15
ORDER_NATURAL, ORDER_MINDEGREE, ORDER_ND, ORDER_SCHUR, ORDER_SCHUR_ND = range(5)

STATE_DIM = {VAR_POSE2: 3, VAR_POSE3: 12, VAR_CAMERA: 17}
TANGENT_DIM = {VAR_POSE2: 3, VAR_POSE3: 6, VAR_CAMERA: 9}

_STATUS_NAMES = {
    GSX_OK: "GSX_OK", GSX_E_INVALID: "GSX_E_INVALID", GSX_E_NO_DEVICE: "GSX_E_NO_DEVICE",
    GSX_E_BAD_ORDERING: "GSX_E_BAD_ORDERING", GSX_E_INDETERMINATE: "GSX_E_INDETERMINATE",
    GSX_E_STATE: "GSX_E_STATE", GSX_E_NOMEM: "GSX_E_NOMEM",
}


AMALGAMATION_AUTO = -1.0   # gsx_set_amalgamation: the library's own choice (the default of a new handle)


class GsxError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        self.status = status
        super().__init__(f"{what}: {_STATUS_NAMES.get(status, status)} {detail}".strip())


class IndeterminantLinearSystemException(GsxError):
    """gtsam/linear/linearExceptions.h:94-97 — carries a key of the failing clique."""

    def __init__(self, key, what: str = "solve", detail: str = ""):
        self.key = key   # None when the entry point does not report one
        near = f"near variable {key}" if key is not None else "(no variable reported by this entry point)"
        super().__init__(GSX_E_INDETERMINATE, what, (near + " " + detail).strip())


_p = C.POINTER


class ProblemDesc(C.Structure):
    _fields_ = [
        ("n_vars", C.c_int32), ("var_keys", _p(C.c_uint64)), ("var_types", _p(C.c_int32)),
        ("var_dims", _p(C.c_int32)),
        ("n_factors", C.c_int32), ("f_type", _p(C.c_int32)), ("f_rows", _p(C.c_int32)),
        ("f_key_ptr", _p(C.c_int32)), ("f_vars", _p(C.c_int32)),
        ("f_meas_ptr", _p(C.c_int64)), ("meas", _p(C.c_double)),
        ("f_noise_kind", _p(C.c_int32)), ("f_noise_ptr", _p(C.c_int64)), ("noise", _p(C.c_double)),
    ]


class LMParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("relative_error_tol", C.c_double),
        ("absolute_error_tol", C.c_double), ("error_tol", C.c_double),
        ("lambda_initial", C.c_double), ("lambda_factor", C.c_double),
        ("lambda_upper_bound", C.c_double), ("lambda_lower_bound", C.c_double),
        ("min_model_fidelity", C.c_double), ("diagonal_damping", C.c_int32),
        ("use_fixed_lambda_factor", C.c_int32), ("min_diagonal", C.c_double),
        ("max_diagonal", C.c_double), ("verbosity", C.c_int32),
    ]


class LMResult(C.Structure):
    _fields_ = [
        ("initial_error", C.c_double), ("final_error", C.c_double), ("final_lambda", C.c_double),
        ("iterations", C.c_int32), ("inner_iterations", C.c_int32), ("n_solve_failures", C.c_int32),
        ("trace_len", C.c_int32), ("trace_cap", C.c_int32),
        ("trace_error", _p(C.c_double)), ("trace_lambda", _p(C.c_double)),
        ("trace_accepted", _p(C.c_int32)),
    ]


class Stats(C.Structure):
    _fields_ = (
        [(n, C.c_int64) for n in ("n_fronts", "n_levels", "max_front_dim", "max_front_rows",
                                  "n_small_fronts", "n_big_fronts")]
        + [(n, C.c_double) for n in ("factor_flops", "front_bytes", "lpanel_bytes", "jacobian_bytes",
                                     "hessian_bytes", "total_dim",
                                     "ms_linearize", "ms_assemble_hessian", "ms_factorize", "ms_backsolve",
                                     "ms_linear_error", "ms_retract", "ms_error")]
        + [(n, C.c_int64) for n in ("n_linearize", "n_factorize", "n_backsolve", "n_error", "n_cheirality")]
        + [("amalgamation_relax", C.c_double), ("amalgamation_max_frontal_dim", C.c_int64),
           ("n_medium_fronts", C.c_int64), ("n_tree_fronts", C.c_int64), ("n_upper_levels", C.c_int64),
           ("n_constraint_rows", C.c_int64), ("n_constrained_fronts", C.c_int64)]
    )

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def lm_params_legacy() -> LMParams:
    """LevenbergMarquardtParams::SetLegacyDefaults (LevenbergMarquardtParams.h:69-82)."""
    return LMParams(100, 1e-5, 1e-5, 0.0, 1e-5, 10.0, 1e5, 0.0, 1e-3, 0, 1, 1e-6, 1e32, 0)


def lm_params_ceres() -> LMParams:
    """LevenbergMarquardtParams::SetCeresDefaults (LevenbergMarquardtParams.h:85-98)."""
    return LMParams(50, 1e-6, 0.0, 0.0, 1e-4, 2.0, 1e32, 1e-16, 1e-3, 1, 0, 1e-6, 1e32, 0)


# ---- flat problem arrays -----------------------------------------------------
@dataclass
class ProblemArrays:
    """Numpy image of gsx_problem_desc plus the packed initial Values."""
    var_keys: np.ndarray
    var_types: np.ndarray
    var_dims: np.ndarray
    f_type: np.ndarray
    f_rows: np.ndarray
    f_key_ptr: np.ndarray
    f_vars: np.ndarray
    f_meas_ptr: np.ndarray
    meas: np.ndarray
    f_noise_kind: np.ndarray
    f_noise_ptr: np.ndarray
    noise: np.ndarray
    values: Optional[np.ndarray] = None
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        c = np.ascontiguousarray
        self.var_keys = c(self.var_keys, dtype=np.uint64)
        for n in ("var_types", "var_dims", "f_type", "f_rows", "f_key_ptr", "f_vars", "f_noise_kind"):
            setattr(self, n, c(getattr(self, n), dtype=np.int32))
        for n in ("f_meas_ptr", "f_noise_ptr"):
            setattr(self, n, c(getattr(self, n), dtype=np.int64))
        for n in ("meas", "noise"):
            a = c(getattr(self, n), dtype=np.float64)
            if a.size == 0:
                a = np.zeros(1)  # keep a valid pointer
            setattr(self, n, a)
        if self.values is not None:
            self.values = c(self.values, dtype=np.float64)

    def with_factor(self, f_type, var_indices, rows, meas, noise_kind, noise_params=()):
        """A copy with one more factor appended (NonlinearFactorGraph::add on the lowered arrays), e.g. the prior the
        reference's examples add after load2D (examples/Pose2SLAMExample_graph.cpp:44-47)."""
        nm, nn = int(self.f_meas_ptr[-1]), int(self.f_noise_ptr[-1])
        meas = np.asarray(meas, float).ravel()
        npar = np.asarray(noise_params, float).ravel()
        return ProblemArrays(
            self.var_keys, self.var_types, self.var_dims,
            np.append(self.f_type, f_type), np.append(self.f_rows, rows),
            np.append(self.f_key_ptr, self.f_key_ptr[-1] + len(var_indices)), np.append(self.f_vars, var_indices),
            np.append(self.f_meas_ptr, nm + meas.size), np.concatenate([self.meas[:nm], meas]),
            np.append(self.f_noise_kind, noise_kind), np.append(self.f_noise_ptr, nn + npar.size),
            np.concatenate([self.noise[:nn], npar]), None if self.values is None else self.values.copy(),
            dict(self.meta))

    @property
    def n_vars(self):
        return int(self.var_keys.shape[0])

    @property
    def n_factors(self):
        return int(self.f_type.shape[0])

    def state_dims(self) -> np.ndarray:
        sd = self.var_dims.copy()
        sd[self.var_types == VAR_POSE3] = 12
        sd[self.var_types == VAR_CAMERA] = 17
        return sd

    def state_offsets(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.state_dims())]).astype(np.int64)

    def tangent_offsets(self) -> np.ndarray:
        return np.concatenate([[0], np.cumsum(self.var_dims)]).astype(np.int64)

    def jacobian_offsets(self) -> np.ndarray:
        """Offset of each factor's [A b] (m x (sum d + 1), column-major) in gsx_get_jacobians."""
        nk = np.diff(self.f_key_ptr)
        dsum = np.add.reduceat(self.var_dims[self.f_vars], self.f_key_ptr[:-1]) if self.f_vars.size else np.zeros(0)
        dsum = np.where(nk > 0, dsum, 0)
        sizes = self.f_rows.astype(np.int64) * (dsum.astype(np.int64) + 1)
        return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    def desc(self) -> ProblemDesc:
        def ptr(a, t):
            return a.ctypes.data_as(_p(t))
        return ProblemDesc(
            self.n_vars, ptr(self.var_keys, C.c_uint64), ptr(self.var_types, C.c_int32),
            ptr(self.var_dims, C.c_int32),
            self.n_factors, ptr(self.f_type, C.c_int32), ptr(self.f_rows, C.c_int32),
            ptr(self.f_key_ptr, C.c_int32), ptr(self.f_vars, C.c_int32),
            ptr(self.f_meas_ptr, C.c_int64), ptr(self.meas, C.c_double),
            ptr(self.f_noise_kind, C.c_int32), ptr(self.f_noise_ptr, C.c_int64), ptr(self.noise, C.c_double),
        )


def _dptr(a):
    return a.ctypes.data_as(_p(C.c_double))


class Backend:
    """One solver handle.  `lib` is a ctypes CDLL, `prefix` is "gsx_" or "orc_"."""

    def __init__(self, lib, prefix: str, arrays: ProblemArrays, device: int = 0, set_initial: bool = True):
        self._lib = lib
        self._pfx = prefix
        self.arrays = arrays
        self._h = C.c_void_p()
        self._desc = arrays.desc()  # keeps pointers alive together with `arrays`
        if prefix == "gsx_":
            st = self._fn("create")(C.byref(self._desc), C.c_int32(device), C.byref(self._h))
        else:
            st = self._fn("create")(C.byref(self._desc), C.byref(self._h))
        if st != GSX_OK:
            raise GsxError(st, prefix + "create")
        self.state_size = int(self._fn("state_size", C.c_int64)(self._h))
        self.tangent_size = int(self._fn("tangent_size", C.c_int64)(self._h))
        self.jacobian_size = int(self._fn("jacobian_size", C.c_int64)(self._h))
        if set_initial and arrays.values is not None:
            self.set_values(arrays.values)

    def _fn(self, name, restype=C.c_int):
        f = getattr(self._lib, self._pfx + name)
        f.restype = restype
        return f

    def _check(self, st, what, key=None):
        if st == GSX_OK:
            return
        detail = ""
        if self._pfx == "gsx_":
            f = self._fn("last_error", C.c_char_p)
            detail = (f(self._h) or b"").decode()
        if st == GSX_E_INDETERMINATE:   # IndeterminantLinearSystemException wherever a factorization is involved
            raise IndeterminantLinearSystemException(key, self._pfx + what, detail)
        raise GsxError(st, self._pfx + what, detail)

    def close(self):
        if self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ordering ---------------------------------------------------------
    def set_ordering(self, keys):
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        self._check(self._fn("set_ordering")(self._h, k.ctypes.data_as(_p(C.c_uint64)), C.c_int32(k.size)),
                    "set_ordering")

    def compute_ordering(self, kind: int) -> np.ndarray:
        out = np.zeros(self.arrays.n_vars, dtype=np.uint64)
        self._check(self._fn("compute_ordering")(self._h, C.c_int32(kind), out.ctypes.data_as(_p(C.c_uint64))),
                    "compute_ordering")
        return out

    def get_tree(self):
        """(parent list, [(frontal var indices, separator var indices)]) of the Bayes tree."""
        n, ns = C.c_int32(), C.c_int64()
        self._check(self._fn("get_tree")(self._h, C.byref(n), C.byref(ns), None, None, None, None, None),
                    "get_tree")
        nf = n.value
        parent = np.zeros(max(nf, 1), np.int32)
        fptr = np.zeros(nf + 1, np.int32)
        sptr = np.zeros(nf + 1, np.int32)
        fv = np.zeros(max(self.arrays.n_vars, 1), np.int32)
        sv = np.zeros(max(1, ns.value), np.int32)
        ip = lambda a: a.ctypes.data_as(_p(C.c_int32))
        self._check(self._fn("get_tree")(self._h, C.byref(n), C.byref(ns), ip(parent), ip(fptr), ip(fv),
                                         ip(sptr), ip(sv)), "get_tree")
        fronts = []
        for c in range(nf):
            fronts.append((fv[fptr[c]:fptr[c + 1]].tolist(), sv[sptr[c]:sptr[c + 1]].tolist()))
        return parent[:nf].tolist(), fronts

    # -- state -------------------------------------------------------------
    def set_values(self, packed):
        a = np.ascontiguousarray(packed, dtype=np.float64)
        self._check(self._fn("set_values")(self._h, _dptr(a), C.c_int64(a.size)), "set_values")

    def get_values(self) -> np.ndarray:
        out = np.zeros(self.state_size)
        self._check(self._fn("get_values")(self._h, _dptr(out), C.c_int64(out.size)), "get_values")
        return out

    # -- hot path ------------------------------------------------------------
    def error(self) -> float:
        e = C.c_double()
        self._check(self._fn("error")(self._h, C.byref(e)), "error")
        return e.value

    def linearize(self):
        self._check(self._fn("linearize")(self._h), "linearize")

    def jacobians(self) -> np.ndarray:
        out = np.zeros(max(1, self.jacobian_size))
        self._check(self._fn("get_jacobians")(self._h, _dptr(out), C.c_int64(self.jacobian_size)), "get_jacobians")
        return out[: self.jacobian_size]

    def hessian_diagonal(self) -> np.ndarray:
        out = np.zeros(self.tangent_size)
        self._check(self._fn("hessian_diagonal")(self._h, _dptr(out), C.c_int64(out.size)), "hessian_diagonal")
        return out

    def solve(self, lam=0.0, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32, want_delta=True):
        out = np.zeros(self.tangent_size) if want_delta else None
        bad = C.c_uint64(0)
        st = self._fn("solve")(self._h, C.c_double(lam), C.c_int32(int(diagonal_damping)),
                               C.c_double(min_diagonal), C.c_double(max_diagonal),
                               _dptr(out) if want_delta else None, C.c_int64(self.tangent_size), C.byref(bad))
        if st == GSX_E_INDETERMINATE:
            raise IndeterminantLinearSystemException(bad.value, self._pfx + "solve")
        self._check(st, "solve")
        return out

    def linear_error(self):
        e0, ed = C.c_double(), C.c_double()
        self._check(self._fn("linear_error")(self._h, C.byref(e0), C.byref(ed)), "linear_error")
        return e0.value, ed.value

    def retract(self, delta=None, commit=False, want_error=True):
        err = C.c_double()
        if delta is None:
            dp, n = None, 0
        else:
            d = np.ascontiguousarray(delta, dtype=np.float64)
            dp, n = _dptr(d), d.size
        self._check(self._fn("retract")(self._h, dp, C.c_int64(n), C.c_int32(int(commit)),
                                        C.byref(err) if want_error else None), "retract")
        return err.value if want_error else None

    # -- optimizers ------------------------------------------------------------
    @staticmethod
    def _result(trace_cap):
        tr_e = np.zeros(trace_cap)
        tr_l = np.zeros(trace_cap)
        tr_a = np.zeros(trace_cap, np.int32)
        r = LMResult()
        r.trace_cap = trace_cap
        r.trace_error = _dptr(tr_e)
        r.trace_lambda = _dptr(tr_l)
        r.trace_accepted = tr_a.ctypes.data_as(_p(C.c_int32))
        return r, (tr_e, tr_l, tr_a)

    @staticmethod
    def _result_dict(r, tr):
        n = min(r.trace_len, r.trace_cap)
        return dict(initial_error=r.initial_error, final_error=r.final_error, final_lambda=r.final_lambda,
                    iterations=r.iterations, inner_iterations=r.inner_iterations,
                    n_solve_failures=r.n_solve_failures,
                    trace_error=tr[0][:n].copy(), trace_lambda=tr[1][:n].copy(), trace_accepted=tr[2][:n].copy())

    def lm_optimize(self, params: LMParams, trace_cap=4096):
        r, tr = self._result(trace_cap)
        self._check(self._fn("lm_optimize")(self._h, C.byref(params), C.byref(r)), "lm_optimize")
        return self._result_dict(r, tr)

    def lm_reset(self, params: LMParams):
        self._check(self._fn("lm_reset")(self._h, C.byref(params)), "lm_reset")

    def lm_iterate(self, params: LMParams):
        e, l = C.c_double(), C.c_double()
        self._check(self._fn("lm_iterate")(self._h, C.byref(params), C.byref(e), C.byref(l)), "lm_iterate")
        return e.value, l.value

    def marginal_covariance(self, key) -> np.ndarray:
        """Marginals::marginalCovariance(key): the dA x dA block of H^-1 of the current linearization."""
        i = int(np.searchsorted(self.arrays.var_keys, np.uint64(key)))
        d = int(self.arrays.var_dims[i])
        out = np.zeros(d * d)
        self._check(self._fn("marginal_covariance")(self._h, C.c_uint64(int(key)), out.ctypes.data_as(_p(C.c_double)),
                                                    C.c_int64(d * d)), "marginal_covariance")
        return out.reshape(d, d).T  # column-major

    def joint_marginal_covariance(self, keys) -> np.ndarray:
        """Marginals::jointMarginalCovariance(keys): the D x D joint covariance, blocks in the order of `keys`."""
        ks = np.ascontiguousarray(keys, dtype=np.uint64)
        D = int(sum(int(self.arrays.var_dims[int(np.searchsorted(self.arrays.var_keys, k))]) for k in ks))
        out = np.zeros(D * D)
        self._check(self._fn("joint_marginal_covariance")(self._h, ks.ctypes.data_as(_p(C.c_uint64)), C.c_int32(ks.size),
                                                          out.ctypes.data_as(_p(C.c_double)), C.c_int64(D * D)),
                    "joint_marginal_covariance")
        return out.reshape(D, D)

    def dogleg_optimize(self, delta_initial=1.0, max_iterations=100, relative_error_tol=1e-5, absolute_error_tol=1e-5,
                        error_tol=0.0, trace_cap=4096):
        """DoglegOptimizer (ONE_STEP_PER_ITERATION); trace_lambda / final_lambda carry the trust-region radius."""
        r, tr = self._result(trace_cap)
        st = self._fn("dogleg_optimize")(self._h, C.c_double(delta_initial), C.c_int32(max_iterations),
                                         C.c_double(relative_error_tol), C.c_double(absolute_error_tol),
                                         C.c_double(error_tol), C.byref(r))
        if st == GSX_E_INDETERMINATE:
            raise IndeterminantLinearSystemException(None, self._pfx + "dogleg_optimize")
        self._check(st, "dogleg_optimize")
        return self._result_dict(r, tr)

    def gn_optimize(self, max_iterations=100, relative_error_tol=1e-5, absolute_error_tol=1e-5, error_tol=0.0,
                    trace_cap=4096):
        r, tr = self._result(trace_cap)
        st = self._fn("gn_optimize")(self._h, C.c_int32(max_iterations), C.c_double(relative_error_tol),
                                     C.c_double(absolute_error_tol), C.c_double(error_tol), C.byref(r))
        if st == GSX_E_INDETERMINATE:
            raise IndeterminantLinearSystemException(None, self._pfx + "gn_optimize")
        self._check(st, "gn_optimize")
        return self._result_dict(r, tr)
