"""Loader of the product library (csrc/libgsx.so).  There is NO fallback: if the HIP
library is missing or cannot be loaded every entry point raises."""
from __future__ import annotations

import ctypes as C

import numpy as np
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgsx.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
        _lib = C.CDLL(LIB_PATH)
    return _lib


def device_count() -> int:
    f = load().gsx_device_count
    f.restype = C.c_int32
    return int(f())


def version() -> str:
    f = load().gsx_version
    f.restype = C.c_char_p
    return f().decode()


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64)


class ShardInfo(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("n_cap_fronts", C.c_int32), ("n_own_fronts", C.c_int32),
                ("cap_level0", C.c_int32), ("n_own_factors", C.c_int32), ("cap_doubles", C.c_int64),
                ("cap_flops", C.c_double), ("own_flops", C.c_double), ("total_flops", C.c_double)]


class PartialStats(C.Structure):
    _fields_ = [("n_factors_relinearized", C.c_int32), ("n_panels_reassembled", C.c_int32),
                ("n_fronts_reeliminated", C.c_int32), ("n_fronts", C.c_int32)]


class UpdateStats(C.Structure):
    _fields_ = [("n_vars_added", C.c_int32), ("n_vars_removed", C.c_int32), ("n_factors_added", C.c_int32),
                ("n_factors_removed", C.c_int32), ("n_vars_affected", C.c_int32), ("n_fronts", C.c_int32),
                ("host_symbolic_s", C.c_double), ("device_s", C.c_double)]


class ProductBackend(A.Backend):
    def __init__(self, arrays: A.ProblemArrays, device: int = 0, host_only: bool = False):
        """host_only=True skips the upload of the initial values: only the host-side entry points
        (orderings, symbolic analysis, get_tree, stats) are usable — for tests without a GPU."""
        super().__init__(load(), "gsx_", arrays, device, set_initial=not host_only)

    def set_amalgamation(self, relax: float, max_frontal_dim: int = 128):
        """Relaxed clique amalgamation for the next set_ordering (include/gsx.h); 0 = the reference's Bayes tree."""
        self._check(self._fn("set_amalgamation")(self._h, C.c_double(relax), C.c_int32(max_frontal_dim)),
                    "set_amalgamation")

    # -- one problem over several GPUs (include/gsx.h: gsx_set_shard) ------------------------------------------
    def set_shard(self, rank: int, world: int, allreduce):
        """Make this handle rank `rank` of `world` handles solving ONE problem; call before set_ordering.
        `allreduce(ptr, count) -> None` sums `count` doubles of device memory at address `ptr` over the ranks in place
        (see distributed.torch_allreduce).  Every later call on the handle is collective."""
        def cb(_user, ptr, count):
            try:
                allreduce(int(ptr), int(count))
                return 0
            except Exception as e:  # a raise cannot cross the C frame
                import sys
                print(f"gsx allreduce callback failed: {e!r}", file=sys.stderr)
                return 1
        self._shard_cb = ALLREDUCE_FN(cb)   # keep the trampoline alive as long as the handle
        self._check(self._fn("set_shard")(self._h, C.c_int32(rank), C.c_int32(world), self._shard_cb, None),
                    "set_shard")

    def front_classes(self) -> np.ndarray:
        """gsx_get_front_classes: per front, bits 0-1 = 0 leaf kernel / 1 LDS / 2 blocked / 3 medium, bit 2 = tree front,
        bit 3 = lean leaf."""
        n = C.c_int32()
        self._check(self._fn("get_tree")(self._h, C.byref(n), None, None, None, None, None, None), "get_tree")
        out = np.zeros(max(n.value, 1), np.int32)
        self._check(self._fn("get_front_classes")(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))), "get_front_classes")
        return out[:n.value]

    def shard_probe_buffer(self):
        """(device address, doubles) of a buffer the library hipMalloc'd itself and that holds nothing between calls — what
        distributed.checked_allreduce probes the in-place collective on (gsx_scratch_buffer)."""
        ptr = C.POINTER(C.c_double)()
        n = C.c_int64()
        self._check(self._fn("scratch_buffer")(self._h, C.byref(ptr), C.byref(n)), "scratch_buffer")
        return C.cast(ptr, C.c_void_p).value, n.value

    def shard_info(self):
        """(info dict, front_owner[n_fronts] with -1 = cap, factor_owned[n_factors]) of the current ordering."""
        info = ShardInfo()
        n = C.c_int32()
        self._check(self._fn("get_tree")(self._h, C.byref(n), None, None, None, None, None, None), "get_tree")
        owner = np.zeros(max(n.value, 1), np.int32)
        owned = np.zeros(max(self.arrays.n_factors, 1), np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        self._check(self._fn("get_shard")(self._h, C.byref(info), ip(owner), ip(owned)), "get_shard")
        d = {k: getattr(info, k) for k, _ in ShardInfo._fields_}
        return d, owner[:n.value], owned[:self.arrays.n_factors]

    def relinearize_partial(self, keys, states=None) -> dict:
        """gsx_relinearize_partial (include/gsx.h): move the variables `keys` to `states` (their packed states one after the
        other; None = the handle's values are already current), re-linearize their factors and re-eliminate only the
        cliques that hold them and their ancestors.  Returns the counts of what was redone."""
        ks = np.ascontiguousarray(keys, dtype=np.uint64)
        st = PartialStats()
        if states is None:
            sp, ns = None, 0
        else:
            sv = np.ascontiguousarray(states, dtype=np.float64)
            sp, ns = sv.ctypes.data_as(C.POINTER(C.c_double)), sv.size
        self._check(self._fn("relinearize_partial")(self._h, ks.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int32(ks.size),
                                                    sp, C.c_int64(ns), C.byref(st)), "relinearize_partial")
        return {k: getattr(st, k) for k, _ in PartialStats._fields_}

    def backsubstitute_wildfire(self, threshold: float, want_delta: bool = True):
        """gsx_backsubstitute_wildfire (include/gsx.h): ISAM2's partial back-substitution on the resident undamped
        factorization.  Returns (delta or None, number of frontal variables back-substituted)."""
        out = np.zeros(self.tangent_size) if want_delta else None
        cnt = C.c_int64()
        bad = C.c_uint64()
        ptr = out.ctypes.data_as(C.POINTER(C.c_double)) if want_delta else None
        st = self._fn("backsubstitute_wildfire")(self._h, C.c_double(threshold), ptr,
                                                 C.c_int64(self.tangent_size if want_delta else 0),
                                                 C.byref(cnt), C.byref(bad))
        self._check(st, "backsubstitute_wildfire", key=bad.value)
        return out, cnt.value

    def get_ordering(self) -> np.ndarray:
        """gsx_get_ordering: the keys in the handle's current elimination order."""
        out = np.zeros(self.arrays.n_vars, dtype=np.uint64)
        self._check(self._fn("get_ordering")(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "get_ordering")
        return out

    def update(self, arrays: A.ProblemArrays, factor_origin, new_values) -> dict:
        """gsx_update (include/gsx.h) — ISAM2::update's structural part on the live handle: `arrays` is the WHOLE graph
        after the update, factor_origin[i] the index of its factor i in the current graph (-1: new; current factors not
        named are removed), new_values the packed states of the new variables in ascending-key order.  Kept variables
        keep their linearization point and kept factors their [A b]; only the new factors are linearized."""
        fo = np.ascontiguousarray(factor_origin, dtype=np.int32)
        nv = np.ascontiguousarray(new_values, dtype=np.float64)
        if fo.size != arrays.n_factors:
            raise ValueError("factor_origin must have one entry per factor of the updated graph")
        desc = arrays.desc()
        st = UpdateStats()
        try:
            self._check(self._fn("update")(self._h, C.byref(desc), fo.ctypes.data_as(C.POINTER(C.c_int32)),
                                           nv.ctypes.data_as(C.POINTER(C.c_double)) if nv.size else None,
                                           C.c_int64(nv.size), C.byref(st)), "update")
        except A.GsxError:
            # a failure before the switch leaves the handle as it was; one in the middle of it leaves the NEW problem
            # with every state flag dropped (set_values + set_ordering rebuild it): follow whichever the handle holds
            if int(self._fn("state_size", C.c_int64)(self._h)) == int(arrays.values.size) != int(self.arrays.values.size):
                self.arrays, self._desc = arrays, desc
                self.state_size = int(arrays.values.size)
                self.tangent_size = int(self._fn("tangent_size", C.c_int64)(self._h))
                self.jacobian_size = int(self._fn("jacobian_size", C.c_int64)(self._h))
            raise
        self.arrays, self._desc = arrays, desc
        self.state_size = int(self._fn("state_size", C.c_int64)(self._h))
        self.tangent_size = int(self._fn("tangent_size", C.c_int64)(self._h))
        self.jacobian_size = int(self._fn("jacobian_size", C.c_int64)(self._h))
        return {k: getattr(st, k) for k, _ in UpdateStats._fields_}

    def set_block_jacobians(self, first_factor: int, blocks):
        """gsx_set_block_jacobians: refresh the [A b] blocks of the GSX_F_LINEAR factors first_factor.. in place (blocks =
        the factors' m x (sum d + 1) column-major blocks one after the other, unwhitened: the slot's noise model is folded
        in by the library) — the S5 fallback for factor types the backend does not know."""
        bl = [np.asfortranarray(b, dtype=np.float64).ravel(order="F") for b in blocks]
        flat = np.concatenate(bl) if bl else np.zeros(0)
        self._check(self._fn("set_block_jacobians")(self._h, C.c_int32(first_factor), C.c_int32(len(bl)),
                                                    flat.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(flat.size)),
                    "set_block_jacobians")

    def solve_gfg_h(self, blocks=None):
        """gsx_solve_gfg_h: the linear seam on a kept handle (GSX_F_LINEAR factors only): new numbers, same structure."""
        out = np.zeros(int(self.arrays.var_dims.sum()))
        bad = C.c_uint64()
        if blocks is None:
            bp, nb = None, 0
        else:
            flat = np.concatenate([np.asfortranarray(b, dtype=np.float64).ravel(order="F") for b in blocks])
            bp, nb = flat.ctypes.data_as(C.POINTER(C.c_double)), flat.size
        st = self._fn("solve_gfg_h")(self._h, bp, C.c_int64(nb), out.ctypes.data_as(C.POINTER(C.c_double)),
                                     C.c_int64(out.size), C.byref(bad))
        if st == A.GSX_E_INDETERMINATE:
            raise A.IndeterminantLinearSystemException(bad.value, "solve_gfg_h")
        self._check(st, "solve_gfg_h")
        return out

    def conditional(self, front: int) -> np.ndarray:
        """gsx_get_conditional: [R S d] of clique `front` (gsx_get_tree numbering), n_frontal x n_cols."""
        nf, nc = C.c_int32(), C.c_int32()
        self._check(self._fn("get_conditional")(self._h, C.c_int32(front), C.byref(nf), C.byref(nc), None, C.c_int64(0)),
                    "get_conditional")
        out = np.zeros(nf.value * nc.value)
        self._check(self._fn("get_conditional")(self._h, C.c_int32(front), C.byref(nf), C.byref(nc),
                                                out.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(out.size)),
                    "get_conditional")
        return out.reshape(nc.value, nf.value).T

    def lm_trial(self, relinearize=True, lam=0.0, diagonal_damping=False, min_diagonal=1e-6, max_diagonal=1e32):
        """One LM trial without the policy (gsx_lm_trial): (linear error at 0, at delta, nonlinear error of the trial)."""
        e0, ed, et = C.c_double(), C.c_double(), C.c_double()
        self._check(self._fn("lm_trial")(self._h, C.c_int32(int(relinearize)), C.c_double(lam),
                                         C.c_int32(int(diagonal_damping)), C.c_double(min_diagonal),
                                         C.c_double(max_diagonal), C.byref(e0), C.byref(ed), C.byref(et)), "lm_trial")
        return e0.value, ed.value, et.value

    def stats(self) -> dict:
        s = A.Stats()
        self._check(self._fn("get_stats")(self._h, C.byref(s)), "get_stats")
        return s.as_dict()

    def reset_stats(self):
        self._check(self._fn("reset_stats")(self._h), "reset_stats")

    def set_profiling(self, level: int):
        self._check(self._fn("set_profiling")(self._h, C.c_int32(level)), "set_profiling")

    def synchronize(self):
        self._check(self._fn("synchronize")(self._h), "synchronize")

    def kernel_time(self, name: str):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._fn("kernel_time")(self._h, name.encode(), C.byref(ms), C.byref(n)), "kernel_time")
        return ms.value, n.value


def _dataset_to_arrays(ds, meta) -> A.ProblemArrays:
    lib = load()
    desc = A.ProblemDesc()
    vals, nvals = C.POINTER(C.c_double)(), C.c_int64()
    st = lib.gsx_dataset_get(ds, C.byref(desc), C.byref(vals), C.byref(nvals))
    if st != A.GSX_OK:
        lib.gsx_dataset_free(ds)
        raise A.GsxError(st, "gsx_dataset_get", "")
    nv, nf = desc.n_vars, desc.n_factors

    def arr(ptr, n, dtype):
        return np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[:int(n)].astype(dtype, copy=True)
    kp = arr(desc.f_key_ptr, nf + 1, np.int32)
    mp = arr(desc.f_meas_ptr, nf + 1, np.int64)
    npx = arr(desc.f_noise_ptr, nf + 1, np.int64)
    out = A.ProblemArrays(
        var_keys=arr(desc.var_keys, nv, np.uint64), var_types=arr(desc.var_types, nv, np.int32),
        var_dims=arr(desc.var_dims, nv, np.int32), f_type=arr(desc.f_type, nf, np.int32),
        f_rows=arr(desc.f_rows, nf, np.int32), f_key_ptr=kp, f_vars=arr(desc.f_vars, kp[-1], np.int32),
        f_meas_ptr=mp, meas=arr(desc.meas, mp[-1], np.float64), f_noise_kind=arr(desc.f_noise_kind, nf, np.int32),
        f_noise_ptr=npx, noise=arr(desc.noise, npx[-1], np.float64),
        values=arr(vals, nvals.value, np.float64), meta=meta)
    lib.gsx_dataset_free(ds)
    return out


def read_g2o(path: str, is3D: bool = False) -> A.ProblemArrays:
    """Native g2o reader (gsx_read_g2o, include/gsx.h): graph + anchoring prior + initial values."""
    lib = load()
    lib.gsx_read_g2o.restype = C.c_int32
    lib.gsx_dataset_get.restype = C.c_int32
    lib.gsx_dataset_free.restype = None
    ds = C.c_void_p()
    st = lib.gsx_read_g2o(str(path).encode(), C.c_int32(1 if is3D else 0), C.byref(ds))
    if st != A.GSX_OK:
        raise A.GsxError(st, "gsx_read_g2o", str(path))
    return _dataset_to_arrays(ds, dict(kind="pose3" if is3D else "pose2", source=str(path)))


def load2d(path: str, model_sigmas=None, max_index: int = 0, smart: bool = True,
           noise_format: int = A.NOISE_FORMAT_AUTO, kernel: int = 0) -> A.ProblemArrays:
    """load2D of the reference (gsx_load2d, include/gsx.h): TORO / "graph" 2-D files with poses, landmarks, odometry and
    bearing-range measurements; landmark j has key Symbol('l', j)."""
    lib = load()
    lib.gsx_load2d.restype = C.c_int32
    lib.gsx_dataset_get.restype = C.c_int32
    lib.gsx_dataset_free.restype = None
    ds = C.c_void_p()
    ms = None
    if model_sigmas is not None:
        ms = np.ascontiguousarray(model_sigmas, dtype=np.float64)
        assert ms.size == 3
    st = lib.gsx_load2d(str(path).encode(), None if ms is None else ms.ctypes.data_as(C.POINTER(C.c_double)),
                        C.c_int64(max_index), C.c_int32(int(smart)), C.c_int32(noise_format), C.c_int32(kernel),
                        C.byref(ds))
    if st != A.GSX_OK:
        raise A.GsxError(st, "gsx_load2d", str(path))
    return _dataset_to_arrays(ds, dict(kind="planar", source=str(path)))


def read_bal(path: str, priors: bool = False) -> A.ProblemArrays:
    """Native BAL reader (gsx_read_bal, include/gsx.h)."""
    lib = load()
    lib.gsx_read_bal.restype = C.c_int32
    lib.gsx_dataset_get.restype = C.c_int32
    lib.gsx_dataset_free.restype = None
    ds = C.c_void_p()
    st = lib.gsx_read_bal(str(path).encode(), C.c_int32(1 if priors else 0), C.byref(ds))
    if st != A.GSX_OK:
        raise A.GsxError(st, "gsx_read_bal", str(path))
    return _dataset_to_arrays(ds, dict(kind="bal", source=str(path)))


def write_g2o(path: str, arrays: A.ProblemArrays, packed_values) -> None:
    """Native g2o writer (gsx_write_g2o, include/gsx.h)."""
    lib = load()
    lib.gsx_write_g2o.restype = C.c_int32
    vals = np.ascontiguousarray(packed_values, dtype=np.float64)
    desc = arrays.desc()
    st = lib.gsx_write_g2o(C.byref(desc), vals.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(vals.size),
                           str(path).encode())
    if st != A.GSX_OK:
        raise A.GsxError(st, "gsx_write_g2o", str(path))


def save2d(path: str, arrays: A.ProblemArrays, packed_values, model_sigmas) -> None:
    """save2D of the reference (gsx_save2d, include/gsx.h): TORO VERTEX2 / EDGE2 file."""
    lib = load()
    lib.gsx_save2d.restype = C.c_int32
    vals = np.ascontiguousarray(packed_values, dtype=np.float64)
    ms = np.ascontiguousarray(model_sigmas, dtype=np.float64)
    assert ms.size == 3
    desc = arrays.desc()
    st = lib.gsx_save2d(C.byref(desc), vals.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(vals.size),
                        ms.ctypes.data_as(C.POINTER(C.c_double)), str(path).encode())
    if st != A.GSX_OK:
        raise A.GsxError(st, "gsx_save2d", str(path))


def write_bal(path: str, arrays: A.ProblemArrays, packed_values) -> None:
    """writeBALfromValues of the reference (gsx_write_bal, include/gsx.h)."""
    lib = load()
    lib.gsx_write_bal.restype = C.c_int32
    vals = np.ascontiguousarray(packed_values, dtype=np.float64)
    desc = arrays.desc()
    st = lib.gsx_write_bal(C.byref(desc), vals.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(vals.size),
                           str(path).encode())
    if st != A.GSX_OK:
        raise A.GsxError(st, "gsx_write_bal", str(path))


def dogleg_point(delta: float, dx_u, dx_n) -> np.ndarray:
    """DoglegOptimizerImpl::ComputeDoglegPoint (gsx_dogleg_point, host)."""
    u = np.ascontiguousarray(dx_u, dtype=np.float64)
    n = np.ascontiguousarray(dx_n, dtype=np.float64)
    out = np.zeros_like(u)
    f = load().gsx_dogleg_point
    f.restype = C.c_int32
    st = f(C.c_double(delta), u.ctypes.data_as(C.POINTER(C.c_double)), n.ctypes.data_as(C.POINTER(C.c_double)),
           C.c_int64(u.size), out.ctypes.data_as(C.POINTER(C.c_double)))
    if st != A.GSX_OK:
        raise A.GsxError(st, "gsx_dogleg_point", "")
    return out


def product_backend(arrays: A.ProblemArrays, device: int = 0) -> ProductBackend:
    return ProductBackend(arrays, device)
