"""Loader of the product library (csrc/libgsx.so).  There is NO fallback: if the HIP
library is missing or cannot be loaded every entry point raises."""
from __future__ import annotations

import ctypes as C
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


class ProductBackend(A.Backend):
    def __init__(self, arrays: A.ProblemArrays, device: int = 0, host_only: bool = False):
        """host_only=True skips the upload of the initial values: only the host-side entry points
        (orderings, symbolic analysis, get_tree, stats) are usable — for tests without a GPU."""
        super().__init__(load(), "gsx_", arrays, device, set_initial=not host_only)

    def set_amalgamation(self, relax: float, max_frontal_dim: int = 128):
        """Relaxed clique amalgamation for the next set_ordering (include/gsx.h); 0 = the reference's Bayes tree."""
        self._check(self._fn("set_amalgamation")(self._h, C.c_double(relax), C.c_int32(max_frontal_dim)),
                    "set_amalgamation")

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


def product_backend(arrays: A.ProblemArrays, device: int = 0) -> ProductBackend:
    return ProductBackend(arrays, device)
