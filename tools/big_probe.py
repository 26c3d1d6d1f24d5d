"""Diagnostic: phase stamps of the big-front kernels (build with GSX_STAMP=1) on single dense fronts."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_petercdev_amd import _lib
lib = _lib.load()
for n, F in [(400, 144), (400, 144), (415, 170), (388, 160), (200, 64)]:
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n + 5, n)); S = B.T @ B + n * np.eye(n)
    m = np.asfortranarray(S).copy(order="F"); ok = C.c_int32()
    lib.gsx_cholesky_partial(m.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(n), C.c_int32(F), C.c_int32(0), C.byref(ok))
print("done")
