"""Diagnostic: time the in-LDS partial Cholesky alone (dense_small_kernel) under rocprofv3."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_petercdev_amd import _lib
lib = _lib.load()
for n, F in [(140, 60), (140, 120), (100, 40), (60, 24), (48, 12), (40, 3)]:
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n + 5, n)); S = B.T @ B + n * np.eye(n)
    for rep in range(3):
        m = np.asfortranarray(S).copy(order="F"); ok = C.c_int32()
        lib.gsx_cholesky_partial(m.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(n), C.c_int32(F), C.c_int32(0), C.byref(ok))
print("done")
