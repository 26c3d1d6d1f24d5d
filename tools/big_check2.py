import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_petercdev_amd import _lib
lib = _lib.load()
for n, F in [(150, 65), (150, 80), (150, 96), (150, 112), (150, 128), (150, 144), (170, 160)]:
    rng = np.random.default_rng(n * 1000 + F)
    B = rng.normal(size=(n + 5, n)); S = B.T @ B + n * np.eye(n)
    m = np.asfortranarray(S).copy(order="F"); ok = C.c_int32()
    lib.gsx_cholesky_partial(m.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(n), C.c_int32(F), C.c_int32(0), C.byref(ok))
    L11 = np.linalg.cholesky(S[:F, :F]); got = np.triu(m).T
    err = np.abs(got[:F, :F] - L11) / np.abs(L11).max()
    bad = np.argwhere(err > 1e-10)
    print(f"DBG={os.environ.get('GSX_DBG','0')} n={n} F={F} max relerr {err.max():.2e} first bad {bad[0] if len(bad) else None} nbad {len(bad)} badcols {sorted(set((bad[:,1]//16).tolist()))[:12]} badrows {sorted(set((bad[:,0]//16).tolist()))[:12]}")
