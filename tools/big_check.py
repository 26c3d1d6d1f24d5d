"""Diagnostic: the blocked big-front path (bigfront.hip) through gsx_cholesky_partial against numpy, with wall times."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gtsam_petercdev_amd import _lib
lib = _lib.load()
worst = 0.0
for n, F in [(141, 31), (150, 150), (200, 200), (200, 37), (333, 150), (388, 387), (400, 144), (415, 170), (515, 257), (700, 400), (900, 64)]:
    rng = np.random.default_rng(n * 1000 + F)
    B = rng.normal(size=(n + 5, n)); S = B.T @ B + n * np.eye(n)
    m = np.asfortranarray(S).copy(order="F"); ok = C.c_int32()
    t0 = time.perf_counter()
    st = lib.gsx_cholesky_partial(m.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(n), C.c_int32(F), C.c_int32(0), C.byref(ok))
    dt = time.perf_counter() - t0
    # expected: L11 = chol(S11), L21 = S21 L11^-T, C = S22 - L21 L21' (lower)
    L11 = np.linalg.cholesky(S[:F, :F])
    L21 = np.linalg.solve(L11, S[:F, F:]).T
    Cs = S[F:, F:] - L21 @ L21.T
    got = np.triu(m).T   # the ABI returns R = L' in the upper triangle (reference convention)
    e1 = np.abs(got[:F, :F] - L11).max() / np.abs(L11).max()
    e2 = np.abs(got[F:, :F] - L21).max() / max(np.abs(L21).max(), 1e-300) if F < n else 0.0
    e3 = np.abs(got[F:, F:] - np.tril(Cs)).max() / max(np.abs(Cs).max(), 1e-300) if F < n else 0.0
    worst = max(worst, e1, e2, e3)
    print(f"n={n:4d} F={F:4d} st={st} ok={ok.value} relerr L11 {e1:.2e} L21 {e2:.2e} C {e3:.2e}  ({dt*1e3:.1f} ms wall incl. copies)")
print("WORST", worst)
assert worst < 1e-11
