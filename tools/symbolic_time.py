"""Host time of gsx_set_ordering's symbolic analysis at config-5 size (no GPU work): GSX_TIME_SYMBOLIC=1 prints the passes."""
import sys
import time
sys.path.insert(0, ".")
from gtsam_petercdev_amd import _abi as A, _lib, datasets
K, M, OBS = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 1000000, 4500000)
a2, _ = datasets.synth_visual_slam(K, M, OBS)
pb = _lib.ProductBackend(a2, host_only=True)
t0 = time.perf_counter()
o = pb.compute_ordering(A.ORDER_SCHUR_ND)
t1 = time.perf_counter()
pb.set_ordering(o)
t2 = time.perf_counter()
print("ordering %.3f s  symbolic %.3f s  fronts %d" % (t1 - t0, t2 - t1, pb.stats()["n_fronts"]), flush=True)
