#!/bin/bash
# usage (GPU box): tools/sweep_amal.sh <workload> <tiers-cfg> <relax,maxf> ...
wl=$1; shift; cfg=$1; shift
mkdir -p gpurun_out/r03
for am in "$@"; do
  GSX_TREE_TIERS=$cfg timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 10 --amalgamation $am $EXTRA > /tmp/sw.json 2>/tmp/sw.err || { echo "$am FAILED"; tail -3 /tmp/sw.err; continue; }
  python - "$am" <<'PY' | tee -a gpurun_out/r03/sweep_amal_$wl.txt
import json, sys
d = json.load(open("/tmp/sw.json"))
k = d["kernels"]; s = d["symbolic"]
print("%-10s ms/step %.3f  small %.3f leaf %.3f big %.3f (%d launches diag) backsolve %.3f asm %.3f | fronts %d levels %d big %d GF %.2f" % (sys.argv[1], d["ms_per_step"], d["factor_small_ms"], d["factor_leaf_ms"], d["factor_big_ms"], k["big_diag_kernel"]["launches_per_factorization"], d["phases_ms"]["ms_backsolve"], d["phases_ms"]["ms_assemble_hessian"], s["n_fronts"], s["n_levels"], s["n_big_fronts"], s["factor_flops"]/1e9))
PY
done
