#!/bin/bash
# usage (GPU box): tools/sweep_tree.sh <workload> <cfg> [<cfg> ...]   cfg = GSX_TREE_TIERS value, e.g. 45:128,79:256,140:512
wl=$1; shift
mkdir -p gpurun_out/r03
for cfg in "$@"; do
  GSX_TREE_TIERS=$cfg timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-secondary --steps 10 > /tmp/sw.json 2>/tmp/sw.err || { echo "$cfg FAILED"; tail -3 /tmp/sw.err; continue; }
  python - "$cfg" <<'PY' | tee -a gpurun_out/r03/sweep_tree_$wl.txt
import json, sys
d = json.load(open("/tmp/sw.json"))
k = d["kernels"]
print("%-44s ms/step %.3f  small %.3f (%d launches) leaf %.3f big %.3f backsolve %.3f" % (sys.argv[1], d["ms_per_step"], d["factor_small_ms"], [v for n, v in k.items() if n.startswith("front_small")][0]["launches_per_factorization"], d["factor_leaf_ms"], d["factor_big_ms"], d["phases_ms"]["ms_backsolve"]))
PY
done
