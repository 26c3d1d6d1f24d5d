#!/bin/bash
# round 3: the leaf launches of BAL-1723 — one launch for all heights (as before), split by panel size, split + four to a workgroup
mkdir -p gpurun_out/r03
for v in "orig GSX_LEAF_SPLIT=100000000 GSX_LEAF_PACK_OFF=1" "pack1024" "pack256 GSX_LEAF_SPLIT=256"; do
  set -- $v; name=$1; shift
  bash tools/trace_launches.sh r03/leaf_$name bal1723 "$@" > /dev/null 2>&1
  echo "== $name"; grep -E "front_leaf|total span" gpurun_out/r03/leaf_$name/last_solve.txt | head -8
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/r03/leaf_$name.json 2>/dev/null && python tools/bench_summary.py gpurun_out/r03/leaf_$name.json | head -1
done
