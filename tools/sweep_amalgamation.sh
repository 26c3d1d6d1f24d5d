#!/bin/bash
# Sweep of the relaxed-amalgamation settings (gsx_set_amalgamation) per workload on the GPU box:
#   gpurun -- 'bash tools/sweep_amalgamation.sh > gpurun_out/sweep.log 2>&1'
# Prints: relax max_frontal_dim workload ms/LM-iteration levels big-fronts GFLOP.  bench.py's AMALGAMATION table holds
# the winners.
run() {
  timeout -k 10 200 python bench.py --workload $3 --amalgamation $1,$2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3', round(d['ms_per_step'],3), d['symbolic']['n_levels'], d['symbolic']['n_big_fronts'], round(d['symbolic']['factor_flops']/1e9,1))"
}
for cfg in "0 128" "0.5 64" "0.5 256" "0.75 128" "1.0 128" "1.0 64" "2.0 64"; do set -- $cfg; for w in pose3_100k pose2_100k; do run $1 $2 $w; done; done
for cfg in "0 128" "0.15 128" "0.25 64" "0.25 96" "0.25 128" "0.2 128" "0.3 128"; do set -- $cfg; run $1 $2 bal1723; done
