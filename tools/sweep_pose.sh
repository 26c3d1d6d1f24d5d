#!/bin/bash
# Re-sweep of the amalgamation settings of the pose-graph workloads (after the launch-structure changes).
run() {
  timeout -k 10 200 python bench.py --workload $3 --amalgamation $1,$2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3', round(d['ms_per_step'],3), d['symbolic']['n_levels'], d['symbolic']['n_big_fronts'], round(d['symbolic']['factor_flops']/1e9,1))"
}
# (round 1, after the launch-structure changes: pose3_100k best at 0.5/80 = 6.8 ms, pose2_100k at 0.5/48..64 = 3.6 ms;
#  below relax 0.5 the level count jumps from ~20 to > 50 and the time doubles)
for cfg in "0.5 48" "0.5 64" "0.5 80" "0.5 96" "1.0 64" "0.35 64"; do set -- $cfg; for w in pose3_100k pose2_100k; do run $1 $2 $w; done; done
