#!/bin/bash
# usage (on the GPU box, via gpurun): tools/sweep_amalgamation.sh <outfile> [workloads...]
# ms per LM iteration for explicit (relax, max_frontal_dim) settings next to the library's own choice (no flag)
out=$1; shift
wl=${@:-"bal1723 pose3_100k pose2_100k"}
: > $out
for w in $wl; do
  for a in auto 0,128 0.125,64 0.25,64 0.25,128 0.5,32 0.5,64 0.5,96 1.0,32 1.0,64 2.0,64; do
    if [ $a = auto ]; then flag=""; else flag="--amalgamation $a"; fi
    python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline $flag 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-11s %-9s -> relax %-6s maxf %-4s  %.3f ms/iter  levels %d  flops %.3g' % ('$w', '$a', d['config']['amalgamation']['relax'], d['config']['amalgamation']['max_frontal_dim'], d['ms_per_step'], d['symbolic']['n_levels'], d['symbolic']['factor_flops']))" >> $out
  done
done
cat $out
