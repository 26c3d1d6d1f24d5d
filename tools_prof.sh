#!/bin/bash
# usage: tools_prof.sh <outdir-under-gpurun_out> [bench args...]   (run on the GPU box via gpurun)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-44s calls %6s total_us %10.1f avg_us %9.2f pct %s" % (r['Name'][:44], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
