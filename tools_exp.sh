#!/bin/bash
# usage: tools_exp.sh <workload> <kernel-substring> <variant>...   (on the GPU box): for each build/exp_<variant>/libgsx.so,
# run tools/kernel_probe.py under rocprofv3 and print the stats rows of the kernels that match
wl=$1; shift; pat=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  cp $root/build/exp_$v/libgsx.so $root/gtsam_petercdev_amd/csrc/libgsx.so
  out=$root/gpurun_out/exp_$v; rm -rf $out; mkdir -p $out
  (cd $root && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/kernel_probe.py $wl > $out/log.txt 2>&1)
  python3 - <<PY
import csv, glob
f = glob.glob("$out/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "$pat" in r['Name']:
        print("%-10s %-40s calls %5s avg_us %9.2f max_us %9.2f" % ("$v", r['Name'].replace('gsx::','')[:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
