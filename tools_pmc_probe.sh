#!/bin/bash
# usage: tools_pmc_probe.sh <workload> <kernel-substring> <outdir-under-gpurun_out> "<COUNTERS pass 1>" ["<COUNTERS pass 2>" ...]
# (on the GPU box): one rocprofv3 --pmc pass of tools/kernel_probe.py per counter set; prints, per pass, the counters of the
# matching kernel summed over its dispatches and for its largest dispatch (by the first counter).
wl=$1; shift; pat=$1; shift; name=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  out=$root/gpurun_out/$name/pass$i; rm -rf $out; mkdir -p $out
  (cd $root && timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -- python3 tools/kernel_probe.py $wl > $out/log.txt 2>&1) || exit 1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/*/*counter_collection.csv")[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if "$pat" in r["Kernel_Name"]:
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
tot = collections.defaultdict(float)
for d in per.values():
    for k, v in d.items(): tot[k] += v
first = "$ctrs".split()[0]
big = max(per.values(), key=lambda d: d.get(first, 0.0)) if per else {}
print("pass $i: dispatches", len(per))
for k in "$ctrs".split():
    print("  %-34s total %16.0f   largest dispatch %16.0f" % (k, tot.get(k, 0.0), big.get(k, 0.0)))
PY
done
