#!/bin/bash
# usage: tools/trace_launches.sh <outdir-under-gpurun_out> <workload> [env assignments...]
# rocprofv3 kernel trace of a few factorizations; prints the launches of the LAST solve in order (name, grid, wg, us, gap)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; wl=$2; shift; shift
mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/kernel_probe.py $wl > $out/probe.log 2> $out/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$out/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last begin_factorization marks the last solve
idx = [i for i, r in enumerate(rows) if 'begin_factorization' in r['Kernel_Name']]
i0 = idx[-1]
prev_end = None
tot = {}
with open("$out/last_solve.txt", "w") as o:
    for r in rows[i0:]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('gsx::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:40]
        gap = 0 if prev_end is None else (s - prev_end) / 1e3
        o.write("%-40s q%s grid %8s wg %5s lds %7s  start %8.1f  %8.1f us  gap %6.1f\n" % (name, r.get('Queue_Id', '?'), r['Grid_Size_X'], r['Workgroup_Size_X'], r.get('LDS_Block_Size', '?'), (s - int(rows[i0]['Start_Timestamp'])) / 1e3, (e - s) / 1e3, gap))
        tot[name] = tot.get(name, 0) + (e - s) / 1e3
        prev_end = e
    o.write("total span %.1f us\n" % ((int(rows[-1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        o.write("  %-40s %8.1f us\n" % (k, v))
PY
