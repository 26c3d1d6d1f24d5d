#!/bin/bash
# round 3: the three workloads with the library's ordering and with the reference's METIS ordering
mkdir -p gpurun_out/r03
for wl in pose3_100k pose2_100k bal1723; do
  for ord in lib metis; do
    extra=""; [ $ord = metis ] && extra="--ordering metis"
    timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary $extra > gpurun_out/r03/${wl}_${ord}_now.json 2> gpurun_out/r03/${wl}_${ord}_now.err || tail -3 gpurun_out/r03/${wl}_${ord}_now.err
  done
done
python tools/bench_summary.py gpurun_out/r03/*_now.json | grep -v "^      "
