#!/bin/bash
# round 3: after tools/final_capture_r03.sh — keep the newest run of every pass, summarise the counters into profiles/
F=gpurun_out/r03final
for d in pmc_fetch_bal pmc_write_bal pmc_fetch_p3 pmc_write_p3 sq_bal_1 sq_bal_2 sq_bal_3 sq_p3_1 sq_p3_2 sq_p3_3 prof_bal1723 prof_pose3; do
  newest=$(ls -t $F/$d/runc/*_kernel_trace.csv | head -1 | xargs basename | cut -d_ -f1)
  for f in $F/$d/runc/*; do b=$(basename $f); [ "${b%%_*}" != "$newest" ] && rm -f $f; done
done
python tools/pmc_summary.py $F/pmc_fetch_bal $F/pmc_write_bal profiles/r03_bal1723_pmc_traffic.json > /dev/null
python tools/pmc_summary.py $F/pmc_fetch_p3 $F/pmc_write_p3 profiles/r03_pose3_100k_pmc_traffic.json > /dev/null
python tools/pmc_sq_summary.py profiles/r03_bal1723_pmc_sq.json $F/sq_bal_1 $F/sq_bal_2 $F/sq_bal_3 > /dev/null
python tools/pmc_sq_summary.py profiles/r03_pose3_100k_pmc_sq.json $F/sq_p3_1 $F/sq_p3_2 $F/sq_p3_3 > /dev/null
cp $(ls -t $F/prof_bal1723/runc/*_kernel_stats.csv | head -1) profiles/r03_bal1723_kernel_stats.csv
cp $F/prof_bal1723/bench.json profiles/r03_bal1723_bench_under_rocprof.json
cp $(ls -t $F/prof_pose3/runc/*_kernel_stats.csv | head -1) profiles/r03_pose3_100k_kernel_stats.csv
cp $F/prof_pose3/bench.json profiles/r03_pose3_100k_bench_under_rocprof.json
for w in bal1723 pose3_100k pose2_100k; do cp $F/${w}_metis_bench.json profiles/r03_${w}_metis_ordering_bench.json; done
cp $F/pose3_100k_hard_prior.json profiles/r03_pose3_100k_hard_prior.json
cp $F/pose2_100k_hard_prior.json profiles/r03_pose2_100k_hard_prior.json
