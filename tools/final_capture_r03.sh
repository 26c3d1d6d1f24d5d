#!/bin/bash
# round 3 final captures (run on the GPU box via gpurun): bench lines, rocprofv3 kernel stats, FETCH/WRITE passes, SQ pass
set -o pipefail
out=gpurun_out/r03final; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bal1723_bench.json 2> $out/bal1723_bench.err && echo bal ok
for w in pose3_100k pose2_100k bal49; do timeout -k 10 300 python bench.py --workload $w > $out/${w}_bench.json 2> $out/${w}_bench.err && echo $w ok; done
timeout -k 10 300 python bench.py --ordering metis --no-cpu-baseline --no-secondary > $out/bal1723_metis_bench.json 2>/dev/null && echo bal metis ok
timeout -k 10 300 python bench.py --workload pose3_100k --ordering metis --no-cpu-baseline > $out/pose3_100k_metis_bench.json 2>/dev/null && echo pose3 metis ok
timeout -k 10 300 python bench.py --workload pose2_100k --ordering metis --no-cpu-baseline > $out/pose2_100k_metis_bench.json 2>/dev/null && echo pose2 metis ok
for w in pose3_100k pose2_100k; do timeout -k 10 300 python bench.py --workload $w --hard-prior --no-cpu-baseline --no-secondary > $out/${w}_hard_prior.json 2>/dev/null && echo $w hard prior ok; done
bash tools_prof.sh r03final/prof_bal1723 --steps 20 --warmup 3 --no-secondary > $out/prof_bal1723.txt 2>&1 && echo prof bal ok
bash tools_prof.sh r03final/prof_pose3 --workload pose3_100k --steps 20 --warmup 3 > $out/prof_pose3.txt 2>&1 && echo prof pose3 ok
bash tools_pmc.sh r03final/pmc_fetch_bal FETCH_SIZE --steps 5 --warmup 1 --no-secondary > /dev/null 2>&1 && echo pmc fetch bal ok
bash tools_pmc.sh r03final/pmc_write_bal WRITE_SIZE --steps 5 --warmup 1 --no-secondary > /dev/null 2>&1 && echo pmc write bal ok
bash tools_pmc.sh r03final/pmc_fetch_p3 FETCH_SIZE --workload pose3_100k --steps 5 --warmup 1 > /dev/null 2>&1 && echo pmc fetch p3 ok
bash tools_pmc.sh r03final/pmc_write_p3 WRITE_SIZE --workload pose3_100k --steps 5 --warmup 1 > /dev/null 2>&1 && echo pmc write p3 ok
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  bash tools_pmc.sh r03final/sq_bal_$i "$c" --steps 3 --warmup 1 --no-secondary > /dev/null 2>&1 && echo sq bal $i ok
  bash tools_pmc.sh r03final/sq_p3_$i "$c" --workload pose3_100k --steps 3 --warmup 1 > /dev/null 2>&1 && echo sq p3 $i ok
done
python - <<PY
import json
for w in ("bal1723","pose3_100k","pose2_100k","bal49"):
    d=json.load(open(f"gpurun_out/r03final/{w}_bench.json")); print(w, round(d["ms_per_step"],4), round(d["value"],1), round(d["ms_per_linear_solve"],4), d["roofline"]["kernel"], round(d["roofline"]["frac"],5), d["roofline"]["traffic_source"], round(d["cpu_baseline"]["ms_per_step"],1))
PY
