#!/bin/bash
# round 3: the bench lines with the committed PMC traffic of the same sources (run after tools/final_capture_r03.sh + pmc_summary)
set -o pipefail
out=gpurun_out/r03final; mkdir -p $out
timeout -k 10 400 python bench.py > $out/bal1723_bench.json 2> $out/bal1723_bench.err && echo bal ok
for w in pose3_100k pose2_100k bal49; do timeout -k 10 300 python bench.py --workload $w > $out/${w}_bench.json 2> $out/${w}_bench.err && echo $w ok; done
python - <<PY
import json
for w in ("bal1723","pose3_100k","pose2_100k","bal49"):
    d=json.load(open(f"gpurun_out/r03final/{w}_bench.json")); print(w, round(d["ms_per_step"],4), round(d["value"],1), round(d["ms_per_linear_solve"],4), d["roofline"]["kernel"], round(d["roofline"]["frac"],5), d["roofline"]["traffic_source"], d["roofline"]["traffic"], round(d["cpu_baseline"]["ms_per_step"],1))
PY
