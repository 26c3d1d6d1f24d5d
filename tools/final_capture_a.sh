set -o pipefail
out=gpurun_out/final2; mkdir -p $out
timeout -k 10 800 python -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -2 $out/gputest.log
timeout -k 10 300 python bench.py > $out/bal1723_bench.json 2> $out/bal1723_bench.err && echo bal ok
for w in pose3_100k pose2_100k bal49; do timeout -k 10 300 python bench.py --workload $w > $out/${w}_bench.json 2> $out/${w}_bench.err && echo $w ok; done
bash tools_prof.sh final2/prof_bal1723 --steps 20 --warmup 3 > $out/prof_bal1723.txt 2>&1 && echo prof bal ok
bash tools_prof.sh final2/prof_pose3 --workload pose3_100k --steps 20 --warmup 3 > $out/prof_pose3.txt 2>&1 && echo prof pose3 ok
python - <<PY
import json
for w in ("bal1723","pose3_100k","pose2_100k","bal49"):
    d=json.load(open(f"gpurun_out/final2/{w}_bench.json")); print(w, round(d["ms_per_step"],4), round(d["value"],1), round(d["ms_per_linear_solve"],4), d["roofline"]["kernel"], round(d["roofline"]["frac"],5), d["roofline"]["traffic_source"], round(d["cpu_baseline"]["ms_per_step"],1))
PY
