# (on the GPU box) final bench line with the fresh traffic file, the shard-share table, the 2-rank gloo rehearsal
out=gpurun_out/final3; mkdir -p $out
timeout -k 10 300 python bench.py > $out/bal1723_bench.json 2> $out/bal1723_bench.err && echo bal ok
timeout -k 10 300 python bench.py --workload pose3_100k > $out/pose3_100k_bench.json 2> $out/p3.err && echo p3 ok
timeout -k 10 600 bash tools/shard_share.sh $out/shard > $out/shard_share.txt 2>&1 && echo share ok
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --backend gloo > $out/n2_gloo.json 2> $out/n2_gloo.err && echo n2 ok
tail -12 $out/shard_share.txt
python - <<PY
import json
d=json.load(open("gpurun_out/final3/bal1723_bench.json")); print("bal", d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
d=json.loads(open("gpurun_out/final3/n2_gloo.json").read().strip().splitlines()[-1]); print("n2", d["ms_per_step"], d["config"]["workload"], d["scaling"], d["shard"]["trial_vs_single_gpu_max_rel_diff"], d["shard"]["single_gpu_ms_per_step"])
PY
