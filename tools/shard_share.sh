#!/bin/bash
# Compute share of rank 0 of a W-way sharded problem on one GPU (exchange stubbed), next to the single-GPU time.
# usage: tools/shard_share.sh OUTDIR
out=${1:-gpurun_out/shard}
mkdir -p "$out"
for wl in pose3_100k pose2_100k bal1723; do
  python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline > "$out/${wl}_n1.json" || exit 1
  for w in 2 4 8; do
    python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --shard-share $w > "$out/${wl}_share$w.json" || exit 1
  done
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    sh = d.get("shard") or {}
    print(os.path.basename(f), "ms/step %.3f" % d["ms_per_step"], {k: sh[k] for k in ("n_cap_fronts", "cap_doubles", "own_flops", "cap_flops", "exchange_calls_per_step", "exchange_mb_per_step") if k in sh})
PY
