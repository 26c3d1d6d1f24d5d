#!/bin/bash
# round 3: subtree-fused elimination (front_tree_kernel) — parity, then ms/step
set -e
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_boundary.py -x -q > gpurun_out/r03/tree_tests.log 2>&1 || { tail -30 gpurun_out/r03/tree_tests.log; exit 1; }
tail -3 gpurun_out/r03/tree_tests.log
for wl in pose3_100k pose2_100k bal1723; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary > gpurun_out/r03/${wl}_tree.json 2> gpurun_out/r03/${wl}_tree.err
done
