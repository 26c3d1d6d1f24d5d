#!/bin/bash
# the whole GPU suite, then the three benches
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03/full_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03/full_tests.log
tail -5 gpurun_out/r03/full_tests.log
grep -q "rc 0" gpurun_out/r03/full_tests.log || exit 1
for wl in pose3_100k pose2_100k bal1723; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-secondary > gpurun_out/r03/${wl}_tree.json 2> gpurun_out/r03/${wl}_tree.err
done
python tools/bench_summary.py gpurun_out/r03/*_tree.json | grep -v "^      "
