#!/bin/bash
# round 3: the reference's METIS ordering through the HIP path — parity tests, then ms/step beside the library's ND
set -e
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_fullsize.py -x -q -k "metis" > gpurun_out/r03/metis_tests.log 2>&1
for wl in bal1723 pose3_100k pose2_100k; do
  python bench.py --workload $wl --no-cpu-baseline --no-secondary > gpurun_out/r03/${wl}_lib.json 2> gpurun_out/r03/${wl}_lib.err
  python bench.py --workload $wl --ordering metis --no-cpu-baseline --no-secondary > gpurun_out/r03/${wl}_metis.json 2> gpurun_out/r03/${wl}_metis.err
done
python bench.py > gpurun_out/r03/default.json 2> gpurun_out/r03/default.err
tail -3 gpurun_out/r03/metis_tests.log
