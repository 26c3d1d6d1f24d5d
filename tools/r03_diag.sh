#!/bin/bash
mkdir -p gpurun_out/r03
( timeout -k 5 120 python -m pytest tests/test_gpu_parity.py -x -q -k "choleskyPartial or underconstrained" 2>&1 | tail -5
for cfg in "400 192 1" "400 192 16" "300 96 8" "250 48 32" "415 360 4" "200 30 64"; do
  echo "--- v2 $cfg"; timeout -k 5 60 ./build/bigfront_bench $cfg
  echo "--- v1 $cfg"; GSX_DIAG_V1=1 timeout -k 5 60 ./build/bigfront_bench $cfg
done ) > gpurun_out/r03/diag.log 2>&1
cat gpurun_out/r03/diag.log
