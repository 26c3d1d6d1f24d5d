#!/bin/bash
mkdir -p gpurun_out/r03
GSX_DEBUG_LAUNCH=1 timeout -k 10 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q -s -k "wide_variables" > gpurun_out/r03/dbg.log 2>&1
echo "rc $?" >> gpurun_out/r03/dbg.log
tail -40 gpurun_out/r03/dbg.log
