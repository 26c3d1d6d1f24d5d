#!/bin/bash
# usage (GPU box): tools/stamp_run.sh <workload> [env assignments]  — runs tools/fs_probe.py on build/exp_stamp/libgsx.so
wl=$1; shift
for kv in "$@"; do export "$kv"; done
cp $GRAFT_REPO_ROOT/gtsam_petercdev_amd/csrc/libgsx.so /tmp/libgsx_product.so
cp $GRAFT_REPO_ROOT/build/exp_stamp/libgsx.so $GRAFT_REPO_ROOT/gtsam_petercdev_amd/csrc/libgsx.so
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/fs_probe.py $wl 2>&1 | grep -A100 "second factorization" | grep "^\[" > $GRAFT_REPO_ROOT/gpurun_out/r03/stamp_$wl.txt
cp /tmp/libgsx_product.so $GRAFT_REPO_ROOT/gtsam_petercdev_amd/csrc/libgsx.so
cat $GRAFT_REPO_ROOT/gpurun_out/r03/stamp_$wl.txt
