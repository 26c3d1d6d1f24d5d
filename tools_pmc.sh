#!/bin/bash
# usage: tools_pmc.sh <outdir-under-gpurun_out> <COUNTER> [bench args...]   (run on the GPU box via gpurun)
# one hardware counter per pass (FETCH_SIZE and WRITE_SIZE need separate runs), with --kernel-trace only
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
ctr=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $out/bench.json 2> $out/err.log
ls $out/*/ | head
