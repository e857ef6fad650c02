#!/bin/bash
# GPU box: knobs of the PATTERN tile kernel on the benchmark matrix, one process per setting
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
R=$OUT/pat_tune.txt
: > $R
run() { echo "== $*" >> $R; env "$@" timeout -k 10 120 python tools/spmv_sweep.py --configs 3:2,2:2 --reps 20 2>&1 | grep -E "family" >> $R || return 1; }
run A=1 && run SMM_HIP_PATTERN_VARIANT=0 && run SMM_HIP_TILE_BATCH=8 && run SMM_HIP_TILE_BATCH=16 && \
run SMM_HIP_STREAM_WGS_PER_CU=3 && run SMM_HIP_STREAM_WGS_PER_CU=4 && run SMM_HIP_STREAM_WGS_PER_CU=6 && run SMM_HIP_STREAM_WGS_PER_CU=8
echo "== lanes 4" >> $R; timeout -k 10 120 python tools/spmv_sweep.py --configs 3:4,3:1 --reps 20 2>&1 | grep family >> $R
cat $R
