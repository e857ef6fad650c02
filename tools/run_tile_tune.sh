#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
{
timeout -k 10 200 tools/bin/spmv_lab 10000000 20 only=mode 2>&1 | cut -c1-200 | grep "mode0\|mode2\|mode5 tpb256 nv6 L2 G13 wg0\|mode5 tpb512 nv3\|mode5 tpb256 nv4 L4 G13"
for ENVS in "SMM_HIP_STREAM_VARIANT=0" "X=1" "SMM_HIP_STREAM_NV=5" "SMM_HIP_STREAM_NV=4" "SMM_HIP_TILE_BATCH=14" "SMM_HIP_TILE_BATCH=9" "SMM_HIP_STREAM_WGS_PER_CU=2"; do
  echo "=== $ENVS"
  env $ENVS timeout -k 10 200 python tools/spmv_sweep.py --matrix banded --configs 2:2,2:4 2>&1 | grep -E "family"
done
} > $OUT/tile_tune.log 2>&1
cat $OUT/tile_tune.log
