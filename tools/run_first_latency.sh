#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
{
echo "# first-SpMV latency on BASELINE config 4's matrix (512^3 Laplacian fp64): round-1 library (host-built tile table: start[] copied back, host loop) vs this round's (device-built)"
timeout -k 10 300 python tools/first_spmv_latency.py tools/bin/libsmm_hip_r01.so 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/first_spmv_latency.py 2>&1 | grep -v amdgpu.ids
} > $OUT/first_spmv_latency.txt 2>&1
cat $OUT/first_spmv_latency.txt
