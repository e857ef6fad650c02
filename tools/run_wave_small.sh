#!/bin/bash
# GPU box: the wave kernel on a cache-resident stencil (108^3 fp64, values read): workgroups per CU (persistent grid vs one tile per workgroup)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/wave_small
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export SMM_HIP_PATTERN_CONST=0
{
for N in 108 160; do
for W in 4 8 12 16 24 32; do
  echo "== $N^3 f64, SMM_HIP_PATTERN_WAVE=$W"; SMM_HIP_PATTERN_WAVE=$W timeout -k 10 100 python tools/spmv_sweep.py --matrix poisson3d --n $N --dtype f64 --configs 3:1 --reps 200 2>&1 | grep family
done
echo "== $N^3 f64, STREAM 1 lane"; timeout -k 10 100 python tools/spmv_sweep.py --matrix poisson3d --n $N --dtype f64 --configs 2:1 --reps 200 2>&1 | grep family
done
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
