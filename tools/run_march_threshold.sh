#!/bin/bash
# GPU box: where the march kernels start to pay: cubic grids, constant diagonals and values read, march forced on / off
set -u
cd $GRAFT_REPO_ROOT
export SMM_HIP_MARCH_MIN_ROWS=1
sweep() { timeout -k 10 100 python tools/spmv_sweep.py --matrix poisson3d --n $1 --dtype $2 --configs 3:1 --reps 200 2>&1 | grep family | awk '{print $5, $6}'; }
for FP in f64 f32; do
for N in 64 80 96 108 128 144 160 200 256; do
  echo "== $N^3 $FP: const gather / const march / masks wave / masks march"
  SMM_HIP_CONST_MARCH=0 sweep $N $FP; sweep $N $FP
  SMM_HIP_PATTERN_CONST=0 SMM_HIP_MASKS_MARCH=0 sweep $N $FP; SMM_HIP_PATTERN_CONST=0 sweep $N $FP
done
done
