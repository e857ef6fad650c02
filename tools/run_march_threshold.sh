#!/bin/bash
# GPU box: where the march kernels start to pay: cubic grids of 1.26 M ... 8 M rows, constant diagonals and values read, march forced on / off
set -u
cd $GRAFT_REPO_ROOT
export SMM_HIP_MARCH_MIN_ROWS=1
sweep() { timeout -k 10 100 python tools/spmv_sweep.py --matrix poisson3d --n $1 --dtype f64 --configs 3:1 --reps 200 2>&1 | grep family | awk '{print $5, $6}'; }
for N in 108 128 144 160 200; do
  echo "== $N^3 fp64: const gather / const march / masks wave / masks march"
  SMM_HIP_CONST_MARCH=0 sweep $N; sweep $N
  SMM_HIP_PATTERN_CONST=0 SMM_HIP_MASKS_MARCH=0 sweep $N; SMM_HIP_PATTERN_CONST=0 sweep $N
done
