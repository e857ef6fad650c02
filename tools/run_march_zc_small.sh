#!/bin/bash
# GPU box: planes per unit on mid-size grids (fewer planes -> more units for the chip's workgroup slots, more planes read twice)
set -u
cd $GRAFT_REPO_ROOT
export SMM_HIP_MARCH_MIN_ROWS=1
sweep() { timeout -k 10 100 python tools/spmv_sweep.py --matrix poisson3d --n $1 --dtype f64 --configs 3:1 --reps 200 2>&1 | grep family | awk '{print $5, $6}'; }
for N in 108 128 160; do
  echo "== $N^3 fp64 constant diagonals: gather, then march with 8 (default) / 4 / 2 planes per unit"
  SMM_HIP_CONST_MARCH=0 sweep $N; sweep $N; SMM_HIP_MARCH_ZC=4 sweep $N; SMM_HIP_MARCH_ZC=2 sweep $N
  echo "== $N^3 fp64 values read: wave, then masks march with 8 / 4 planes per unit"
  SMM_HIP_PATTERN_CONST=0 SMM_HIP_MASKS_MARCH=0 sweep $N; SMM_HIP_PATTERN_CONST=0 sweep $N; SMM_HIP_PATTERN_CONST=0 SMM_HIP_MARCH_ZC=4 sweep $N
done
