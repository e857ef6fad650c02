#!/bin/bash
# ablation of the PATTERN row-per-lane kernel on the 512^3 fp64 Laplacian: how many gathers are real, out[] stored or not
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/lap_dbg.txt
: > $OUT
for F in 0 1 3 5 15 16 31; do
  echo "== DBG_FLAGS $F (low 4 bits: real gathers per row, 0 = all, 15 = none; 16 = no out[] store)" >> $OUT
  SMM_HIP_DBG_FLAGS=$F timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 3:1 --reps 10 2>&1 | grep -E "family" >> $OUT || exit 1
done
cat $OUT
