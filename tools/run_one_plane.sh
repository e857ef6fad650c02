#!/bin/bash
# GPU box: the march kernels in ONE-PLANE mode (every offset near: narrow bands, tall 2-D grids) against the kernels they would replace
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/one_plane
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for FP in f64 f32; do
  echo "== $FP 2-D Laplacian 1024 x 8192 (constant diagonals): gather kernel"; SMM_HIP_CONST_MARCH=0 timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson2d --n 1024 --ny 8192 --dtype $FP --configs 3:1 --reps 20 2>&1 | grep family
  echo "== $FP 2-D Laplacian 1024 x 8192: march kernel (one plane)"; timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson2d --n 1024 --ny 8192 --dtype $FP --configs 3:1 --reps 20 2>&1 | grep family
  echo "== $FP band of 7 random diagonals within +-1000, 8 M rows (values vary): wave kernel"; SMM_HIP_MASKS_MARCH=0 timeout -k 10 200 python tools/spmv_sweep.py --matrix banded --rows 8000000 --k 3 --max-offset 1000 --dtype $FP --configs 3:1 --reps 20 2>&1 | grep family
  echo "== $FP band of 7 random diagonals within +-1000, 8 M rows: masks march (one plane)"; timeout -k 10 200 python tools/spmv_sweep.py --matrix banded --rows 8000000 --k 3 --max-offset 1000 --dtype $FP --configs 3:1 --reps 20 2>&1 | grep family
done
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
