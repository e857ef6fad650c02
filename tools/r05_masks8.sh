#!/bin/bash
# round 5: the masks march (values read) with one byte of mask per row -- parity tests, then the 512^3 SpMV in both precisions
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_march.py tests/test_gpu_fullsize.py tests/test_gpu_pattern.py > $OUT/masks8_tests.txt 2>&1
echo "tests rc $?"; tail -4 $OUT/masks8_tests.txt | cut -c1-300
{ SMM_HIP_PATTERN_CONST=0 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 3:1 --reps 10 2>&1 | grep -E "family";
  SMM_HIP_PATTERN_CONST=0 SMM_HIP_MARCH_MIN_ROWS=0 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f32 --configs 3:1 --reps 10 2>&1 | grep -E "family"; } | tee $OUT/masks8_timing.txt | cut -c1-200
