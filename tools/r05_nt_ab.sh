#!/bin/bash
# round 5: the non-temporal out[] policy of the older SpMV kernels, now a REAL run-time choice (storeOut, smm_device.h): A/B per kernel
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
{
for nt in 0 1; do
  echo "== SMM_HIP_NT_OUT=$nt  benchmark matrix (10 M rows fp32): CSR tile kernel 2:2, PATTERN tile kernel 3:2"
  SMM_HIP_NT_OUT=$nt timeout -k 10 200 python tools/spmv_sweep.py --configs 2:2,3:2 --reps 30 2>&1 | grep -E "family"
  echo "== SMM_HIP_NT_OUT=$nt  512^3 fp64: CSR stream 2:1, masks WAVE kernel 3:1 (constant diagonals and the march off)"
  SMM_HIP_NT_OUT=$nt SMM_HIP_PATTERN_CONST=0 SMM_HIP_MASKS_MARCH=0 timeout -k 10 300 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 2:1,3:1 --reps 10 2>&1 | grep -E "family"
  echo "== SMM_HIP_NT_OUT=$nt  512^3 fp64: constant-diagonal GATHER kernel (march off)"
  SMM_HIP_NT_OUT=$nt SMM_HIP_CONST_MARCH=0 timeout -k 10 300 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 3:1 --reps 10 2>&1 | grep -E "family"
  echo "== SMM_HIP_NT_OUT=$nt  256^3 fp64 (out[] 134 MB): CSR stream, masks wave"
  SMM_HIP_NT_OUT=$nt SMM_HIP_PATTERN_CONST=0 SMM_HIP_MASKS_MARCH=0 timeout -k 10 300 python tools/spmv_sweep.py --matrix poisson3d --n 256 --dtype f64 --configs 2:1,3:1 --reps 20 2>&1 | grep -E "family"
done
} | tee $OUT/nt_out_ab.txt
echo "== parity with the non-temporal arm forced on every size"
SMM_HIP_NT_OUT=1 timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_spmv.py tests/test_gpu_pattern.py tests/test_gpu_fullsize.py 2>&1 | tail -3
