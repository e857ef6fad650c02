#!/bin/bash
# GPU box: rows per lane of the 2.5-D kernel (tiles of 2048 or 1024 rows), both dtypes; the march tests under SMM_HIP_MARCH_R=4 as well
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/march_r
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
SMM_HIP_MARCH_R=4 timeout -k 10 600 python -m pytest tests/test_gpu_march.py -m gpu -q -x -p no:cacheprovider > $OUT/tests_r4.log 2>&1
RC=$?
tail -n 5 $OUT/tests_r4.log
[ $RC -ne 0 ] && exit $RC
sweep() { timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n ${N:-512} --dtype $1 --configs 3:1 --reps 10 2>&1 | grep -E "family" ; }
{
for FP in f64 f32; do
  for R in 8 4; do
    echo "== $FP rows per lane $R"; SMM_HIP_MARCH_R=$R sweep $FP || exit 1
    for W in 2 3 4 5; do echo "== $FP rows per lane $R, workgroups per CU $W"; SMM_HIP_MARCH_R=$R SMM_HIP_MARCH_WGS_PER_CU=$W sweep $FP || exit 1; done
  done
done
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
