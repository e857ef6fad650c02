#!/bin/bash
# GPU box: the march form of the masks kernels (values[] read) -- tests, fuzz, then its time on the 512^3 stencil beside the wave kernel
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/masks_march
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_march.py tests/test_gpu_pattern.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1
RC=$?; tail -n 8 $OUT/tests.log; echo "pytest exit $RC"
[ $RC -ne 0 ] && exit $RC
timeout -k 10 600 python tools/march_fuzz.py 80 > $OUT/march_fuzz.txt 2>&1; echo "march fuzz exit $?"; tail -n 2 $OUT/march_fuzz.txt
export SMM_HIP_PATTERN_CONST=0
sweep() { timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n ${N:-512} --dtype $1 --configs 3:1 --reps 10 2>&1 | grep -E "family" ; }
{
for FP in f64 f32; do
  echo "== $FP wave kernel (SMM_HIP_MASKS_MARCH=0)"; SMM_HIP_MASKS_MARCH=0 sweep $FP || exit 1
  echo "== $FP masks march, defaults"; sweep $FP || exit 1
  for W in 1 2 3; do echo "== $FP masks march, workgroups per CU $W"; SMM_HIP_MARCH_WGS_PER_CU=$W sweep $FP || exit 1; done
  for Z in 8 32 128; do echo "== $FP masks march, planes per unit $Z"; SMM_HIP_MARCH_ZC=$Z sweep $FP || exit 1; done
  echo "== $FP masks march, plain stores (SMM_HIP_NT_OUT=0)"; SMM_HIP_NT_OUT=0 sweep $FP || exit 1
done
echo "== f64 256^3 wave"; N=256 SMM_HIP_MASKS_MARCH=0 sweep f64
echo "== f64 256^3 masks march"; N=256 sweep f64
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
