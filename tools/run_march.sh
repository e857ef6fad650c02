#!/bin/bash
# GPU box: the 2.5-D constant-diagonal kernel -- its tests, then its time on the 512^3 Laplacian beside the gather kernel it replaces
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/march
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_march.py tests/test_gpu_pattern.py tests/test_gpu_misc.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1
RC=$?
tail -n 15 $OUT/tests.log
echo "pytest exit $RC"
[ $RC -ne 0 ] && exit $RC
sweep() { timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n ${N:-512} --dtype $1 --configs 3:1 --reps 10 2>&1 | grep -E "family" ; }
legs() {
for FP in f64 f32; do
  echo "== $FP march kernel, defaults"; sweep $FP || return 1
  for W in 2 3; do echo "== $FP march, workgroups per CU $W"; SMM_HIP_MARCH_WGS_PER_CU=$W sweep $FP || return 1; done
  for Z in 16 64; do echo "== $FP march, planes per unit $Z"; SMM_HIP_MARCH_ZC=$Z sweep $FP || return 1; done
  echo "== $FP march, plain stores (SMM_HIP_NT_OUT=0)"; SMM_HIP_NT_OUT=0 sweep $FP || return 1
done
}
{
echo "== f64 gather kernel (SMM_HIP_CONST_MARCH=0)"; SMM_HIP_CONST_MARCH=0 sweep f64
echo "== f32 gather kernel (SMM_HIP_CONST_MARCH=0)"; SMM_HIP_CONST_MARCH=0 sweep f32
legs
echo "==== rebuilt with -DSMM_MARCH_MIN_WAVES=3"
touch sparse_matrix_math_amd/csrc/smm_spmv_march.hip
make -s -C sparse_matrix_math_amd/csrc all EXTRA="-DSMM_MARCH_MIN_WAVES=3" > $OUT/build.log 2>&1 && legs
touch sparse_matrix_math_amd/csrc/smm_spmv_march.hip
make -s -C sparse_matrix_math_amd/csrc all > /dev/null 2>&1
echo "== f64 256^3 gather"; N=256 SMM_HIP_CONST_MARCH=0 sweep f64
echo "== f64 256^3 march"; N=256 sweep f64
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
