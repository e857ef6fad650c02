#!/bin/bash
# GPU box: 3-D stencil SpMV with the plane-per-XCD tile order (SMM_HIP_STRIP_ORDER=0) and the strip order (=1).
# The strip order was measured in round 2 (profiles/r02/strip_order_ab.txt), found slightly slower and removed from the library again:
# this script documents how the table was made (it needs the commit that had the knob).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
{
for REP in 1 2 3; do
for V in 0 1; do
  echo "=== SMM_HIP_STRIP_ORDER=$V"
  SMM_HIP_STRIP_ORDER=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 2:1 2>&1 | grep -E "family"
  SMM_HIP_STRIP_ORDER=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f32 --configs 2:1 2>&1 | grep -E "family"
  SMM_HIP_STRIP_ORDER=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 256 --dtype f64 --configs 2:1 2>&1 | grep -E "family"
  SMM_HIP_STRIP_ORDER=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 384 --dtype f64 --configs 2:1 2>&1 | grep -E "family"
done
done
} > $OUT/strip_ab.log 2>&1
cat $OUT/strip_ab.log
timeout -k 10 600 python -m pytest tests/test_gpu_spmv.py tests/test_gpu_fullsize.py tests/test_gpu_property.py tests/test_gpu_distributed.py tests/test_gpu_pattern.py -m gpu -q -p no:cacheprovider > $OUT/strip_tests.log 2>&1
RC=$?; tail -n 8 $OUT/strip_tests.log; echo "tests exit $RC"
exit $RC
