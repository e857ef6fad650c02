#!/bin/bash
# GPU box: the STREAM family's two kernels side by side (SMM_HIP_STREAM_VARIANT=0: pipelined spmvStreamKernel; default: spmvTileKernel)
# on the matrices of BASELINE.json, then the GPU test-suite on the new default.
set -u
TAG=${1:-ab}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for V in 0 1; do
  echo "=== SMM_HIP_STREAM_VARIANT=$V"
  SMM_HIP_STREAM_VARIANT=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix banded --configs 2:1,2:2,2:4 2>&1 | grep -E "matrix|family"
  SMM_HIP_STREAM_VARIANT=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 2:1 2>&1 | grep -E "matrix|family"
  SMM_HIP_STREAM_VARIANT=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f32 --configs 2:1 2>&1 | grep -E "matrix|family"
  SMM_HIP_STREAM_VARIANT=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 256 --dtype f64 --configs 2:1 2>&1 | grep -E "matrix|family"
  SMM_HIP_STREAM_VARIANT=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson2d --n 1000 --dtype f64 --configs 2:1 --reps 200 2>&1 | grep -E "matrix|family"
  SMM_HIP_STREAM_VARIANT=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson2d --n 4000 --dtype f64 --configs 2:1 --reps 50 2>&1 | grep -E "matrix|family"
done
} > $OUT/${TAG}_ab.log 2>&1
cat $OUT/${TAG}_ab.log
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/${TAG}_tests.log 2>&1
RC=$?; tail -n 15 $OUT/${TAG}_tests.log; echo "tests exit $RC"
exit $RC
