#!/bin/bash
# GPU box: preconditioner sweeps -- parity tests, per-apply timing per mode, config 5 totals per mode
set -u
TAG=${1:-sweeps}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_precond.py tests/test_gpu_misc.py -m gpu -q -p no:cacheprovider > $OUT/${TAG}_tests.log 2>&1
RC=$?; tail -n 15 $OUT/${TAG}_tests.log; echo "tests exit $RC"
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/sweep_timing.py > $OUT/${TAG}_timing.log 2>&1
RC=$?; cat $OUT/${TAG}_timing.log; echo "timing exit $RC"
[ $RC -ne 0 ] && exit $RC
for MODE in 2 3; do
  SMM_HIP_SWEEP=$MODE timeout -k 10 300 python tools/configs_timing.py > $OUT/${TAG}_configs_mode$MODE.log 2>&1
  echo "configs mode $MODE exit $?"; grep config5 $OUT/${TAG}_configs_mode$MODE.log
done
