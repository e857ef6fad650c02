#!/bin/bash
# GPU box: the block-preconditioner tests, then their timing on config 5's stand-in
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_precond_block.py -m gpu -q -x -p no:cacheprovider > $OUT/block_tests.log 2>&1
RC=$?
tail -n 40 $OUT/block_tests.log
echo "pytest exit $RC"
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python tools/block_precond_timing.py --block-rows 0,512 --poisson2d 1000 > $OUT/block_timing.txt 2>&1
RC=$?
cat $OUT/block_timing.txt
exit $RC
