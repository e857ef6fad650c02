#!/bin/bash
# GPU box: whole gpu suite, then a kernel trace of the block-preconditioner timing tool
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/r03a_tests.log 2>&1
RC=$?
tail -n 30 $OUT/r03a_tests.log
echo "pytest exit $RC"
[ $RC -ne 0 ] && exit $RC
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/r03a_blockprof -o blk -- python3 $GRAFT_REPO_ROOT/tools/block_precond_timing.py --skip-global --block-rows 0 > $OUT/r03a_blockprof.txt 2>&1
echo "rocprof exit $?"
tail -n 5 $OUT/r03a_blockprof.txt
find $OUT/r03a_blockprof -name "*kernel_stats.csv" | head -1 | xargs head -40
