#!/bin/bash
# GPU box: pattern / AUTO tests, then the bench line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_pattern.py tests/test_gpu_misc.py tests/test_gpu_fullsize.py tests/test_gpu_spmv.py -m gpu -q -x -p no:cacheprovider > $OUT/r03b_tests.log 2>&1
RC=$?
tail -n 30 $OUT/r03b_tests.log
echo "pytest exit $RC"
[ $RC -ne 0 ] && exit $RC
timeout -k 10 600 python bench.py > $OUT/r03b_bench.json 2> $OUT/r03b_bench.err
echo "bench exit $?"
tail -c 6000 $OUT/r03b_bench.json
tail -n 5 $OUT/r03b_bench.err
