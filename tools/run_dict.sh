#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_pattern.py -q -x -p no:cacheprovider > gpurun_out/dict_tests.log 2>&1
RC=$?; tail -n 25 gpurun_out/dict_tests.log; [ $RC -ne 0 ] && exit $RC
OUT=gpurun_out/dict_timing.txt
: > $OUT
for K in 32; do
  echo "== banded, 10M rows, k = $K ($((2*K+1)) diagonals), fp32: STREAM 2 lanes | PATTERN (codes) 2, 4 lanes" >> $OUT
  timeout -k 10 300 python tools/spmv_sweep.py --rows 10000000 --k $K --configs 2:2,2:4,3:2,3:4 --reps 10 2>&1 | grep -E "matrix|family" >> $OUT || exit 1
done
cat $OUT
