#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_precond_block.py -q -x -p no:cacheprovider > gpurun_out/cap_tests.log 2>&1
RC=$?; tail -n 15 gpurun_out/cap_tests.log; [ $RC -ne 0 ] && exit $RC
timeout -k 10 400 python tools/block_precond_timing.py --skip-global --block-rows 0 --level-caps 0,32,16,12,8,6,4 2>&1 | grep -E "BiCGStab" > gpurun_out/cap_timing.txt
cat gpurun_out/cap_timing.txt
