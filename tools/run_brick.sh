#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_precond_block.py -q -x -p no:cacheprovider > gpurun_out/brick_tests.log 2>&1
RC=$?; tail -n 30 gpurun_out/brick_tests.log; [ $RC -ne 0 ] && exit $RC
timeout -k 10 400 python tools/block_precond_timing.py --skip-global --block-rows 0 --level-caps=-1,32,0 --poisson2d 1000 2>&1 | grep -E "BiCGStab" | cut -c1-330 > gpurun_out/brick_timing.txt
SMM_HIP_BLOCK_BRICKS=0 timeout -k 10 400 python tools/block_precond_timing.py --skip-global --block-rows 0 --level-caps=-1 2>&1 | grep -E "BiCGStab.*block" | cut -c1-330 | sed 's/^/contiguous: /' >> gpurun_out/brick_timing.txt
cat gpurun_out/brick_timing.txt
