#!/bin/bash
# GPU box: precond tests, then create-time before (round-1 library: host analysis) / after (device analysis)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_precond.py tests/test_gpu_solvers.py tests/test_gpu_misc.py -m gpu -x -v --timeout=90 -p no:cacheprovider > $OUT/precond_tests.log 2>&1
RC=$?; tail -n 25 $OUT/precond_tests.log; echo "pytest exit $RC"; [ $RC -eq 0 ] || exit $RC
timeout -k 10 600 python tools/precond_create_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/precond_create_after.txt || { cat $OUT/precond_create_after.txt; exit 1; }
cat $OUT/precond_create_after.txt
SMM_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libsmm_hip_r01.so timeout -k 10 1000 python tools/precond_create_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/precond_create_before.txt || { cat $OUT/precond_create_before.txt; exit 1; }
cat $OUT/precond_create_before.txt
