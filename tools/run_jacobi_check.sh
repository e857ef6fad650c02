#!/bin/bash
# GPU box: after folding the Jacobi apply into the SpMV rows -- solver / preconditioner / drop-in tests, then config 5's timings
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_gpu_solvers.py tests/test_gpu_precond.py tests/test_gpu_spmv.py tests/test_cpp_dropin.py tests/test_gpu_mtx.py tests/test_gpu_fma_flavour.py -m gpu -x -q -p no:cacheprovider > $OUT/jacobi_tests.log 2>&1
RC=$?; tail -n 15 $OUT/jacobi_tests.log; echo "pytest exit $RC"; [ $RC -eq 0 ] || exit $RC
timeout -k 10 400 python tools/configs_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/jacobi_configs.txt
