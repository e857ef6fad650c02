#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_solvers.py tests/test_gpu_resident.py tests/test_gpu_fullsize.py::test_config4_laplacian_512_cg > $OUT/lazy_tests.txt 2>&1
echo "tests rc $?"; tail -8 $OUT/lazy_tests.txt | cut -c1-300
{ echo "== deferred x update (default)"; timeout -k 10 300 python tools/configs_timing.py 2>&1 | grep config4; echo "== SMM_HIP_CG_LAZY_X=0"; SMM_HIP_CG_LAZY_X=0 timeout -k 10 300 python tools/configs_timing.py 2>&1 | grep config4; } | tee $OUT/cg_lazy_x_ab.txt
