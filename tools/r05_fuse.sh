#!/bin/bash
# round 5: ConjugateGradient at config 4's size -- the direction formed inside the 2.5-D SpMV + x deferred, x deferred only, the eager loop
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_solvers.py tests/test_gpu_fullsize.py tests/test_gpu_march.py tests/test_gpu_dist_native.py::test_native_cg_on_slabs_of_a_big_grid_uses_the_march_kernel > $OUT/fuse_tests.txt 2>&1
echo "tests rc $?"; tail -6 $OUT/fuse_tests.txt | cut -c1-300
{ echo "== p formed in the SpMV + deferred x (default: half tiles in the fused launch)"; timeout -k 10 300 python tools/configs_timing.py 2>&1 | grep config4; echo "== SMM_HIP_MARCH_FUSE_FULL_TILES=1 (the plain launch's tiles in the fused one)"; SMM_HIP_MARCH_FUSE_FULL_TILES=1 timeout -k 10 300 python tools/configs_timing.py 2>&1 | grep config4; echo "== SMM_HIP_CG_FUSE_P=0 (deferred x only)"; SMM_HIP_CG_FUSE_P=0 timeout -k 10 300 python tools/configs_timing.py 2>&1 | grep config4; echo "== SMM_HIP_CG_LAZY_X=0 (the eager three-launch loop)"; SMM_HIP_CG_LAZY_X=0 timeout -k 10 300 python tools/configs_timing.py 2>&1 | grep config4; } | tee $OUT/cg_fuse_ab.txt | cut -c1-150
