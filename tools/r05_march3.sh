#!/bin/bash
# round 5: the three-window march kernel for clustered far offsets (19- / 27-point stencils): parity, fuzz, timing against the gather kernel
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout -k 10 600 python tools/march_fuzz.py 96 2>&1 | grep -v amdgpu.ids > $OUT/march_fuzz.txt
echo "fuzz rc $?"; tail -3 $OUT/march_fuzz.txt; grep -c "March3Kernel ok" $OUT/march_fuzz.txt
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_march.py > $OUT/march3_tests.txt 2>&1
echo "tests rc $?"; tail -5 $OUT/march3_tests.txt | cut -c1-300
{ timeout -k 10 400 python tools/march3_timing.py; SMM_HIP_CONST_MARCH=0 timeout -k 10 400 python tools/march3_timing.py; } 2>&1 | grep -v amdgpu.ids > $OUT/march3_timing.txt
grep "256\^3" $OUT/march3_timing.txt
