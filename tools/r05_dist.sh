#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout -k 10 1000 python -m pytest -q -m gpu tests/test_gpu_dist_native.py tests/test_gpu_distributed.py > $OUT/dist_tests.txt 2>&1
echo "tests rc $?"; tail -15 $OUT/dist_tests.txt | cut -c1-400
