#!/bin/bash
# round 6, ONE gpurun call: the whole GPU suite as the driver runs it, then the round's one-GPU measurements of the row-partitioned loop
# (tools/lab/rank_loop_streams.py, split_spmv_timing.py) and of the headline kernel's bounds (x_reuse_bound.py, row_cost.py)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.txt 2>&1; echo "suite exit $?"; tail -3 $OUT/gpu_suite.txt
timeout -k 10 200 python tools/lab/rank_loop_streams.py 1250000 65536 37500 2>&1 | grep -v amdgpu.ids > $OUT/rank_loop_streams_split.txt; cat $OUT/rank_loop_streams_split.txt | cut -c1-220
timeout -k 10 200 python tools/lab/split_spmv_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/split_spmv_forms.txt; cat $OUT/split_spmv_forms.txt | cut -c1-200
timeout -k 10 300 python tools/lab/x_reuse_bound.py 2>&1 | grep -v amdgpu.ids > $OUT/x_reuse_bound.txt
timeout -k 10 300 python tools/lab/row_cost.py 2>&1 | grep -v amdgpu.ids > $OUT/pattern_kernels_row_cost.txt
