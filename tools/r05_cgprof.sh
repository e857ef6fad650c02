#!/bin/bash
# round 5: per-kernel times of ConjugateGradient at config 4's size (the fused SpMV', the r update, the flush of x)
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cgprof -- python3 $ROOT/tools/configs_timing.py > $OUT/cgprof_stdout.txt 2> /dev/null || { echo "rocprof failed"; exit 1; }
cp $(ls $OUT/cgprof/*/*_kernel_stats.csv | head -1) $OUT/cg_config4_kernel_stats.csv
rm -rf $OUT/cgprof
head -25 $OUT/cg_config4_kernel_stats.csv | cut -c1-260
