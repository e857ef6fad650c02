#!/bin/bash
# Regenerates the measured files under profiles/<round>/ in ONE call on the GPU box (about 5 GPU-minutes):
#   gpurun --timeout 1200 -- 'bash tools/make_profiles.sh r03'
# Everything is written under gpurun_out/profiles_<round>/ (gpurun merges that directory back); copy what should be judged into
# profiles/<round>/.  The counter passes (tools/pmc_traffic.sh) are a separate, longer call.
set -u
ROUND=${1:-rXX}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$ROUND
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.txt || { echo "bench failed"; tail -5 $OUT/bench_stderr.txt; exit 1; }
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof -- python3 $ROOT/bench.py --cpu-seconds 0 > $OUT/bench_under_rocprof_stdout.json 2> /dev/null ) || { echo "rocprofv3 run failed"; exit 1; }
cp $(ls $OUT/rocprof/*/*_kernel_stats.csv | head -1) $OUT/bench_kernel_stats.csv
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof_block -- python3 $ROOT/tools/block_precond_timing.py --skip-global --block-rows 0 > $OUT/block_under_rocprof.txt 2> /dev/null ) || { echo "rocprofv3 block run failed"; exit 1; }
cp $(ls $OUT/rocprof_block/*/*_kernel_stats.csv | head -1) $OUT/block_kernel_stats.csv
timeout -k 10 300 python tools/block_precond_timing.py --block-rows 0,512,768 --level-caps=-1,0 --poisson2d 1000 2>&1 | grep -v amdgpu.ids > $OUT/block_precond_timing.txt || { echo "block timing failed"; exit 1; }
timeout -k 10 400 python tools/configs_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/configs_timing.txt || { echo "configs_timing failed"; exit 1; }
timeout -k 10 100 python tools/cg_c2.py 2>&1 | grep -v amdgpu.ids > $OUT/cg_config2.txt || { echo "cg_c2 failed"; exit 1; }
timeout -k 10 300 python tools/spmv_sweep.py --configs 2:1,2:2,2:4,3:1,3:2,3:4 2>&1 | grep -E "matrix|family" > $OUT/spmv_sweep_c3.txt
timeout -k 10 300 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 2:1,3:1 --reps 10 2>&1 | grep -E "matrix|family" > $OUT/spmv_sweep_laplacian512.txt
rm -rf $OUT/rocprof $OUT/rocprof_block
ls -la $OUT
