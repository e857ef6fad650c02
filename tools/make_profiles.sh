#!/bin/bash
# Regenerates the measured files under profiles/<round>/ in ONE call on the GPU box (about 3 GPU-minutes):
#   gpurun --timeout 1200 -- 'bash tools/make_profiles.sh r02'
# Everything is written under gpurun_out/profiles_<round>/ (gpurun merges that directory back); copy what should be judged into
# profiles/<round>/.  The counter passes (tools/pmc_traffic.sh, tools/pmc_spmv.sh) are separate, longer calls.
set -u
ROUND=${1:-rXX}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$ROUND
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.txt || { echo "bench failed"; exit 1; }
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof -- python3 $ROOT/bench.py --cpu-seconds 0 > $OUT/bench_under_rocprof_stdout.json 2> /dev/null ) || { echo "rocprofv3 run failed"; exit 1; }
cp $(ls $OUT/rocprof/*/*_kernel_stats.csv | head -1) $OUT/bench_kernel_stats.csv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o /tmp/membw tools/membw.hip && timeout -k 10 200 /tmp/membw > $OUT/membw.txt 2>&1 || { echo "membw failed"; exit 1; }
timeout -k 10 200 python tools/sweep_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/sweep_timing.txt || { echo "sweep_timing failed"; exit 1; }
timeout -k 10 400 python tools/configs_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/configs_timing.txt || { echo "configs_timing failed"; exit 1; }
timeout -k 10 100 python tools/cg_c2.py 2>&1 | grep -v amdgpu.ids > $OUT/cg_config2.txt || { echo "cg_c2 failed"; exit 1; }
timeout -k 10 300 python tools/spmv_sweep.py --configs 2:1,2:2,2:4,3:2 2>&1 | grep -E "matrix|family" > $OUT/spmv_sweep_c3.txt
ls -la $OUT
