#!/bin/bash
# round 5: the row-partitioned loops after (a) the deferred x update in ConjugateGradient, (b) sums finished without L2 write-backs --
# tests, one rank's slab of config 4 at 8 GPUs with per-kernel times, the benchmark matrix through the row-partitioned BiCGStab on one rank
set -u
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
timeout -k 10 1000 python -m pytest -x -q -m gpu tests/test_gpu_dist_native.py -k "cg" > $OUT/distlazy_tests.txt 2>&1
echo "tests rc $?"; tail -6 $OUT/distlazy_tests.txt | cut -c1-300
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/distprof -- python3 $ROOT/tools/dist_cg_timing.py > $OUT/dist_cg_timing.txt 2> /dev/null ) || { echo "rocprof failed"; exit 1; }
cp $(ls $OUT/distprof/*/*_kernel_stats.csv | head -1) $OUT/dist_cg_kernel_stats.csv
rm -rf $OUT/distprof
cut -c1-200 $OUT/dist_cg_timing.txt
timeout -k 10 400 python bench.py --dist --cpu-seconds 0 --no-extras > $OUT/bench_dist_1rank.json 2> $OUT/bench_dist_1rank.err || { echo "bench --dist failed"; tail -5 $OUT/bench_dist_1rank.err; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_dist_1rank.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus") if k in d}, d.get("roofline", {}).get("avg_launch_ms"), d.get("config", {}).get("parallelism"))
PY
