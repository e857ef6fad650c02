#!/bin/bash
# round 5: the row-partitioned loops after the reduction hand-off without L2 write-backs and the two-level ticket -- tests, one rank's share of
# the benchmark matrix (1/8 and 1/4 of the rows; the full matrix) through bench.py --dist, one rank's slab of config 4
set -u
ROOT=$GRAFT_REPO_ROOT
cd $ROOT
OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT
timeout -k 10 1000 python -m pytest -x -q -m gpu tests/test_gpu_dist_native.py tests/test_gpu_distributed.py > $OUT/dist_all_tests.txt 2>&1
echo "tests rc $?"; tail -4 $OUT/dist_all_tests.txt | cut -c1-300
for rows in 1250000 2500000 10000000; do
  timeout -k 10 300 python bench.py --dist --rows $rows --steps 200 --warmup 40 --cpu-seconds 0 --no-extras > $OUT/rank_compute_$rows.json 2> $OUT/rank_compute_$rows.err || { echo "bench failed"; tail -5 $OUT/rank_compute_$rows.err; exit 1; }
done
python - <<'PY' | tee gpurun_out/r05/rank_compute.txt
import json
for rows in (1250000, 2500000, 10000000):
    d = json.loads(open(f"gpurun_out/r05/rank_compute_{rows}.json").read().strip().splitlines()[-1])
    print(rows, "rows, row-partitioned loop on a single-rank communicator: ms_per_step", round(d["ms_per_step"], 4), "(with the timing events", round(d["ms_per_step_instrumented"], 4), ") it/s", round(d["value"], 1), "SpMV launch ms", round(d["roofline"]["avg_launch_ms"], 4))
PY
timeout -k 10 300 python tools/dist_cg_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/dist_cg_timing.txt | cut -c1-220
