#!/bin/bash
# GPU box: A/B of SpMV kernel builds / knobs on the benchmark matrix.  Every configuration runs in a process of its own (fresh
# allocations: run-to-run spread on one box is +-3 %), 5 times, interleaved; the table gives the median.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
CFGS=("SMM_HIP_STREAM_VARIANT=0" "SMM_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libsmm_hip_v1.so" "X=1" "SMM_HIP_STREAM_NV=6" "SMM_HIP_TILE_BATCH=14" "SMM_HIP_STREAM_WGS_PER_CU=2")
: > $OUT/tile_tune_raw.log
for REP in 1 2 3 4 5; do
  for I in "${!CFGS[@]}"; do
    MS=$(env ${CFGS[$I]} timeout -k 10 200 python tools/spmv_sweep.py --matrix banded --configs 2:2 --reps 40 2>&1 | grep -E "family" | awk '{print $5}')
    echo "$I $MS" >> $OUT/tile_tune_raw.log
  done
done
python3 - <<PY > $OUT/tile_tune.log
import statistics, collections
cfgs = """${CFGS[@]}""".split()
d = collections.defaultdict(list)
for line in open("$OUT/tile_tune_raw.log"):
    i, ms = line.split()
    d[int(i)].append(float(ms))
for i in sorted(d):
    v = sorted(d[i])
    print(f"{cfgs[i]:70s} median {statistics.median(v):.4f} ms  min {v[0]:.4f}  max {v[-1]:.4f}  ({3.996419572 / statistics.median(v) / 8 * 100:.1f} % of 8 TB/s at the median)")
PY
cat $OUT/tile_tune.log
