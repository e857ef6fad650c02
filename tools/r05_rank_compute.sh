#!/bin/bash
# round 5: what ONE rank of the 8-GPU run computes per BiCGStab iteration, measured on one GPU: the benchmark matrix at an eighth of its rows
# through the row-partitioned loop on a single-rank communicator (no communication: the rank's own kernels, launch gaps and reduction tails)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
for rows in 1250000 2500000; do
  timeout -k 10 300 python bench.py --dist --rows $rows --steps 200 --warmup 40 --cpu-seconds 0 --no-extras > $OUT/rank_compute_$rows.json 2> $OUT/rank_compute_$rows.err || { echo "bench failed"; tail -5 $OUT/rank_compute_$rows.err; exit 1; }
  timeout -k 10 300 python bench.py --rows $rows --steps 200 --warmup 40 --cpu-seconds 0 --no-extras > $OUT/rank_single_$rows.json 2> $OUT/rank_single_$rows.err || { echo "bench failed"; exit 1; }
done
python - <<'PY'
import json
for rows in (1250000, 2500000):
    for kind in ("compute", "single"):
        d = json.loads(open(f"gpurun_out/r05/rank_{kind}_{rows}.json").read().strip().splitlines()[-1])
        print(rows, kind, "ms_per_step", round(d["ms_per_step"], 4), "it/s", round(d["value"], 1), "spmv launch ms", d.get("roofline", {}).get("avg_launch_ms"), d["config"].get("spmv_family"))
PY
