#!/bin/bash
# the whole GPU suite exactly as the driver runs it, then the bench line
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $OUT/gpu_suite.txt 2>&1
echo "suite rc $?"; tail -12 $OUT/gpu_suite.txt | cut -c1-400
timeout -k 10 400 python bench.py > $OUT/bench_stdout.json 2> $OUT/bench_stderr.txt
echo "bench rc $?"
python - <<'PY'
import json, os
p = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05/bench_stdout.json")))
print("value", p["value"], "ms/step", p["ms_per_step"], "roofline", p["roofline"]["frac"], "csr", p["roofline_csr"]["frac"])
e = p["extras"]
for k, v in e["bicgstab_convdiff108_f64"].items():
    if isinstance(v, dict) and "solve_ms" in v:
        print(k, v["iterations"], round(v["solve_ms"], 2), round(v["create_plus_solve_ms"], 2))
for k, v in e["mtx_bicgstab"].items():
    if isinstance(v, dict) and "solve_ms" in v:
        print("mtx", k, v["iterations"], round(v["solve_ms"], 2), round(v["create_plus_solve_ms"], 2))
PY
