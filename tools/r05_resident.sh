#!/bin/bash
# round 5: the single-launch BiCGStab -- parity tests, then config 5's timings (tools/block_precond_timing.py is the bench's extras leg)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout -k 10 600 python -m pytest -q -m gpu tests/test_gpu_resident_bicgstab.py tests/test_gpu_resident.py > $OUT/resident_tests.txt 2>&1
echo "tests rc $?"; tail -5 $OUT/resident_tests.txt | cut -c1-300
timeout -k 10 300 python bench.py --cpu-seconds 0 > $OUT/bench_resident.json 2> /dev/null
python - <<'PY'
import json, os
p = json.load(open("gpurun_out/r05/bench_resident.json"))
print("value", p["value"], "ms/step", p["ms_per_step"])
e = p["extras"]
for k, v in e["bicgstab_convdiff108_f64"].items():
    if isinstance(v, dict) and "solve_ms" in v:
        print(k, v["iterations"], round(v["solve_ms"], 2), round(v["create_plus_solve_ms"], 2))
for k, v in e["mtx_bicgstab"].items():
    if isinstance(v, dict) and "solve_ms" in v:
        print("mtx", k, v["iterations"], round(v["solve_ms"], 2), round(v["create_plus_solve_ms"], 2))
PY
