#!/bin/bash
# round 5: the single-launch BiCGStab -- parity tests, then the bench line (extras.bicgstab_convdiff108_f64, extras.mtx_bicgstab)
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest -q -m gpu tests/test_gpu_resident_bicgstab.py tests/test_gpu_resident.py > $OUT/resident_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; tail -15 $OUT/resident_tests.txt
timeout -k 10 600 python bench.py > $OUT/bench_resident.json 2> $OUT/bench_resident_stderr.txt
echo "bench rc $?"
python - <<'PY'
import json, os
p = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r05/bench_resident.json")))
print("value", p["value"], "ms/step", p["ms_per_step"])
e = p["extras"]
for k, v in e["bicgstab_convdiff108_f64"].items():
    if isinstance(v, dict) and "solve_ms" in v:
        print(k, v["iterations"], round(v["solve_ms"], 2), round(v["create_plus_solve_ms"], 2))
for k, v in e["mtx_bicgstab"].items():
    if isinstance(v, dict) and "solve_ms" in v:
        print("mtx", k, v["iterations"], round(v["solve_ms"], 2), round(v["create_plus_solve_ms"], 2))
PY
