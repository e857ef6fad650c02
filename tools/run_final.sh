#!/bin/bash
# GPU box: what the driver runs at round end -- build check, gpu suite, smoke, bench
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/final_tests.log 2>&1
RC=$?
tail -n 6 $OUT/final_tests.log
echo "pytest exit $RC"
[ $RC -ne 0 ] && exit $RC
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/final_smoke.log 2>&1
echo "smoke exit $?"; tail -n 2 $OUT/final_smoke.log
timeout -k 10 600 python bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
echo "bench exit $?"
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/final_bench.json"))
r,c=d["roofline"],d["roofline_csr"]
print("value",d["value"],"ms/step",d["ms_per_step"],"| roofline",r["kernel"],r["frac"],r["avg_launch_ms"],r["traffic"],"| csr",c["kernel"],c["frac"],c["avg_launch_ms"],c["value"],c["traffic"])
print("consistent:",2*r["avg_launch_ms"]<=d["ms_per_step"],"cpu",{k:v for k,v in d.get("cpu_baseline",{}).items() if k not in ("sample","cpu_model")})
e=d["extras"]["bicgstab_convdiff108_f64"]
print({k:(round(v["create_plus_solve_ms"],2),v["iterations"],round(v.get("apply_us",0),1)) for k,v in e.items() if isinstance(v,dict) and "iterations" in v})
print(d["extras"]["spmv_laplacian512_f64"])
PY
