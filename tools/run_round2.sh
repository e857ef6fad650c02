#!/bin/bash
# GPU box: the row-partitioned tests (halo in pieces), the traffic counters of the bench kernels (-> profiles/spmv_traffic.json), bench
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_dist_native.py tests/test_gpu_mtx.py -m gpu -q -x -p no:cacheprovider > $OUT/dist_tests.log 2>&1
RC=$?; tail -n 6 $OUT/dist_tests.log; echo "pytest exit $RC"
[ $RC -ne 0 ] && exit $RC
bash tools/pmc_traffic.sh r04 > $OUT/traffic_r04.log 2>&1; tail -n 4 $OUT/traffic_r04.log
python3 tools/traffic_json.py gpurun_out/traffic_r04 > $OUT/traffic_json.log 2>&1; tail -n 3 $OUT/traffic_json.log
cp profiles/spmv_traffic.json $OUT/spmv_traffic_r04.json
timeout -k 10 600 python bench.py > $OUT/bench_r04a.json 2> $OUT/bench_r04a.err
echo "bench exit $?"
python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/bench_r04a.json"))
r,c=d["roofline"],d["roofline_csr"]
print("value",d["value"],"ms/step",d["ms_per_step"],"| roofline",r["kernel"],r["frac"],r["avg_launch_ms"],r["traffic"],"| csr",c["kernel"],c["frac"],c["avg_launch_ms"],c["value"],c["traffic"])
print("cpu",{k:v for k,v in d.get("cpu_baseline",{}).items() if k not in ("sample","cpu_model")})
print(d["spmv_kernel"])
print(d["extras"]["spmv_laplacian512_f64"])
PY
