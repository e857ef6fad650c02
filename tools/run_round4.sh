#!/bin/bash
# GPU box: march fuzz, traffic counters of the bench kernels on the final sources (-> profiles/spmv_traffic.json), then the round-end sequence
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/march_fuzz.py 80 > $OUT/march_fuzz.txt 2>&1; echo "march fuzz exit $?"; tail -n 2 $OUT/march_fuzz.txt
bash tools/pmc_traffic.sh r04e > $OUT/traffic_r04e.log 2>&1; tail -n 3 $OUT/traffic_r04e.log
python3 tools/traffic_json.py gpurun_out/traffic_r04e > $OUT/traffic_json.log 2>&1; tail -n 2 $OUT/traffic_json.log
cp profiles/spmv_traffic.json $OUT/spmv_traffic_r04e.json
bash tools/run_final.sh
SMM_HIP_TRACE_SETUP=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-seconds 0 > $OUT/trace_setup2.json 2> $OUT/trace_setup2.err
grep "smm-hip setup" $OUT/trace_setup2.err
