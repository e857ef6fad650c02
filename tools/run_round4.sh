#!/bin/bash
# GPU box: traffic counters of the bench kernels on the final sources (-> profiles/spmv_traffic.json), then the round-end sequence
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh r04b > $OUT/traffic_r04b.log 2>&1; tail -n 3 $OUT/traffic_r04b.log
python3 tools/traffic_json.py gpurun_out/traffic_r04b > $OUT/traffic_json.log 2>&1; tail -n 2 $OUT/traffic_json.log
cp profiles/spmv_traffic.json $OUT/spmv_traffic_r04b.json
bash tools/run_final.sh
