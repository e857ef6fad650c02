#!/bin/bash
# HBM-side traffic of the SpMV kernel (and of the dot kernel as a known-byte-count calibration), one rocprofv3 run per
# counter group (--kernel-trace only), on the GPU box.   tools/pmc_traffic.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "fetch:FETCH_SIZE"
 "write:WRITE_SIZE"
 "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "dram:TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_HIT_sum TCC_MISS_sum"
)
for P in "${PASSES[@]}"; do
  NAME=${P%%:*}; CTR=${P#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/spmv_$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmv_sweep.py --configs 0:0 --reps 5 > $OUT/spmv_$NAME.log 2>&1
  echo "spmv pass $NAME exit $?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/dot_$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/calib_stream.py > $OUT/dot_$NAME.log 2>&1
  echo "dot pass $NAME exit $?"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT spmv > $OUT/summary_spmv.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT dotPartials > $OUT/summary_dot.txt
cat $OUT/summary_spmv.txt $OUT/summary_dot.txt
