#!/bin/bash
# GPU box: memory-side counters of the SpMV variants of tools/spmv_lab.hip (the library kernel, the access-stream ceiling kernels,
# the per-window policy sweep, two structure variants): one rocprofv3 pass per counter group, --kernel-trace only.
#   tools/pmc_lab.sh <tag>      results: gpurun_out/pmc_lab_<tag>/summary.txt
set -u
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "tcc:TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
 "fetch:FETCH_SIZE"
 "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
)
for G in library mode ceil policy; do
  for P in "${PASSES[@]}"; do
    NAME=${P%%:*}; CTR=${P#*:}
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/${G}_$NAME --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/spmv_lab 10000000 6 only=$G > $OUT/${G}_$NAME.log 2>&1
    echo "$G pass $NAME exit $?"
  done
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT Kernel > $OUT/summary.txt
cat $OUT/summary.txt | head -120
