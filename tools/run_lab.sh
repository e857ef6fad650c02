#!/bin/bash
# GPU box: the SpMV variant table (tools/spmv_lab.hip).  tools/run_lab.sh <tag> [rows]
# Each group runs in its own process under its own timeout, joined with &&: a fault or hang in one stops the script.
set -u
TAG=${1:-lab}
ROWS=${2:-10000000}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for G in mode; do
  timeout -k 10 300 tools/bin/spmv_lab $ROWS 20 only=$G > $OUT/${TAG}_$G.log 2>&1
  RC=$?
  echo "$G exit $RC"
  cut -c1-230 $OUT/${TAG}_$G.log | tail -n 30
  if [ $RC -ne 0 ] || grep -q "Memory access fault" $OUT/${TAG}_$G.log; then echo "stopping after $G"; exit 1; fi
done
