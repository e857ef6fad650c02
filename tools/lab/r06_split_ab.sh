#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "LAB_LANES=2,1" "LAB_LANES=2,2" "LAB_LANES=1,1" "LAB_LANES=4,2"; do
  echo "== $v"
  env $v timeout -k 10 200 python tools/lab/split_spmv_timing.py 2>&1 | grep "ONE launch\|TWO" | cut -c1-80
done
