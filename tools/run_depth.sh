#!/bin/bash
# block apply only: chunks requested ahead (SMM_HIP_BLOCK_DEPTH) x level cut (SMM_HIP_BLOCK_LEVEL_CAP) x block rows
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/depth.txt
: > $OUT
for CAP in 16 8; do for D in 1 2 4 8; do for BR in 0 2048; do
  echo "== cap $CAP depth $D block-rows $BR" >> $OUT
  SMM_HIP_BLOCK_LEVEL_CAP=$CAP SMM_HIP_BLOCK_DEPTH=$D timeout -k 10 120 python tools/block_apply_only.py --block-rows $BR 2>&1 | grep apply >> $OUT || exit 1
done; done; done
cat $OUT
