#!/bin/bash
# GPU box: what bounds the SpMV of the 512^3 fp64 Laplacian?  (a) the gathers pointed at x[row] (no far planes) and at x[row + j - len/2]
# (near), (b) LDS / instruction counters of the real thing.   tools/run_lap_ablate.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/lap_ablate
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for MODE in orig row near; do
  echo "== pos-mode $MODE"
  timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 0:0,2:1,2:2,1:4 --pos-mode $MODE 2>&1 | grep -E "family" || exit 1
done > $OUT/times.txt 2>&1
cat $OUT/times.txt
PMC_ONLY="sq2 tcp" bash tools/pmc_spmv.sh lap512b --matrix poisson3d --n 512 --dtype f64 --configs 0:0 --reps 5 > $OUT/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_lap512b spmv
