#!/bin/bash
# GPU box: 3-D stencil SpMV, tile walk in row order (one plane per XCD, round-robin: SMM_HIP_STRIP_TILES=0) against the strip order of
# buildStripOrder (smm_spmv.hip) with strips of S tiles; time, y checksum (must not change) and the fabric read requests of one size.
# The knob is NOT in the library: apply profiles/r02/strip_order_variant.patch first (measured in r02, slower, removed again).
#   tools/run_strip_ab.sh <tag>
set -u
TAG=${1:-strip}
OUT=$GRAFT_REPO_ROOT/gpurun_out/strip_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for REP in 1 2; do
for V in 0 2 4 8 16; do
  echo "=== SMM_HIP_STRIP_TILES=$V"
  for CFG in "512 f64" "512 f32" "384 f64" "256 f64"; do
    set -- $CFG
    SMM_HIP_STRIP_TILES=$V timeout -k 10 200 python tools/spmv_sweep.py --matrix poisson3d --n $1 --dtype $2 --configs 0:0 2>&1 | grep -E "family" | sed "s/^/n=$1 $2  /" || exit 1
  done
done
done
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
cd /tmp && export TMPDIR=/tmp
for V in 0 4; do
  SMM_HIP_STRIP_TILES=$V timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum -d $OUT/pmc_$V --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 0:0 --reps 5 > $OUT/pmc_$V.log 2>&1 || { echo "pmc $V failed"; exit 1; }
  echo "== counters, SMM_HIP_STRIP_TILES=$V"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_$V spmv
done
