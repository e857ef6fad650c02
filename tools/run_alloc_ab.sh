#!/bin/bash
# GPU box: does the memory type of the matrix stream change what the L2 keeps of x?  The lab (tools/spmv_lab.hip) allocates
# positions[] / values[] as default (coarse-grained), fine-grained (1) or uncached (3) device memory; time of the library's kernel
# and of the access-stream ceiling kernels, then one TCC counter pass each.   tools/run_alloc_ab.sh <tag>
set -u
TAG=${1:-alloc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/alloc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for F in 0 3 1; do
  export LAB_MATRIX_ALLOC=$F
  timeout -k 10 200 $GRAFT_REPO_ROOT/tools/bin/spmv_lab 10000000 20 only=library > $OUT/time_library_$F.log 2>&1 || { echo "library alloc $F failed"; tail -5 $OUT/time_library_$F.log; exit 1; }
  timeout -k 10 200 $GRAFT_REPO_ROOT/tools/bin/spmv_lab 10000000 20 only=ceil > $OUT/time_ceil_$F.log 2>&1 || { echo "ceil alloc $F failed"; tail -5 $OUT/time_ceil_$F.log; exit 1; }
  echo "== LAB_MATRIX_ALLOC=$F"; grep -h "avg" $OUT/time_library_$F.log $OUT/time_ceil_$F.log | cut -c1-150
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_$F --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/spmv_lab 10000000 6 only=library > $OUT/pmc_$F.log 2>&1 || { echo "pmc alloc $F failed"; tail -5 $OUT/pmc_$F.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmc_$F spmv > $OUT/pmc_summary_$F.txt 2>&1; cat $OUT/pmc_summary_$F.txt | cut -c1-200 | head -12
done
