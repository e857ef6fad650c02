#!/bin/bash
# GPU box: what bounds spmvPatternConstKernel on the 512^3 fp64 Laplacian?  Rebuilds the library with -DSMM_EXP_CONST=<bits>
# (1 no out[] store, 2 every gather -> x[row], 4 no mask stream, 8 only the far (plane) gathers -> x[row]) and times family 3 / 1 lane
# each time on ONE box; then the counters of the product build.     tools/run_const_ablate.sh [fp: f64|f32]
set -u
FP=${1:-f64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/const_ablate_$FP
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() {
  touch sparse_matrix_math_amd/csrc/smm_spmv_pattern.hip
  make -s -C sparse_matrix_math_amd/csrc all EXTRA="$1" > $OUT/build.log 2>&1 || { echo "build failed: $1"; tail -5 $OUT/build.log; return; }
  echo "== build [$1]"
  for W in 8 4 16; do
    echo "   workgroups per CU $W"
    SMM_HIP_CONST_WGS_PER_CU=$W timeout -k 10 150 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype $FP --configs 3:1 --reps 10 2>&1 | grep -E "family" || return 1
  done
}
{
run ""
for BITS in 1 2 3 4 5 7 8 9; do run "-DSMM_EXP_CONST=$BITS" || break; done
} > $OUT/times.txt 2>&1
touch sparse_matrix_math_amd/csrc/smm_spmv_pattern.hip
make -s -C sparse_matrix_math_amd/csrc all > /dev/null 2>&1   # leave the tree with the product build
cat $OUT/times.txt
