#!/bin/bash
# Ablation of the STREAM SpMV on the GPU box: rebuilds libsmm_hip.so with -DSMM_EXP_* switches (smm_spmv.hip) and times the same
# matrix each time, then the pure-read calibration, all on one box.   tools/ablate_spmv.sh "<switch sets separated by ;>" <spmv_sweep args...>
set -u
cd $GRAFT_REPO_ROOT
SETS=$1; shift
run() {
  touch sparse_matrix_math_amd/csrc/smm_spmv.hip
  make -C sparse_matrix_math_amd/csrc all EXTRA="$1" > /dev/null 2>&1 || { echo "build failed: $1"; return; }
  echo "== build [$1]"
  timeout -k 10 150 python tools/spmv_sweep.py "${@:2}" 2>&1 | grep -E "family"
}
run "" "$@"
IFS=';' read -ra ARR <<< "$SETS"
for S in "${ARR[@]}"; do run "$S" "$@"; done
run "" "$@" > /dev/null   # leave the tree with the product build
