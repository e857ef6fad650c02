#!/bin/bash
# GPU box: 512^3 fp64 Laplacian SpMV against the launch knobs of the STREAM family (workgroups per CU, staging passes, kernel variant)
set -u
cd $GRAFT_REPO_ROOT
for VAR in 0 1; do for NV in 1 2; do for WG in 3 4 6 8; do
  echo -n "variant $VAR nv $NV wgs/CU $WG: "
  SMM_HIP_STREAM_VARIANT=$VAR SMM_HIP_STREAM_NV=$NV SMM_HIP_STREAM_WGS_PER_CU=$WG timeout -k 10 120 python tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 2:1 --reps 10 2>&1 | grep -E "family" | cut -c1-70 || exit 1
done; done; done
