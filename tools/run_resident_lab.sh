#!/bin/bash
# GPU box: where the time of a resident CG iteration goes -- config 2, 5000 iterations, with parts switched off
# (SMM_RESIDENT_LAB bits: 1 no foreign gathers, 2 no barrier waits, 4 no published stores, 8 no global sums; results are wrong, times are
# not), for the write-through and the plain + write-back way of publishing (tools/build_resident_lab.sh builds both libraries)
set -u
cd $GRAFT_REPO_ROOT
for TAG in "" _plain; do
  export SMM_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/bin/libsmm_hip_lab$TAG.so
  for LAB in ${LABS:-0 1 4 8 2 10 15}; do
    echo "== publish${TAG:-_writethrough} SMM_RESIDENT_LAB=$LAB"
    SMM_RESIDENT_LAB=$LAB timeout -k 10 120 python tools/cg_c2.py resident 2>&1 | grep "register-resident.*fixed 5000" || { echo failed; exit 1; }
  done
done
