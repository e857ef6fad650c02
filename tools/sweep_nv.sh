#!/bin/bash
# sweeps (lanes, staging passes) of the stream kernel on the GPU box:  tools/sweep_nv.sh [sweep args]
for NV in 1 2 3 4; do
  echo "== SMM_HIP_STREAM_NV=$NV"
  SMM_HIP_STREAM_NV=$NV timeout -k 10 200 python tools/spmv_sweep.py --configs 2:1,2:2,2:4,2:8 "$@" 2>&1 | grep -E "family|rror"
done
