#!/bin/bash
# XCD chunk-dealing sweep on the GPU box:  tools/sweep_chunk.sh "<chunk values>" <spmv_sweep args>
CH=$1; shift
for C in $CH; do
  echo "== SMM_HIP_XCD_CHUNK_TILES=$C"
  SMM_HIP_XCD_CHUNK_TILES=$C timeout -k 10 200 python tools/spmv_sweep.py "$@" 2>&1 | grep -E "family|rror"
done
echo "== heuristic"
timeout -k 10 200 python tools/spmv_sweep.py "$@" 2>&1 | grep -E "family|rror"
