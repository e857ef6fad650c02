#!/bin/bash
# r06 lab: the row-partitioned SpMV of one rank's share by form (stand-in B, 61 / 39 % split by distance): SpMV alone, the blocks alone, the loop
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/lab/split_spmv_timing.py 2>&1 | grep "launch\|alone"
timeout -k 10 200 python tools/lab/rank_loop_streams.py 1250000 65536 37500 2>&1 | grep "us per\|bit for bit"
