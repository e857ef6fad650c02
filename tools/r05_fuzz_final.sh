#!/bin/bash
# round 5: the fuzzers from the FINAL sources (the artefacts under profiles/r05 must not predate the last kernel-side change)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
timeout -k 10 600 python tools/march_fuzz.py 96 > $OUT/march_fuzz.txt 2>&1; echo "march_fuzz rc $?"; tail -2 $OUT/march_fuzz.txt | cut -c1-200
timeout -k 10 600 python tools/cg_fuse_check.py > $OUT/cg_fuse_check.txt 2>&1; echo "cg_fuse_check rc $?"; tail -2 $OUT/cg_fuse_check.txt | cut -c1-200
timeout -k 10 900 python tools/resident_bicg_fuzz.py 60 > $OUT/resident_bicg_fuzz.txt 2>&1; echo "resident fuzz rc $?"; tail -2 $OUT/resident_bicg_fuzz.txt | cut -c1-200
timeout -k 10 300 python tools/march3_timing.py > $OUT/march3_timing.txt 2>&1; echo "march3 rc $?"; tail -3 $OUT/march3_timing.txt | cut -c1-200
