#!/bin/bash
# GPU box: march tests + fuzz tools on the final sources, then the profiles of the round (bench under rocprofv3 --stats etc.)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_march.py -m gpu -q -x -p no:cacheprovider > $OUT/march_tests.log 2>&1
RC=$?; tail -n 4 $OUT/march_tests.log; [ $RC -ne 0 ] && exit $RC
timeout -k 10 600 python tools/march_fuzz.py 80 > $OUT/march_fuzz.txt 2>&1; echo "march fuzz exit $?"; tail -n 3 $OUT/march_fuzz.txt
timeout -k 10 600 python tools/pattern_fuzz.py > $OUT/pattern_fuzz.txt 2>&1; echo "pattern fuzz exit $?"; tail -n 2 $OUT/pattern_fuzz.txt
bash tools/make_profiles.sh r04 > $OUT/make_profiles.log 2>&1; echo "profiles exit $?"; tail -n 3 $OUT/make_profiles.log
