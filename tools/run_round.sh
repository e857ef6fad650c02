#!/bin/bash
# GPU box: where the first SpMV of the bench matrix spends its set-up time, then the round-end sequence (tests, smoke, bench)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
SMM_HIP_TRACE_SETUP=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --cpu-seconds 0 > $OUT/trace_setup.json 2> $OUT/trace_setup.err
echo "trace exit $?"; grep "smm-hip setup" $OUT/trace_setup.err
bash tools/run_final.sh
