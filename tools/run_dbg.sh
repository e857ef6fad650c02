#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 30" -ex "info threads" --args python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -p no:cacheprovider > $OUT/dbg.log 2>&1
echo "exit $?"
grep -n "SIGABRT\|#[0-9]" $OUT/dbg.log | head -60
