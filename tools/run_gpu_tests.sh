#!/bin/bash
# GPU box: pytest -m gpu on the given test files (default: all), log under gpurun_out/
set -u
TAG=${1:-gputests}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest ${@:-tests} -m gpu -q -p no:cacheprovider > $OUT/$TAG.log 2>&1
RC=$?
tail -n 40 $OUT/$TAG.log
echo "pytest exit $RC"
exit $RC
