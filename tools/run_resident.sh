#!/bin/bash
# GPU box: the register-resident CG -- its tests, then the config-2 timing in both modes.   tools/run_resident.sh <tag>
set -u
TAG=${1:-resident}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -m gpu -x -q -p no:cacheprovider > $OUT/${TAG}_tests.log 2>&1
RC=$?
tail -n 30 $OUT/${TAG}_tests.log
echo "pytest exit $RC"
[ $RC -eq 0 ] || exit $RC
timeout -k 10 200 python tools/cg_c2.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_cg_c2.txt
RC=$?
cat $OUT/${TAG}_cg_c2.txt
exit $RC
