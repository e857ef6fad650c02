#!/bin/bash
# round 6, the last call: the whole GPU suite as the driver runs it, then smoke, then the one-launch M^-1 (A v) against SpMV + apply
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.txt 2>&1; rc=$?; echo "suite exit $rc"; tail -3 $OUT/gpu_suite.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; rc=$?; echo "smoke exit $rc"; tail -2 $OUT/smoke.txt
[ $rc -eq 0 ] || exit $rc
bash tools/lab/block_spmv_inside_apply.sh > /dev/null 2>&1; cp gpurun_out/r06b/block_spmv_inside_apply.txt $OUT/; tail -12 $OUT/block_spmv_inside_apply.txt | cut -c1-200
