#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest -q -m gpu tests/test_gpu_resident_bicgstab.py > $OUT/resident_tests.txt 2>&1
rc=$?; echo "tests rc $rc"; tail -15 $OUT/resident_tests.txt
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from write_mtx import write_mtx
from sparse_matrix_math_amd import generators as gen
write_mtx("/tmp/cdv108.mtx", gen.convdiff3d_varying(108, 0.3, dtype=np.float64), shuffle=True, seed=1)
PY
SMM_HIP_TRACE_SETUP=1 timeout -k 10 300 tests/cpp/mtx_bicgstab /tmp/cdv108.mtx none,jacobi 2000 1e-8 > $OUT/second_solve_run3.json 2> $OUT/second_solve_trace3.txt
grep -o '"precond": "[a-z]*"\|"solve_s": [0-9.]*\|"iterations": [0-9]*' $OUT/second_solve_run3.json | paste - - -
timeout -k 10 300 tests/cpp/mtx_bicgstab /tmp/cdv108.mtx none,jacobi 2000 1e-8 > $OUT/second_solve_run4.json 2> /dev/null
grep -o '"precond": "[a-z]*"\|"solve_s": [0-9.]*\|"iterations": [0-9]*' $OUT/second_solve_run4.json | paste - - -
