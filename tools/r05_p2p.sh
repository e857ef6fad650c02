#!/bin/bash
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
echo "== p2p thread ranks"
GPU_MAX_HW_QUEUES=32 SMM_HIP_P2P_TIMEOUT_S=5 timeout -k 10 300 python -c "import sys; sys.path.insert(0, 'tests'); import test_gpu_dist_native as t; t.p2p_thread_rank_cases()" > $OUT/p2p_thread.txt 2>&1
echo "rc $?"; grep -v amdgpu.ids $OUT/p2p_thread.txt | tail -12 | cut -c1-1500
echo "== tests"
timeout -k 10 800 python -m pytest -q -m gpu tests/test_gpu_dist_native.py -k "peer_to_peer or rehearsal" > $OUT/p2p_tests.txt 2>&1
echo "tests rc $?"; tail -30 $OUT/p2p_tests.txt | cut -c1-600
echo "== resident + mtx"
true
echo "tests rc $?"; tail -8 $OUT/resident_tests.txt | cut -c1-400
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from write_mtx import write_mtx
from sparse_matrix_math_amd import generators as gen
write_mtx("/tmp/cdv108.mtx", gen.convdiff3d_varying(108, 0.3, dtype=np.float64), shuffle=True, seed=1)
PY
for i in 1; do
timeout -k 10 300 tests/cpp/mtx_bicgstab /tmp/cdv108.mtx none,jacobi,block_ilu0 2000 1e-8 > $OUT/second_solve_run5_$i.json 2> /dev/null
grep -o '"precond": "[a-z_0-9]*"\|"solve_s": [0-9.]*\|"iterations": [0-9]*' $OUT/second_solve_run5_$i.json | paste - - -
done
