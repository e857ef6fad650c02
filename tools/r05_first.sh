#!/bin/bash
# round 5, first GPU call: the new parity tests of this round's plumbing fixes + where the second host-pointer solve loses its time
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd $ROOT
echo "== tests"
timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_dist_native.py::test_one_sided_halo tests/test_gpu_dist_native.py::test_halo_in_pieces \
  tests/test_gpu_march.py::test_march_lds_is_raised_again_for_a_larger_halo tests/test_gpu_fullsize.py::test_config4_laplacian_512_fp32_masks_march \
  tests/test_gpu_misc.py > $OUT/first_tests.txt 2>&1
echo "tests rc $?"; tail -5 $OUT/first_tests.txt
echo "== second solve"
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from write_mtx import write_mtx
from sparse_matrix_math_amd import generators as gen
write_mtx("/tmp/cdv108.mtx", gen.convdiff3d_varying(108, 0.3, dtype=np.float64), shuffle=True, seed=1)
PY
for i in 1 2; do
  SMM_HIP_TRACE_SETUP=1 timeout -k 10 300 tests/cpp/mtx_bicgstab /tmp/cdv108.mtx none,jacobi 2000 1e-8 > $OUT/second_solve_run$i.json 2> $OUT/second_solve_trace$i.txt
  echo "run $i rc $?"; grep -o '"precond": "[a-z]*"\|"solve_s": [0-9.]*' $OUT/second_solve_run$i.json | paste - -
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $OUT/prof_second -- $ROOT/tests/cpp/mtx_bicgstab /tmp/cdv108.mtx none,jacobi 2000 1e-8 > $OUT/second_solve_under_rocprof.json 2> /dev/null )
echo "rocprof rc $?"
ls $OUT/prof_second/* | head
# keep the summaries, drop the big traces if they are huge
du -sh $OUT/prof_second
