#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
SMM_HIP_TRACE_SETUP=1 python -c "
import sparse_matrix_math_amd as smm
smm.init(0)
" 2>&1 | grep "init"
timeout -k 10 300 python tools/configs_timing.py 2>&1 | grep -v amdgpu.ids | grep config5
timeout -k 10 300 python bench.py --cpu-seconds 0 > $OUT/bench_quick.json 2>/dev/null
python - <<'PY'
import json
p = json.load(open("gpurun_out/r05/bench_quick.json"))
print("value", p["value"], p["extras"]["spmv_stencil27_192_f64"])
PY
