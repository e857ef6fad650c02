#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05; mkdir -p $OUT
for nt in 0 1 0 1; do
  SMM_HIP_NT_OUT=$nt timeout -k 10 300 python bench.py --cpu-seconds 0 > $OUT/bench_nt$nt.json 2>/dev/null
  python - <<PY
import json
p = json.load(open("$OUT/bench_nt$nt.json"))
print("NT_OUT=$nt auto leg it/s %.1f ms/step %.4f spmv %.4f ms | csr leg it/s %.1f ms/step %.4f spmv %.4f ms frac %.3f" % (p["value"], p["ms_per_step"], p["roofline"]["avg_launch_ms"], p["roofline_csr"]["value"], p["roofline_csr"].get("ms_per_step", -1), p["roofline_csr"]["avg_launch_ms"], p["roofline_csr"]["frac"]))
e = p["extras"]
l = e["spmv_laplacian512_f64"]
print("   512^3 f64: CSR stream %.4f ms" % l["avg_launch_ms"])
PY
done 2>&1 | tee $OUT/nt_out_bench_ab.txt
