#!/bin/bash
# ONE gpurun call: x = M^-1 (A v) in one launch (blkApplyKernel<..., SPMV = true>) against SpMV + apply on config 5's stand-in (108^3) and
# configs 1 / 2's matrix (2-D Poisson 1000^2), fp64: BiCGStab's time per pass either way, then the launches' own durations from a kernel trace
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06b; mkdir -p $OUT
F=$OUT/block_spmv_inside_apply.txt
echo "== SMM_HIP_BLOCK_FUSE_SPMV=0: SpMV and apply as two launches" > $F
SMM_HIP_BLOCK_FUSE_SPMV=0 timeout -k 10 200 python tools/block_precond_timing.py --skip-global --poisson2d 1000 2>&1 | grep block_ >> $F
echo "== SMM_HIP_BLOCK_FUSE_SPMV=1: A p / A s formed inside the apply's launch" >> $F
SMM_HIP_BLOCK_FUSE_SPMV=1 timeout -k 10 200 python tools/block_precond_timing.py --skip-global --poisson2d 1000 2>&1 | grep block_ >> $F
echo "== default (one launch where the preconditioner has at most 4 blocks per CU)" >> $F
timeout -k 10 200 python tools/block_precond_timing.py --skip-global --poisson2d 1000 2>&1 | grep block_ >> $F
export SMM_HIP_BLOCK_FUSE_SPMV=1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rocprof_fuse -- python3 $GRAFT_REPO_ROOT/tools/block_precond_timing.py --skip-global --poisson2d 1000 > /dev/null 2>&1 )
echo "== kernel trace of the forced one-launch run (calls, average ns, min, max): both matrices share the kernel names" >> $F
python3 - $(ls $OUT/rocprof_fuse/*/*_kernel_stats.csv | head -1) >> $F <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if "blkApplyKernel" in row["Name"] or "spmvPattern" in row["Name"]:
        print(f'{row["Name"][:88]:88s} calls {row["Calls"]:>5s}  avg {float(row["AverageNs"]) / 1e3:7.1f} us  min {float(row["MinNs"]) / 1e3:7.1f}  max {float(row["MaxNs"]) / 1e3:7.1f}')
PY
rm -rf $OUT/rocprof_fuse
cat $F | cut -c1-250
