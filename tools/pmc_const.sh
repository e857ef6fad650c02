#!/bin/bash
# fabric-side traffic of the constant-diagonal SpMV kernel on the 512^3 fp64 Laplacian (same passes as tools/pmc_traffic.sh)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_const
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "fetch:FETCH_SIZE"
 "write:WRITE_SIZE"
 "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "dram:TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
)
for P in "${PASSES[@]}"; do
  NAME=${P%%:*}; CTR=${P#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/lap_$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 3:1 --reps 5 > $OUT/lap_$NAME.log 2>&1
  echo "lap pass $NAME exit $?"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "spmvPatternConstKernel<double" > $OUT/summary_lap_const.txt
cat $OUT/summary_lap_const.txt
