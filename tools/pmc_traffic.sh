#!/bin/bash
# HBM-side traffic of the SpMV kernels (and of the dot kernel as a known-byte-count calibration), one rocprofv3 run per counter group
# (--kernel-trace + --pmc only), on the GPU box.   tools/pmc_traffic.sh <tag>
#   spmv_*  the benchmark matrix, STREAM family forced (spmvTileKernel<float, 2, 13>: the roofline's kernel) and the PATTERN family AUTO picks
#   lap_*   the 512^3 fp64 Laplacian (config 4's matrix), STREAM (spmvStreamKernel<double, 1>) and PATTERN with values[] read (SMM_HIP_PATTERN_CONST=0:
#           spmvPatternMasksMarchKernel<double, 8, ..>, what a big stencil with varying coefficients gets); the constant-diagonal kernel: tools/pmc_march.sh
#   dot_*   4.000 GB read by the dot kernel: calibrates the FETCH_SIZE correction
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG
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
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/spmv_$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmv_sweep.py --configs 2:2,3:2 --reps 5 > $OUT/spmv_$NAME.log 2>&1
  echo "spmv pass $NAME exit $?"
  SMM_HIP_PATTERN_CONST=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/lap_$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype f64 --configs 2:1,3:1 --reps 5 > $OUT/lap_$NAME.log 2>&1
  echo "lap pass $NAME exit $?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/dot_$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/calib_stream.py > $OUT/dot_$NAME.log 2>&1
  echo "dot pass $NAME exit $?"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "spmvTileKernel<float" > $OUT/summary_spmv.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "spmvPatternTileKernel<float" > $OUT/summary_pattern.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "spmvStreamKernel<double" > $OUT/summary_lap_stream.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "spmvPatternMasksMarchKernel<double" > $OUT/summary_lap_pattern.txt
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT dotPartials > $OUT/summary_dot.txt
cat $OUT/summary_spmv.txt $OUT/summary_pattern.txt $OUT/summary_lap_stream.txt $OUT/summary_lap_pattern.txt $OUT/summary_dot.txt
