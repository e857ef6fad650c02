#!/bin/bash
# fabric-side traffic + SQ / TCP counters of the 2.5-D constant-diagonal kernel on the 512^3 Laplacian (one rocprofv3 pass per group)
#   tools/pmc_march.sh [f64|f32]
set -u
FP=${1:-f64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_march_$FP
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "fetch:FETCH_SIZE"
 "write:WRITE_SIZE"
 "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "dram:TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
 "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE"
 "sq2:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"
 "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
)
for P in "${PASSES[@]}"; do
  NAME=${P%%:*}; CTR=${P#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmv_sweep.py --matrix poisson3d --n 512 --dtype $FP --configs 3:1 --reps 5 > $OUT/$NAME.log 2>&1
  echo "pass $NAME exit $?"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT "spmvPatternConstMarchKernel" > $OUT/summary_march.txt
cat $OUT/summary_march.txt
