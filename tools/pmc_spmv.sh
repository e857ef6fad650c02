#!/bin/bash
# PMC passes for the SpMV kernel on the GPU box (separate rocprofv3 runs per counter group; --kernel-trace only).
#   tools/pmc_spmv.sh <tag> <sweep args...>      results: gpurun_out/pmc_<tag>/<pass>/...csv
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE"
 "sq2:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"
 "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "fetch:FETCH_SIZE"
 "write:WRITE_SIZE"
 "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "ta:TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
 "ea:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
)
# PMC_ONLY="tcc ea": run only these passes
for P in "${PASSES[@]}"; do
  NAME=${P%%:*}; CTR=${P#*:}
  if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $NAME "; then continue; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/spmv_sweep.py "$@" > $OUT/$NAME.log 2>&1
  echo "pass $NAME exit $?"
done
find $OUT -name "*.csv" | head -30
