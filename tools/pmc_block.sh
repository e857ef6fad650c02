#!/bin/bash
# Fabric-side traffic of the block preconditioners' apply launches on config 5's stand-in (108^3 fp64): the 64-lane apply and the
# one-launch M^-1 (A v), one rocprofv3 run per counter group (--kernel-trace + --pmc only), ONE gpurun call.   tools/pmc_block.sh <tag>
# Read requests are priced by their size classes (128 / 64 / 32 bytes: the records are 8-byte-per-lane loads, for which FETCH_SIZE is
# not calibrated -- MI355X_MICROARCH.md, HBM); dispatches that found the solve finished (a few microseconds, no traffic) are left out by
# taking the median of the upper half.
set -u
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_block_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SMM_HIP_BLOCK_FUSE_SPMV=1
PASSES=(
 "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "write:WRITE_SIZE"
 "dram:TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"
)
for P in "${PASSES[@]}"; do
  NAME=${P%%:*}; CTR=${P#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $CTR -d $OUT/blk_$NAME --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/block_precond_timing.py --skip-global > $OUT/blk_$NAME.log 2>&1
  echo "block pass $NAME exit $?"
done
python3 - $OUT > $OUT/summary_block.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row.get("Kernel_Name", "")
        if "blkApplyKernel" in k:
            acc[k.split("(")[0][-62:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
def med_upper(v):
    v = sorted(v)[len(v) // 2:]
    return v[len(v) // 2]
for k, c in sorted(acc.items()):
    m = {n: med_upper(v) for n, v in c.items()}
    print(k, " dispatches", len(next(iter(c.values()))))
    for n in sorted(m): print(f"   {n:28s} {m[n]:.6g}")
    if "TCC_EA0_RDREQ_sum" in m:
        b128, b64, b32 = m.get("TCC_EA0_RDREQ_128B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_32B_sum", 0)
        rest = m["TCC_EA0_RDREQ_sum"] - b128 - b64 - b32
        print(f"   read bytes by request size: {(128 * b128 + 64 * b64 + 32 * b32 + 64 * max(rest, 0)) / 1e6:.1f} MB (requests outside the three classes, priced at 64 B: {rest:.4g})")
    if "WRITE_SIZE" in m: print(f"   WRITE_SIZE: {m['WRITE_SIZE'] * 1024 / 1e6:.1f} MB if the unit is KiB")
PY
cat $OUT/summary_block.txt
