#!/usr/bin/env python3
"""Per-VARIANT counter table of a tools/pmc_lab.sh run.  Variants that share a kernel (the policy / flavour sweeps) are told apart by
dispatch order: spmv_lab launches every variant 1 + reps times in the order of the table it prints, so dispatches are cut into groups of
that size and labelled with the names of the run's own log.

    python tools/pmc_lab_summary.py gpurun_out/pmc_lab_<tag> [reps=6]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    per = reps + 1
    for group in ("library", "mode", "ceil", "flavor", "policy"):
        logs = sorted(glob.glob(os.path.join(root, f"{group}_*.log")))
        if not logs:
            continue
        names = []
        for line in open(logs[0]):
            m = re.match(r"(.+?)\s+vgpr\s+\d+", line)
            if m:
                names.append((m.group(1).strip(), re.search(r"avg ([0-9.]+) ms", line).group(1)))
        table = defaultdict(dict)
        for p in sorted(glob.glob(os.path.join(root, f"{group}_*"))):
            if not os.path.isdir(p):
                continue
            rows = []
            for path in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        k = row.get("Kernel_Name", "")
                        if "Kernel<" not in k or "smm::" in k:  # the lab's own kernels only
                            continue
                        rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"])))
            by_counter = defaultdict(list)
            for d, c, v in sorted(rows):
                by_counter[c].append(v)
            for c, vals in by_counter.items():
                for g in range(len(vals) // per):
                    tail = vals[g * per + 1:(g + 1) * per]  # the first launch of a variant is its correctness run
                    table[g][c] = sum(tail) / len(tail)
        print(f"== {group}: mean over the {reps} timed launches (un-profiled time of the same run in ms)")
        for g in sorted(table):
            name, ms = names[g] if g < len(names) else (f"variant {g}", "?")
            c = table[g]
            rd = c.get("TCC_EA0_RDREQ_sum", 0)
            print(f"{name:52s} {ms:>7s} ms  EA_RDREQ {rd / 1e6:7.2f} M = {rd * 128 / 1e9:5.2f} GB  TCC_HIT {c.get('TCC_HIT_sum', 0) / 1e6:6.2f} M  "
                  f"TCC_MISS {c.get('TCC_MISS_sum', 0) / 1e6:6.2f} M  TCP->TCC {c.get('TCP_TCC_READ_REQ_sum', 0) / 1e6:6.2f} M")


if __name__ == "__main__":
    main()
