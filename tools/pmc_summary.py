#!/usr/bin/env python3
"""Summarises the counter CSVs written by tools/pmc_spmv.sh: per kernel name, mean of every counter over its dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "spmv"
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if pat not in k:
                    continue
                short = k.split("(")[0][-60:]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, ctrs in acc.items():
        print(k)
        for c, v in sorted(ctrs.items()):
            # first dispatches are warm-up: use the tail half
            tail = v[len(v) // 2:]
            print(f"   {c:45s} n={len(v):4d} mean={sum(tail) / len(tail):.6g}")


if __name__ == "__main__":
    main()
