#!/usr/bin/env python3
"""Per-kernel calls / average / total from a rocprofv3 kernel trace kept as a rocpd SQLite database (what `rocprofv3 --kernel-trace` writes when no
CSV output format is asked for).  usage: rocpd_kernel_summary.py results.db [rows]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"select s.kernel_name, count(*), avg(k.end - k.start), sum(k.end - k.start) from {kd} k join {ks} s on k.kernel_id = s.id group by s.kernel_name order by 4 desc limit {rows}"
print(f"{'kernel':90s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s}")
for name, calls, avg, total in con.execute(q):
    print(f"{name[:90]:90s} {calls:6d} {avg / 1e3:9.1f} {total / 1e6:9.1f}")
