#!/usr/bin/env python3
"""Writes a Matrix Market coordinate file from one of the package's generators -- the committed generator of the `.mtx` fixtures the
tests and `bench.py --mtx` use (BASELINE config 5: a NON-symmetric matrix from a file; the real SuiteSparse atmosmodd cannot be
fetched offline, so the stand-in is the 3-D 7-point convection-diffusion operator of SURVEY.md section 8d).

    python tools/write_mtx.py convdiff3d 24 /tmp/cd24.mtx [--shuffle] [--c 0.3]
    python tools/write_mtx.py poisson2d 32 /tmp/p32.mtx --symmetric      (lower triangle only, `symmetric` banner)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_math_amd import generators as gen  # noqa: E402


def write_mtx(path, csr, symmetric=False, shuffle=False, seed=0, pattern=False):
    start, positions, values = csr
    n = len(start) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(start))
    cols = positions.astype(np.int64)
    vals = values.astype(np.float64)
    if symmetric:
        keep = cols <= rows
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
    if shuffle:
        order = np.random.default_rng(seed).permutation(len(rows))
        rows, cols, vals = rows[order], cols[order], vals[order]
    field = "pattern" if pattern else "real"
    header = (f"%%MatrixMarket matrix coordinate {field} {'symmetric' if symmetric else 'general'}\n"
              "% written by tools/write_mtx.py\n"
              f"{n} {n} {len(rows)}\n")
    if not pattern and len(rows) > 200_000 and _write_entries_arrow(path, header, rows, cols, vals):
        return len(rows)
    with open(path, "w") as f:
        f.write(header)
        if pattern:
            np.savetxt(f, np.column_stack([rows + 1, cols + 1]), fmt="%d %d")
        else:
            lines = [f"{r + 1} {c + 1} {v!r}\n" for r, c, v in zip(rows.tolist(), cols.tolist(), vals.tolist())]
            f.writelines(lines)
    return len(rows)


def _write_entries_arrow(path, header, rows, cols, vals):
    """Millions of entries (bench.py writes config 5's 8.8 M-entry file on every run): Arrow's CSV writer, space-delimited, doubles in their
    shortest round-trip form -- seconds instead of a minute of Python string formatting.  False when pyarrow is missing."""
    try:
        import pyarrow as pa
        import pyarrow.csv as pacsv
    except ImportError:
        return False
    table = pa.table({"r": pa.array(rows + 1), "c": pa.array(cols + 1), "v": pa.array(vals)})
    with open(path, "wb") as f:
        f.write(header.encode())
        pacsv.write_csv(table, f, write_options=pacsv.WriteOptions(include_header=False, delimiter=" ", quoting_style="none"))
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=["convdiff3d", "convdiff3d_varying", "poisson2d"])
    ap.add_argument("n", type=int)
    ap.add_argument("path")
    ap.add_argument("--c", type=float, default=0.3, help="upwind asymmetry of the convection-diffusion stencil")
    ap.add_argument("--shuffle", action="store_true", help="entries in random order (a coordinate file promises no order)")
    ap.add_argument("--symmetric", action="store_true")
    args = ap.parse_args()
    csr = (gen.convdiff3d(args.n, args.c, dtype=np.float64) if args.kind == "convdiff3d" else
           gen.convdiff3d_varying(args.n, args.c, dtype=np.float64) if args.kind == "convdiff3d_varying" else gen.poisson2d(args.n, dtype=np.float64))
    if args.symmetric and args.kind != "poisson2d":
        raise SystemExit("only the Poisson matrix is symmetric")
    entries = write_mtx(args.path, csr, symmetric=args.symmetric, shuffle=args.shuffle)
    print(f"{args.path}: {len(csr[0]) - 1} rows, {entries} entries")


if __name__ == "__main__":
    main()
