#!/usr/bin/env python3
"""GPU box: the PATTERN family's three encodings (row masks, masks + constant diagonals, dictionary codes) against the oracle on random
matrices -- bit equality with one lane per row, bit equality with STREAM at 2 / 4 lanes (away from the directly streamed tails)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sparse_matrix_math_amd as smm
from oracle.oracle import Oracle

smm.init(0)
oracle = Oracle()
rng = np.random.default_rng(77)
bad = 0
NAMES = {0: "none", 1: "masks", 2: "codes", 3: "const"}
for trial in range(90):
    dtype = (np.float32, np.float64)[trial % 2]
    rows = int(rng.integers(1, 30000))
    cols = rows if trial % 3 else int(rows + rng.integers(0, 2000))
    kind = trial % 5
    if kind in (0, 1):  # a set of diagonals, random or constant values, random holes
        k = int(rng.integers(1, 40 if kind == 0 else 90))
        offs = np.unique(rng.integers(-min(rows, 5000), min(cols, 5000), size=k))
        keep = rng.random((rows, len(offs))) < rng.uniform(0.3, 1.0)
        if trial % 7 == 0:
            keep[:: int(rng.integers(2, 50))] = False  # empty rows
        r, j = np.nonzero(keep)
        c = r + offs[j]
        ok = (c >= 0) & (c < cols)
        r, j, c = r[ok], j[ok], c[ok]
        if trial % 4 == 0:  # constant diagonals
            dv = rng.uniform(-2, 2, len(offs)).astype(dtype)
            v = dv[j]
        else:
            v = rng.uniform(-1, 1, len(c)).astype(dtype)
    else:  # random columns in a band: many offsets
        per = int(rng.integers(1, 80))
        band = int(rng.integers(per + 1, max(per + 2, min(cols, 40000))))
        r = np.repeat(np.arange(rows), per)
        c = (r + rng.integers(-band, band, size=len(r))).clip(0, cols - 1)
        rc = np.unique(np.stack([r, c], axis=1), axis=0)
        r, c = rc[:, 0], rc[:, 1]
        v = rng.uniform(-1, 1, len(c)).astype(dtype)
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    start = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=rows), out=start[1:])
    csr = (start, c.astype(np.int32), v)
    if len(c) == 0:
        continue
    A = smm.CSRMatrix(rows, cols, *csr)
    x = rng.uniform(-1, 1, cols).astype(dtype)
    lhs = rng.uniform(-1, 1, rows).astype(dtype)
    try:
        A.set_kernel(3, 1)
        enc, nk = A.pattern_info()
    except smm.SmmHipError:
        rowof = np.repeat(np.arange(rows), np.diff(start))
        distinct = len(np.unique(c.astype(np.int64) - rowof))
        ok = distinct > 65536
        bad += not ok
        print(f"trial {trial}: rows {rows} nnz {len(c)} refused; distinct offsets {distinct} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        continue
    ok = True
    for op in (0, 1, 2):
        out = np.zeros(rows, dtype=dtype)
        (A.rMult(x, out) if op == 0 else A.rMultAdd(lhs, x, out) if op == 1 else A.rMultSub(lhs, x, out))
        ok = ok and np.array_equal(out, oracle.spmv(csr, op, lhs if op else None, x))
    body = start[1:] <= start[-1] - 8200
    for lanes in (2, 4):
        A.set_kernel(2, lanes)
        want = np.zeros(rows, dtype=dtype); A.rMultSub(lhs, x, want)
        A.set_kernel(3, lanes)
        got = np.zeros(rows, dtype=dtype); A.rMultSub(lhs, x, got)
        ok = ok and np.array_equal(got[body], want[body])
    bad += not ok
    print(f"trial {trial}: rows {rows} cols {cols} nnz {len(c)} {np.dtype(dtype).name} kind {kind} -> {NAMES[enc]} ({nk} offsets) {'ok' if ok else 'MISMATCH'}", flush=True)
    A.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
