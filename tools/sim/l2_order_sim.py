#!/usr/bin/env python3
"""LRU model of one XCD's 4 MB L2 under the SpMV access stream of the bench matrix (config 3): which tile ORDER minimises the x[] lines
fetched from beyond L2?  Accesses are generated at tile granularity (83 rows x 51 windows + the tile's matrix lines); `conc` tiles are
in flight together (their accesses interleave).  Prints x-line misses per (line, window) pair for each order.

    python tools/sim/l2_order_sim.py [rows_per_xcd]
"""
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, ".")
from sparse_matrix_math_amd import generators as gen

LINE = 32  # floats per 128-byte line
ROWS_TILE = 83
L2_LINES = 4 * 1024 * 1024 // 128


def simulate(order, offs, matrix_pollutes=True, conc=128, seed=0):
    """order: list of tile indices in issue order.  Returns (x line requests, x misses)."""
    rng = np.random.default_rng(seed)
    lru = OrderedDict()
    req = miss = 0
    mat_id = -1
    # tiles are issued in waves of `conc`: inside a wave the per-window accesses of the tiles interleave
    for w0 in range(0, len(order), conc):
        batch = order[w0:w0 + conc]
        acc = []
        for t in batch:
            r0 = t * ROWS_TILE
            for k, d in enumerate(offs):
                lo = (r0 + d) // LINE
                hi = (r0 + d + ROWS_TILE - 1) // LINE
                for ln in range(lo, hi + 1):
                    acc.append(ln)
            if matrix_pollutes:
                acc.extend([-1] * 256)  # 32 KB of matrix lines, each unique
        acc = np.array(acc)
        rng.shuffle(acc)
        for ln in acc:
            if ln == -1:
                mat_id -= 1
                lru[mat_id] = None
                if len(lru) > L2_LINES:
                    lru.popitem(last=False)
                continue
            req += 1
            if ln in lru:
                lru.move_to_end(ln)
            else:
                miss += 1
                lru[ln] = None
                if len(lru) > L2_LINES:
                    lru.popitem(last=False)
    return req, miss


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    offs = np.array(gen.band_offsets(10_000_000, 25, 0x5EED, 1 << 20), dtype=np.int64)
    offs = np.sort(np.concatenate([-offs, [0], offs])) + (1 << 21)  # keep line numbers positive
    ntiles = rows // ROWS_TILE
    pairs = ntiles * ROWS_TILE * len(offs) / LINE
    base = list(range(ntiles))
    for name, order, pol in [
        ("sequential, matrix pollutes", base, True),
        ("sequential, matrix bypasses L2", base, False),
    ]:
        req, miss = simulate(order, offs, pol)
        print(f"{name:50s} requests {req:9d} misses {miss:9d}  misses per (line, window) pair {miss / pairs:.3f}")
    # m fronts `gap` rows apart, advanced together
    for m, gap in [(2, 30300), (2, 26296), (2, 61905), (4, 30300), (4, 14847), (8, 14847), (8, 30300), (3, 118446 // 2)]:
        gt = gap // ROWS_TILE
        chunk = gt  # each front sweeps `gap` rows, then all fronts jump m*gap ahead
        order = []
        t0 = 0
        while t0 < ntiles:
            for j in range(chunk):
                for f in range(m):
                    t = t0 + f * gt + j
                    if t < ntiles:
                        order.append(t)
            t0 += m * gt
        for pol in (True, False):
            req, miss = simulate(order, offs, pol)
            print(f"{m} fronts {gap:6d} rows apart, pollute={pol!s:5s}          requests {req:9d} misses {miss:9d}  misses per pair {miss / pairs:.3f}")


if __name__ == "__main__":
    main()
