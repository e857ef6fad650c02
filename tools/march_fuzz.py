#!/usr/bin/env python3
"""GPU box: the 2.5-D kernels (spmvPatternConstMarchKernel, spmvPatternConstMarch3Kernel, spmvPatternMasksMarchKernel) against the oracle, bit for bit, on random grid-shaped
matrices far below its production threshold (SMM_HIP_MARCH_MIN_ROWS=1 for this process): random plane sizes (tiles that are partial,
one tile per plane, planes that are no multiple of anything but the pack), random plane counts with a partial last plane, random near
offsets, one or both far offsets -- single or in clusters --, random holes in every diagonal (the masks), empty rows, fp32 / fp64, all three ops, in place."""
import os
import sys

os.environ["SMM_HIP_MARCH_MIN_ROWS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import sparse_matrix_math_amd as smm  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

smm.init(0)
oracle = Oracle()
rng = np.random.default_rng(2024)
bad = 0
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for trial in range(trials):
    dtype = (np.float32, np.float64)[trial % 2]
    vec = 4 if dtype == np.float32 else 2
    one_plane = trial % 5 == 4
    if one_plane:  # every offset near: a band inside the halo a lane can hold
        hcap = 1024 if dtype == np.float64 else 2048
        rows = int(rng.integers(3000, 60000)) // vec * vec
        nnear = int(rng.integers(1, 9))
        offs = np.unique(np.concatenate(([0], rng.integers(-hcap + 8, hcap - 8, size=nnear))))
        P = rows
    else:
        P = int(rng.integers(8192, 20000)) // vec * vec
        planes = int(rng.integers(2, 14))  # (the masks kernels march from 8 planes up; below that the wave kernel is checked instead)
        rows = (P * planes - int(rng.integers(0, P // 2)) * (trial % 3 == 0)) // vec * vec  # every third trial: a partial last plane
        nnear = int(rng.integers(1, 7))
        hmax = int(rng.integers(2, 600))
        near = np.unique(np.concatenate(([0], rng.integers(-hmax, hmax + 1, size=nnear))))
        far = [(-P,), (P,), (-P, P)][trial % 3] if trial % 7 else (-P, P)
        if trial % 8 in (3, 6):
            # far offsets in CLUSTERS around -P / +P (19- / 27-point stencil shapes): spmvPatternConstMarch3Kernel.  The plan takes the plane
            # size from the centre of a cluster, so a cluster is symmetric about its centre (holes inside are fine)
            m = int(rng.integers(1, hmax + 1))
            inner = rng.integers(-m, m + 1, size=int(rng.integers(0, 6)))
            shape = np.unique(np.concatenate(([-m, m], inner)))
            far = np.concatenate([c + shape for c in far])
        offs = np.unique(np.concatenate((near, np.array(far))))
    keep = rng.random((rows, len(offs))) < rng.uniform(0.55, 1.0)
    if trial % 6 == 0:
        keep[:: int(rng.integers(3, 40))] = False  # empty rows
    if trial % 4 == 1:
        keep[:] = True  # every row full wherever the column exists: the interior fast path
    r, j = np.nonzero(keep)
    c = r + offs[j]
    ok = (c >= 0) & (c < rows)
    r, j, c = r[ok], j[ok], c[ok]
    dv = rng.uniform(-2, 2, len(offs)).astype(dtype)
    v = dv[j]
    varying = (trial // 2) % 2 == 1 and not (not one_plane and trial % 8 in (3, 6))  # every diagonal varies: MASKS (values[] read) -- spmvPatternMasksMarchKernel (clustered far offsets: constant only)
    if varying:
        v = rng.uniform(-2, 2, len(j)).astype(dtype)
    start = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=rows), out=start[1:])
    csr = (start, c.astype(np.int32), v)
    if len(c) == 0:
        continue
    A = smm.CSRMatrix(rows, rows, *csr)
    try:
        A.set_kernel(3, 1)
    except smm.SmmHipError as e:
        print(f"trial {trial}: refused ({e})")
        A.close()
        continue
    enc = A.pattern_info()[0]
    kernel = A.kernel_desc()[0]
    x = rng.uniform(-1, 1, rows).astype(dtype)
    lhs = rng.uniform(-1, 1, rows).astype(dtype)
    okk = True
    for op in (0, 1, 2):
        out = np.full(rows, np.nan, dtype=dtype)
        {0: lambda: A.rMult(x, out), 1: lambda: A.rMultAdd(lhs, x, out), 2: lambda: A.rMultSub(lhs, x, out)}[op]()
        ref = oracle.spmv(csr, op, lhs, x)
        if not np.array_equal(out, ref):
            okk = False
            w = np.nonzero(out != ref)[0]
            print(f"trial {trial}: MISMATCH op {op} at {len(w)} rows, first {w[:5]}, rows {rows} P {P} offs {offs.tolist()} dtype {np.dtype(dtype).name} kernel {kernel}")
    z = lhs.copy()
    A.rMultSub(z, x, z)
    if not np.array_equal(z, oracle.spmv(csr, 2, lhs, x)):
        okk = False
        print(f"trial {trial}: MISMATCH in place")
    bad += 0 if okk else 1
    print(f"trial {trial:3d}: rows {rows:7d} P {P:6d} offsets {len(offs):2d} {'one plane' if one_plane else 'march    '} {np.dtype(dtype).name} {'varying ' if varying else 'constant'} encoding {enc} {kernel} {'ok' if okk else 'BAD'}", flush=True)
    A.close()
print("march fuzz:", "ALL OK" if bad == 0 else f"{bad} BAD")
sys.exit(1 if bad else 0)
