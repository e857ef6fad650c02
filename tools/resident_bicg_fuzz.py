#!/usr/bin/env python3
"""GPU box: the single-launch BiCGStab (csrc/smm_resident_bicg.hip) against the loop and the oracle on random banded matrices: random
offset sets (2-16 offsets within +-range, the diagonal always present and dominant), random holes in every diagonal, ragged ends, row counts
that leave the last workgroup partly or wholly idle, constant and varying diagonals, with and without Jacobi, fp32 / fp64, 1-5 iterations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import sparse_matrix_math_amd as smm  # noqa: E402
from oracle.oracle import PRECOND_JACOBI, Oracle  # noqa: E402
from sparse_matrix_math_amd import host  # noqa: E402

smm.init(0)
oracle = Oracle()
rng = np.random.default_rng(77)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for trial in range(trials):
    dtype = (np.float32, np.float64)[trial % 2]
    rows = int(rng.integers(1500, 90000))
    if trial % 5 == 0:
        rows = 8 * 1024 * int(rng.integers(1, 6)) + int(rng.integers(-3, 4))  # around whole chunks of 2 rows x 512 lanes x 8 workgroups
    k = int(rng.integers(1, 16))
    reach = int(rng.integers(2, max(3, rows // 3)))
    offs = np.unique(np.concatenate(([0], rng.integers(-reach, reach + 1, size=k))))
    keep = rng.random((rows, len(offs))) < rng.uniform(0.5, 1.0)
    keep[:, np.searchsorted(offs, 0)] = True  # the diagonal
    r, j = np.nonzero(keep)
    c = r + offs[j]
    ok = (c >= 0) & (c < rows)
    r, j, c = r[ok], j[ok], c[ok]
    varying = trial % 3 != 0
    if varying:
        v = rng.uniform(-1, 1, len(j))
    else:
        v = rng.uniform(-1, 1, len(offs))[j]
    diag = offs[j] == 0
    rowsum = np.bincount(r, weights=np.abs(v) * ~diag, minlength=rows)
    if varying:
        v[diag] = rowsum[r[diag]] + rng.uniform(0.5, 2.0, diag.sum())
    else:
        v[diag] = float(rowsum.max()) + 1.0  # one value on the whole diagonal: constant diagonals survive
    v = v.astype(dtype)
    start = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(np.bincount(r, minlength=rows), out=start[1:])
    csr = (start, c.astype(np.int32), v)
    A = smm.CSRMatrix(rows, rows, *csr)
    A.set_kernel(3, 1)
    enc = A.pattern_info()[0]
    x_true = rng.uniform(0.5, 1.5, rows).astype(dtype)
    b = oracle.spmv(csr, 0, None, x_true)
    jac = trial % 4 >= 2
    M = A.getPreconditioner(smm.SolverPreconditioner.JACOBI) if jac else None
    _, dvals = oracle.jacobi_setup(csr)
    maxit = int(rng.integers(1, 6))
    out = {}
    for mode in (host.CG_RESIDENT_REQUIRE, host.CG_RESIDENT_OFF):
        host.bicgstab_resident(mode)
        x = np.zeros(rows, dtype=dtype)
        info = {}
        st = smm.BiCGStab(A, b.copy(), x, maxit, 1e-30, M, info=info)
        out[mode] = (int(st), info["iterations"], x.astype(np.float64))
    st_o, x_o, it_o, _ = oracle.bicgstab(csr, b, np.zeros(rows, dtype=dtype), maxit, 1e-30, PRECOND_JACOBI if jac else 0, dvals if jac else None)
    res, loop = out[host.CG_RESIDENT_REQUIRE], out[host.CG_RESIDENT_OFF]
    scale = max(1.0, float(np.max(np.abs(x_o))))
    tol_paths, tol_oracle = (3e-4, 3e-3) if dtype == np.float32 else (1e-10, 1e-9)
    d_paths = float(np.max(np.abs(res[2] - loop[2]))) / scale
    d_oracle = float(np.max(np.abs(res[2] - x_o))) / scale
    good = res[0] == loop[0] == st_o and res[1] == loop[1] == it_o and d_paths <= tol_paths and d_oracle <= tol_oracle
    bad += 0 if good else 1
    print(f"trial {trial:3d}: rows {rows:6d} offsets {len(offs):2d} reach {reach:6d} {np.dtype(dtype).name} {'varying ' if varying else 'constant'} encoding {enc} "
          f"{'jacobi' if jac else 'none  '} {maxit} its: vs loop {d_paths:.1e} vs oracle {d_oracle:.1e} {'ok' if good else 'BAD ' + str((res[:2], loop[:2], (st_o, it_o)))}", flush=True)
    if M is not None:
        M.close()
    A.close()
host.bicgstab_resident(host.CG_RESIDENT_AUTO)
print("single-launch BiCGStab fuzz:", "ALL OK" if bad == 0 else f"{bad} BAD")
sys.exit(1 if bad else 0)
