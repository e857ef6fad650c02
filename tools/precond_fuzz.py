#!/usr/bin/env python3
"""GPU box: the device-side ILU0 / IC0 factorisation, the level sets and the block preconditioners (random block sizes and level cuts)
against the oracle on random matrices (bit equality)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import generators as gen
from oracle.oracle import Oracle
import test_gpu_resident as T
from test_gpu_precond_block import permuted

smm.init(0)
oracle = Oracle()
P = smm.SolverPreconditioner
bad = 0
rng = np.random.default_rng(2026)
for trial in range(40):
    dtype = (np.float32, np.float64)[trial % 2]
    n = int(rng.integers(3, 4000))
    maxlen = int(rng.integers(1, min(n, 90)))
    csr = gen.random_rows(n, n, 1, maxlen, seed=1000 + trial, dtype=dtype, diag_dominant=True)
    A = smm.CSRMatrix(n, n, *csr)
    rhs = rng.uniform(-1, 1, n).astype(dtype)
    e, lu = oracle.ilu0_factorize(csr)
    try:
        M = A.getPreconditioner(P.ILU0)
        ok = e == 0 and np.array_equal(M.values(), lu)
        x = np.zeros(n, dtype=dtype); M.apply(rhs, x)
        ok = ok and np.array_equal(x, oracle.ilu0_apply(csr, lu, rhs)[1])
        S = A.getPreconditioner(P.SYMMETRIC_GAUS_SEIDEL); S.apply(rhs, x)
        ok = ok and np.array_equal(x, oracle.sgs_apply(csr, rhs)[1])
    except smm.SmmHipError as err:
        ok = e != 0
    bad += not ok
    print(f"ilu0/sgs trial {trial}: n {n} maxlen {maxlen} {np.dtype(dtype).name} oracle err {e} -> {'ok' if ok else 'MISMATCH'}", flush=True)
for trial in range(20):
    dtype = (np.float32, np.float64)[trial % 2]
    n = int(rng.integers(10, 6000))
    per_row = int(rng.integers(2, 40))
    csr = T.random_spd(n, per_row, 500 + trial, dtype)
    A = smm.CSRMatrix(n, n, *csr)
    rhs = rng.uniform(-1, 1, n).astype(dtype)
    e, ic = oracle.ic0_factorize(csr)
    try:
        M = A.getPreconditioner(P.IC0)
        ok = e == 0 and np.array_equal(M.values(), ic)
        x = np.zeros(n, dtype=dtype); M.apply(rhs, x)
        ok = ok and np.array_equal(x, oracle.ic0_apply(csr, ic, rhs)[1])
    except smm.SmmHipError as err:
        ok = e != 0
    bad += not ok
    print(f"ic0 trial {trial}: n {n} per_row {per_row} maxlen {np.diff(csr[0]).max()} {np.dtype(dtype).name} oracle err {e} -> {'ok' if ok else 'MISMATCH'}", flush=True)
# block ILU0 / block SGS: random matrices, block sizes and level cuts; factor (on A's pattern) and apply bit for bit
for trial in range(60):
    dtype = (np.float32, np.float64)[trial % 2]
    n = int(rng.integers(3, 9000))
    kind = trial % 3
    if kind == 0:
        maxlen = int(rng.integers(1, min(n, 60)))
        csr = gen.random_rows(n, n, 1, maxlen, seed=3000 + trial, dtype=dtype, diag_dominant=True)
    elif kind == 1:
        csr = gen.banded_random_spd(n, k=int(rng.integers(1, 20)), seed=3000 + trial, max_offset=int(rng.integers(2, max(3, min(n, 400)))), dtype=dtype)
    else:
        side = max(2, int(round(n ** 0.5)))
        csr = gen.poisson2d(side, int(rng.integers(2, 60)), dtype=dtype)
    n = len(csr[0]) - 1
    csr = (csr[0].astype(np.int32), csr[1], csr[2])
    A = smm.CSRMatrix(n, n, *csr)
    rhs = rng.uniform(-1, 1, n).astype(dtype)
    block_rows = int(rng.choice([0, 64, 100, 256, 777, 1024, 2048]))
    cap = int(rng.choice([-1, 0, 2, 3, 5, 9, 16, 40, 300]))
    try:
        I = A.getPreconditioner(P.BLOCK_ILU0, block_rows, cap)
        bounds = I.block_bounds()
        order, brick = I.block_rows()  # (grid stencils get bricks: the oracle works on P A P^T, where the blocks are contiguous)
        pcsr, src = permuted(csr, order)
        mcsr, keep, deepest = oracle.level_cut_matrix(pcsr, bounds, I.level_cap())
        e, lu = oracle.block_ilu0_factorize(mcsr, bounds)
        ok = e == 0 and max(I.levels()) == deepest
        dev_lu = I.values()[src]
        ok = ok and np.array_equal(dev_lu[keep], lu) and np.array_equal(dev_lu[~keep], pcsr[2][~keep])
        x = np.zeros(n, dtype=dtype); I.apply(rhs, x)
        ok = ok and np.array_equal(x[order], oracle.block_ilu0_apply(mcsr, bounds, lu, rhs[order])[1])
        S = A.getPreconditioner(P.BLOCK_SGS, block_rows, cap); S.apply(rhs, x)
        ok = ok and np.array_equal(x[order], oracle.block_sgs_apply(mcsr, bounds, rhs[order])[1])
        note = f"blocks {len(bounds) - 1} brick {brick} levels {I.levels()} kept {int(keep.sum())}/{len(keep)}"
    except smm.SmmHipError as err:
        ok, note = False, f"refused: {str(err)[:60]}"  # (diagonally dominant matrices: nothing to refuse)
    bad += not ok
    print(f"block trial {trial}: n {n} {('random', 'banded', 'poisson2d')[kind]} {np.dtype(dtype).name} block_rows {block_rows} cap {cap} {note} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
