#!/usr/bin/env python3
"""GPU box: the device-side ILU0 / IC0 factorisation and the level sets against the oracle on random matrices (bit equality)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import generators as gen
from oracle.oracle import Oracle
import test_gpu_resident as T

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
print("mismatches:", bad)
sys.exit(1 if bad else 0)
