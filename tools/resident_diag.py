#!/usr/bin/env python3
"""GPU box: how far the register-resident CG, the three-launch CG and the oracle are from each other on the cases of
tests/test_gpu_resident.py (the evidence behind that file's tolerances)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import generators as gen, host
from oracle.oracle import Oracle
import test_gpu_resident as T
smm.init(0)
oracle = Oracle()
for dtype in (np.float32, np.float64):
    for name, csr in T.cases(dtype).items():
        start, pos, val = csr
        n = len(start) - 1
        A = smm.CSRMatrix(n, n, *csr)
        b = gen.row_sums(start, val)
        x0 = np.zeros(n, dtype=dtype)
        for maxit, eps in ((1, 0.0), (7, 0.0), (40, 0.0), (-1, 1e-5 if dtype == np.float32 else 1e-9)):
            got = {}
            for mode in (2, 0):
                host.cg_resident(mode)
                x = np.full(n, 5, dtype=dtype); info = {}
                st = smm.ConjugateGradient(A, b, x0, x, maxit, eps, info=info)
                got[mode] = (int(st), info["iterations"], x.astype(np.float64))
            st_o, x_o, it_o, _ = oracle.cg(csr, b, x0, maxit, eps)
            x_o = x_o.astype(np.float64)
            d = lambda a, c: float(np.max(np.abs(a - c)))
            print(f"{np.dtype(dtype).name} {name:36s} maxit {maxit:3d}: status {got[2][0]}/{got[0][0]}/{st_o} its {got[2][1]}/{got[0][1]}/{it_o} "
                  f"|res-oracle| {d(got[2][2], x_o):.2e} |three-oracle| {d(got[0][2], x_o):.2e} |res-three| {d(got[2][2], got[0][2]):.2e} |res-1| {d(got[2][2], 1.0):.2e}", flush=True)
