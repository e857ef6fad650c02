#!/usr/bin/env python3
"""BASELINE config 2 timing: CG on the 1000x1000 Poisson matrix (fp64), device-resident, converged run + fixed 500 iterations"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host
smm.init(0)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
N = 1000; n = N * N; nnz = host.gen_poisson2d_nnz(N, N)
ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=torch.float64, device=dev)
host.gen_poisson2d_dev(N, N, ds, dp, dv, np.float64, stream)
A = smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float64)
ones = torch.ones(n, dtype=torch.float64, device=dev); b = torch.empty_like(ones)
A.spmv_dev(0, None, ones, b, stream)
MODES = ((host.CG_RESIDENT_OFF, "three launches per iteration"), (host.CG_RESIDENT_REQUIRE, "register-resident, one launch"))
if "resident" in sys.argv[1:]: MODES = MODES[1:]
for mode, mname in MODES:
    host.cg_resident(mode)
    for maxit, eps, label in ((-1, 1e-6, "converged tol 1e-6"), (500, 0.0, "fixed 500"), (5000, 0.0, "fixed 5000")):
        for rep in range(3):
            x = torch.zeros(n, dtype=torch.float64, device=dev); torch.cuda.synchronize(); t0 = time.perf_counter()
            st, it, res = host.cg_dev(A, b, x, x, maxit, eps, None, stream); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"CG [{mname}] {label}: status {int(st)} iterations {it} in {dt*1e3:.1f} ms -> {it/dt:.0f} it/s, {dt/it*1e6:.1f} us/it, max|x-1| {float((x-1).abs().max()):.2e}", flush=True)
