#!/usr/bin/env python3
"""GPU box: wall time of smm_hip_precond_create (structural checks + level sets of both sweeps + ILU0 / IC0 factorisation) on matrices
held in device memory.  Run once with the current library and once with SMM_HIP_LIBRARY=tools/bin/libsmm_hip_r01.so (host-side
analysis and factorisation) for the before / after table of profiles/r02/precond_create_timing.txt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host

smm.init(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
P = smm.SolverPreconditioner


def stencil(N, lo, hi, npd):
    n, nnz = N ** 3, host.gen_stencil3d_nnz(N, N, N)
    td = torch.float64 if npd == np.float64 else torch.float32
    ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=td, device=dev)
    host.gen_stencil3d_dev(N, N, N, 6.0, lo, hi, ds, dp, dv, npd, stream)
    return n, nnz, (ds, dp, dv)


def banded(n, k, npd):
    nnz = host.gen_banded_nnz(n, k, 0x5EED, 1 << 20)
    td = torch.float64 if npd == np.float64 else torch.float32
    ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=td, device=dev)
    host.gen_banded_dev(n, k, 0x5EED, 1 << 20, ds, dp, dv, npd, stream)
    return n, nnz, (ds, dp, dv)


cases = [
    ("convection-diffusion 108^3 fp64 (config 5 stand-in)", stencil(108, -1.3, -0.7, np.float64), np.float64, (P.SYMMETRIC_GAUS_SEIDEL, P.ILU0)),
    ("Laplacian 108^3 fp64", stencil(108, -1.0, -1.0, np.float64), np.float64, (P.SYMMETRIC_GAUS_SEIDEL, P.ILU0, P.IC0)),
    ("Laplacian 256^3 fp32", stencil(256, -1.0, -1.0, np.float32), np.float32, (P.ILU0, P.IC0)),
    ("banded-random SPD 1M rows x 49 fp32 (the bench matrix's shape)", banded(1_000_000, 25, np.float32), np.float32, (P.ILU0, P.IC0)),
]
print("library:", os.environ.get("SMM_HIP_LIBRARY", "in-tree"))
for name, (n, nnz, arrays), npd, kinds in cases:
    A = smm.CSRMatrix.from_device(n, n, *arrays, npd)
    A.getNonZeroCount()
    for kind in kinds:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        M = A.getPreconditioner(kind)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        lv = M.levels()
        print(f"{name:64s} rows {n:9d} nnz {nnz:10d} {P(kind).name:22s} create {dt * 1e3:10.1f} ms   levels {lv}", flush=True)
        M.close()
    A.close()
