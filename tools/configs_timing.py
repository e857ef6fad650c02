#!/usr/bin/env python3
"""Timings of BASELINE configs 4 (single-GPU slice) and 5 on one MI355X (not the bench headline; numbers quoted in DESIGN.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host

smm.init(0)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
P = smm.SolverPreconditioner


def stencil(N, diag, lo, hi, dtype):
    n = N ** 3; nnz = host.gen_stencil3d_nnz(N, N, N)
    td = torch.float32 if dtype == np.float32 else torch.float64
    ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=td, device=dev)
    host.gen_stencil3d_dev(N, N, N, diag, lo, hi, ds, dp, dv, dtype, stream)
    return smm.CSRMatrix.from_device(n, n, ds, dp, dv, dtype), n, nnz, td


# ---- config 4: 3-D 7-point Laplacian 512^3, fp64, CG, fixed 100 iterations ----
for dtype in (np.float64, np.float32):
    A, n, nnz, td = stencil(512, 6.0, -1.0, -1.0, dtype)
    s = np.dtype(dtype).itemsize
    ones = torch.ones(n, dtype=td, device=dev); b = torch.empty_like(ones); A.spmv_dev(0, None, ones, b, stream)
    x = torch.zeros_like(ones)
    host.cg_dev(A, b, x, x, 10, 0.0, None, stream)
    x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    st, it, res2 = host.cg_dev(A, b, x, x, 100, 0.0, None, stream); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bytes_spmv = nnz * (s + 4) + (n + 1) * 4 + 2 * n * s
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    y = torch.empty_like(ones); e0.record()
    for _ in range(10): A.spmv_dev(0, None, ones, y, stream)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
    print(f"config4 512^3 {np.dtype(dtype).name}: CG {it} iterations in {dt*1e3:.0f} ms = {it/dt:.1f} it/s ({dt/it*1e3:.2f} ms/it); SpMV {ms:.3f} ms (family, lanes {A.get_kernel()}, PATTERN encoding {A.pattern_info()[0]}: 3 = constant diagonals, {A.kernel_desc()[0]} moves {A.kernel_desc()[1] / 1e9:.2f} GB; the CSR layout holds {bytes_spmv/1e9:.2f} GB)")
    del A, b, x, y, ones

# ---- config 5 stand-in: convection-diffusion 108^3, fp64, BiCGStab none / Jacobi / ILU0 / SGS to 1e-8 ----
A, n, nnz, td = stencil(108, 6.0, -1.3, -0.7, np.float64)
ones = torch.ones(n, dtype=td, device=dev); b = torch.empty_like(ones); A.spmv_dev(0, None, ones, b, stream)
for name, kind in (("none", None), ("jacobi", P.JACOBI), ("ilu0", P.ILU0), ("sgs", P.SYMMETRIC_GAUS_SEIDEL)):
    t0 = time.perf_counter(); M = A.getPreconditioner(kind) if kind is not None else None; tc = time.perf_counter() - t0
    x = torch.zeros_like(ones); torch.cuda.synchronize(); t0 = time.perf_counter()
    st, it, res = host.bicgstab_dev(A, b, x, -1, 1e-8, M, stream); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lv = M.levels() if M is not None else (0, 0)
    print(f"config5 108^3 f64 BiCGStab+{name}: create {tc*1e3:.0f} ms, levels {lv}, {it} iterations in {dt*1e3:.1f} ms ({dt/max(it,1)*1e3:.3f} ms/it), max|x-1| {float((x-1).abs().max()):.1e}")
