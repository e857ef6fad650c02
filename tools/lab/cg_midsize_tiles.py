#!/usr/bin/env python3
"""ConjugateGradient on mid-size Laplacians (where the deferred / fused forms start): ms per iteration; run with and without
SMM_HIP_MARCH_FUSE_FULL_TILES=1 to compare the tile heights of CG's launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host

smm.init(0)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
for N, dtype in ((256, np.float64), (320, np.float64), (288, np.float32), (384, np.float32)):
    td = torch.float32 if dtype == np.float32 else torch.float64
    n = N ** 3; nnz = host.gen_stencil3d_nnz(N, N, N)
    ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=td, device=dev)
    host.gen_stencil3d_dev(N, N, N, 6.0, -1.0, -1.0, ds, dp, dv, dtype, stream)
    A = smm.CSRMatrix.from_device(n, n, ds, dp, dv, dtype)
    ones = torch.ones(n, dtype=td, device=dev); b = torch.empty_like(ones); A.spmv_dev(0, None, ones, b, stream)
    x = torch.zeros_like(ones); host.cg_dev(A, b, x, x, 20, 0.0, None, stream)
    best = 1e9
    for _ in range(3):
        x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        st, it, res2 = host.cg_dev(A, b, x, x, 100, 0.0, None, stream); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{N}^3 {np.dtype(dtype).name} ({n * np.dtype(dtype).itemsize / 2**20:.0f} MB per vector): CG {best / it * 1e3:.3f} ms per iteration", flush=True)
    del A, b, x, ones, ds, dp, dv
