#!/usr/bin/env python3
"""Times ONE block-preconditioner apply (HIP events over `reps` back-to-back applies) on the convection-diffusion N^3 matrix -- no solve:
usable with ablation builds whose results are wrong.   python tools/block_apply_only.py [--n 108] [--block-rows 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=108)
ap.add_argument("--block-rows", type=int, default=0)
ap.add_argument("--reps", type=int, default=100)
args = ap.parse_args()
smm.init(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
N = args.n
n, nnz = N ** 3, host.gen_stencil3d_nnz(N, N, N)
ds, dp, dv = (torch.empty(n + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.float64, device=dev))
host.gen_stencil3d_dev(N, N, N, 6.0, -1.3, -0.7, ds, dp, dv, np.float64, stream)
A = smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float64)
b = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.empty_like(b)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for kind in (smm.SolverPreconditioner.BLOCK_ILU0, smm.SolverPreconditioner.BLOCK_SGS):
    M = A.getPreconditioner(kind, args.block_rows or None)
    for _ in range(3):
        M.apply_dev(b, y, stream)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.reps):
        M.apply_dev(b, y, stream)
    e1.record()
    torch.cuda.synchronize()
    print(f"{kind.name}: blocks {len(M.block_bounds()) - 1}, levels {M.levels()}, apply {e0.elapsed_time(e1) / args.reps * 1e3:.1f} us", flush=True)
