#!/usr/bin/env python3
"""One rank's share of the benchmark matrix (1.25 M rows) through the row-partitioned BiCGStab on a single-rank communicator and through
the single-GPU loop, WITHOUT bench.py's live-timing events: for a rocprofv3 --kernel-trace of the time between kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host
from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

smm.init(0)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
which = sys.argv[2] if len(sys.argv) > 2 else "dist"
nnz = host.gen_banded_nnz(n, 25, 0x5EED, 1 << 20)
ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=torch.float32, device=dev)
host.gen_banded_dev(n, 25, 0x5EED, 1 << 20, ds, dp, dv, np.float32, stream, diag_shift=1.0)
torch.cuda.synchronize()
xt = torch.rand(n, dtype=torch.float32, device=dev) + 0.5
b = torch.empty_like(xt)
if which == "dist":
    comm = NativeComm.single()
    A = NativeDistMatrix(comm, n, [0, n], ds, dp, dv, np.float32)
    A.spmv(0, None, xt, b)
    for rep in range(6):
        x = torch.zeros_like(xt); torch.cuda.synchronize()
        res = A.bicgstab(b, x, 20, 0.0)
    torch.cuda.synchronize(); print("dist", res)
else:
    A = smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float32)
    A.spmv_dev(0, None, xt, b, stream)
    for rep in range(6):
        x = torch.zeros_like(xt); torch.cuda.synchronize()
        res = host.bicgstab_dev(A, b, x, 20, 0.0, None, stream)
    torch.cuda.synchronize(); print("single", res)
