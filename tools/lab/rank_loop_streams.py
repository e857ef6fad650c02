#!/usr/bin/env python3
"""One rank's share of the benchmark matrix (1.25 M rows): microseconds per BiCGStab iteration of the row-partitioned loop (single-rank
communicator) and of the single-GPU loop, on the NULL stream and on a stream of the caller's own."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host
from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

smm.init(0)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
maxoff = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20  # (2^20: the benchmark matrix's band -- at 1.25 M rows most rows then lose diagonals at the edges; 2^16: full rows)
nnz = host.gen_banded_nnz(n, 25, 0x5EED, maxoff)
s0 = torch.cuda.current_stream().cuda_stream
ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=torch.float32, device=dev)
host.gen_banded_dev(n, 25, 0x5EED, maxoff, ds, dp, dv, np.float32, s0, diag_shift=1.0)
print(f"{n} rows, offsets below {maxoff}: {nnz} entries ({nnz / n:.1f} per row)")
torch.cuda.synchronize()
xt = torch.rand(n, dtype=torch.float32, device=dev) + 0.5
b = torch.empty_like(xt)
comm = NativeComm.single()
D = NativeDistMatrix(comm, n, [0, n], ds, dp, dv, np.float32)
A = smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float32)
A.spmv_dev(0, None, xt, b, s0)
torch.cuda.synchronize()
own = torch.cuda.Stream(device=dev)
for name, st in (("NULL stream", s0), ("own stream", own.cuda_stream)):
    for kind in ("row-partitioned", "single-GPU"):
        def solve(it):
            x = torch.zeros_like(xt); torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = D.bicgstab(b, x, it, 0.0, st) if kind == "row-partitioned" else host.bicgstab_dev(A, b, x, it, 0.0, None, st)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, res
        solve(20)
        best = min(solve(20)[0] for _ in range(8))
        print(f"{name}, {kind}: {best / 20 * 1e6:.1f} us per iteration (solves of 20 iterations, best of 8)", flush=True)
