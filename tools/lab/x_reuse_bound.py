#!/usr/bin/env python3
"""r06 lab: what ANY scheme that removes the re-fetched x lines could gain on the benchmark matrix -- the same matrix shape (10 M rows, 25
offsets per side, fp32, PATTERN tile kernel at 2 lanes) with its offsets drawn below 2^20 (the benchmark: x lines are re-fetched ~31 times)
and below 2^16 (every line of x a row front touches stays in the XCD's L2: no re-fetch at all); and what a sweep over SHORT rows costs per
entry (the piece-major form of DESIGN section 7.5 would run four sweeps of 13 entries per row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host

smm.init(0)
dev = torch.device("cuda:0")
s0 = torch.cuda.current_stream().cuda_stream
for rows, k, maxoff, lanes in ((10_000_000, 25, 1 << 20, 2), (10_000_000, 25, 1 << 16, 2), (10_000_000, 6, 1 << 20, 1), (10_000_000, 6, 1 << 16, 1), (10_000_000, 6, 1 << 16, 2)):
    nnz = host.gen_banded_nnz(rows, k, 0x5EED, maxoff)
    ds = torch.empty(rows + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=torch.float32, device=dev)
    host.gen_banded_dev(rows, k, 0x5EED, maxoff, ds, dp, dv, np.float32, s0, diag_shift=1.0)
    torch.cuda.synchronize()
    x = torch.rand(rows, dtype=torch.float32, device=dev); y = torch.empty_like(x)
    A = smm.CSRMatrix.from_device(rows, rows, ds, dp, dv, np.float32)
    A.set_kernel(3, lanes)
    for _ in range(5):
        A.spmv_dev(0, None, x, y, s0)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            A.spmv_dev(0, None, x, y, s0)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50.0)
    name, nbytes = A.kernel_desc()
    print(f"rows {rows} k {k} offsets < 2^{maxoff.bit_length() - 1} lanes {lanes}: {nnz} entries ({nnz / rows:.1f} per row), {name}: {best:.1f} us per launch = {best * 1e6 / nnz:.3f} ns per 1000 entries, "
          f"{nbytes / best / 1e6:.2f} TB/s on its own {nbytes / 1e9:.3f} GB", flush=True)
    del A, ds, dp, dv, x, y
