#!/usr/bin/env python3
"""r06 lab: what a ROW costs the PATTERN tile kernel beside what an entry costs: banded matrices of (about) equal entry counts and
different row counts (k offsets per side), fp32, lanes per row 1 / 2 / 4, SpMV launch time from events around 50 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host

smm.init(0)
dev = torch.device("cuda:0")
s0 = torch.cuda.current_stream().cuda_stream
for rows, k in ((1250000, 25), (2500000, 12), (5000000, 6), (1250000, 12), (1250000, 6), (10000000, 25), (10000000, 12)):
    nnz = host.gen_banded_nnz(rows, k, 0x5EED, 1 << 16)
    ds = torch.empty(rows + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=torch.float32, device=dev)
    host.gen_banded_dev(rows, k, 0x5EED, 1 << 16, ds, dp, dv, np.float32, s0, diag_shift=1.0)
    torch.cuda.synchronize()
    x = torch.rand(rows, dtype=torch.float32, device=dev); y = torch.empty_like(x)
    line = f"rows {rows:>9} k {k:>2} entries {nnz:>10} ({nnz / rows:.1f} per row):"
    for lanes in (1, 2, 4):
        A = smm.CSRMatrix.from_device(rows, rows, ds, dp, dv, np.float32)
        A.set_kernel(3, lanes)
        for _ in range(5):
            A.spmv_dev(0, None, x, y, s0)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                A.spmv_dev(0, None, x, y, s0)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 20.0)
        line += f"  L={lanes}: {best:7.1f} us ({A.kernel_desc()[0].replace('spmvPattern', '')})"
        del A
    print(line, flush=True)
    del ds, dp, dv, x, y
