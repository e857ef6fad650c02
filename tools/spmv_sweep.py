#!/usr/bin/env python3
"""Times every SpMV kernel configuration on a generated matrix (GPU box):  python tools/spmv_sweep.py --matrix banded --rows 10000000"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="banded", choices=["banded", "poisson2d", "poisson3d"])
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--max-offset", type=int, default=1 << 20, help="banded: the offsets are drawn from [1, max-offset)")
    ap.add_argument("--ny", type=int, default=0, help="poisson2d: rows of the grid (default: n)")
    ap.add_argument("--configs", default="2:1,2:2,2:4,2:8,2:16,1:4,1:8,1:16,1:32,1:64")
    ap.add_argument("--pos-mode", default="orig", choices=["orig", "row", "near", "random"],
                    help="ablation: overwrite positions[] so every gather hits x[row] (row) or x[row + j - len/2] (near)")
    args = ap.parse_args()
    smm.init(0)
    dev = torch.device("cuda:0")
    npd = np.float32 if args.dtype == "f32" else np.float64
    td = torch.float32 if args.dtype == "f32" else torch.float64
    s = 4 if args.dtype == "f32" else 8
    stream = torch.cuda.current_stream().cuda_stream
    if args.matrix == "banded":
        n = args.rows
        nnz = host.gen_banded_nnz(n, args.k, 0x5EED, args.max_offset)
    elif args.matrix == "poisson2d":
        ny = args.ny or args.n
        n = args.n * ny
        nnz = host.gen_poisson2d_nnz(args.n, ny)
    else:
        n = args.n ** 3
        nnz = host.gen_stencil3d_nnz(args.n, args.n, args.n)
    d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
    d_val = torch.empty(nnz, dtype=td, device=dev)
    if args.matrix == "banded":
        host.gen_banded_dev(n, args.k, 0x5EED, args.max_offset, d_start, d_pos, d_val, npd, stream)
    elif args.matrix == "poisson2d":
        host.gen_poisson2d_dev(args.n, args.ny or args.n, d_start, d_pos, d_val, npd, stream)
    else:
        host.gen_stencil3d_dev(args.n, args.n, args.n, 6.0, -1.0, -1.0, d_start, d_pos, d_val, npd, stream)
    if args.pos_mode != "orig":
        torch.cuda.synchronize()
        lens = (d_start[1:] - d_start[:-1]).to(torch.int64)
        rows_of = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), lens)
        if args.pos_mode == "near":
            j = torch.arange(nnz, device=dev, dtype=torch.int64) - torch.repeat_interleave(d_start[:-1].to(torch.int64), lens)
            rows_of = (rows_of.to(torch.int64) + j - 24).clamp_(0, n - 1).to(torch.int32)
            del j
        if args.pos_mode == "random":
            # worst case for the gather: stratified i.i.d. columns (entry j of a row of length L falls uniformly into the j-th of L equal
            # slices of [0, n)), ascending by construction; SURVEY.md section 8d's "secondary" matrix, SpMV only
            j = torch.arange(nnz, device=dev, dtype=torch.int64) - torch.repeat_interleave(d_start[:-1].to(torch.int64), lens)
            width = (n // torch.repeat_interleave(lens, lens)).clamp_(min=1)
            g = torch.Generator(device=dev).manual_seed(7)
            r = (torch.rand(nnz, device=dev, generator=g, dtype=torch.float64) * width.to(torch.float64)).to(torch.int64)
            rows_of = (j * width + torch.minimum(r, width - 1)).clamp_(0, n - 1).to(torch.int32)
            del j, width, r
        d_pos.copy_(rows_of)
        del rows_of, lens
    A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, npd)
    x = torch.rand(n, dtype=td, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    y = torch.empty(n, dtype=td, device=dev)
    bytes_ = nnz * (s + 4) + (n + 1) * 4 + 2 * n * s
    print(f"matrix {args.matrix} rows {n} nnz {nnz} ({nnz / n:.1f}/row) {args.dtype}: B_spmv = {bytes_ / 1e9:.3f} GB; default kernel {A.get_kernel()}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for cfg in args.configs.split(","):
        fam, lanes = (int(v) for v in cfg.split(":"))
        A.set_kernel(fam, lanes)
        for _ in range(3):
            A.spmv_dev(0, None, x, y, stream)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            A.spmv_dev(0, None, x, y, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        # a checksum of y: the same bits whatever order the tiles are walked in (A/B runs of tile orders compare these)
        chk = int(torch.sum(y.view(torch.int32 if args.dtype == "f32" else torch.int64).to(torch.int64) & 0xFFFFFF).item())
        # priced with the bytes the kernel that ran moves by ITS OWN layout (smm_hip_csr_kernel_desc), never with another layout's
        kname, own = A.kernel_desc()
        print(f"family {fam} lanes {lanes:2d}: {ms:8.4f} ms  {kname} moves {own / 1e9:.3f} GB: {own / ms / 1e6:8.1f} GB/s  {100 * own / ms / 1e6 / 8000:5.1f} % of 8 TB/s  y-checksum {chk}", flush=True)


if __name__ == "__main__":
    main()
