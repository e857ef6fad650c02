#!/usr/bin/env python3
"""BASELINE config 5's stand-in (convection-diffusion 108^3, fp64, BiCGStab to 1e-8) with every preconditioner, on one MI355X:
create time, iterations, solve time, and the time of one apply (HIP events over `reps` back-to-back applies).
    python tools/block_precond_timing.py [--n 108] [--block-rows 0,512,1024] [--skip-global]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=108)
    ap.add_argument("--block-rows", default="0")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--skip-global", action="store_true")
    ap.add_argument("--level-caps", default="-1", help="level cuts of the block kinds to time: -1 = the default, 0 = none, 2 .. 4095")
    ap.add_argument("--poisson2d", type=int, default=0, help="also: 2-D Poisson N x N")
    ap.add_argument("--values-read", action="store_true", help="the constant-diagonal SpMV encoding off: values[] is read (what a stencil with varying coefficients gets)")
    args = ap.parse_args()
    smm.init(0)
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()  # not the null stream: a hipGraph capture (SMM_HIP_SOLVER_GRAPH) needs a created stream
    torch.cuda.set_stream(side)
    stream = torch.cuda.current_stream().cuda_stream
    P = smm.SolverPreconditioner
    mats = []
    N = args.n
    n, nnz = N ** 3, host.gen_stencil3d_nnz(N, N, N)
    ds, dp, dv = (torch.empty(n + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.float64, device=dev))
    host.gen_stencil3d_dev(N, N, N, 6.0, -1.3, -0.7, ds, dp, dv, np.float64, stream)
    mats.append((f"convdiff3d {N}^3", smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float64), n, nnz, (ds, dp, dv)))
    if args.poisson2d:
        N = args.poisson2d
        n, nnz = N * N, host.gen_poisson2d_nnz(N, N)
        ds, dp, dv = (torch.empty(n + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.float64, device=dev))
        host.gen_poisson2d_dev(N, N, ds, dp, dv, np.float64, stream)
        mats.append((f"poisson2d {N}^2", smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float64), n, nnz, (ds, dp, dv)))
    if args.values_read:
        for m in mats:
            m[1].pattern_allow_const(False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, A, n, nnz, _keep in mats:
        ones = torch.ones(n, dtype=torch.float64, device=dev)
        b = torch.empty_like(ones)
        A.spmv_dev(0, None, ones, b, stream)
        kinds = [("none", None, None)]
        if not args.skip_global:
            kinds += [("jacobi", P.JACOBI, None), ("ilu0", P.ILU0, None), ("sgs", P.SYMMETRIC_GAUS_SEIDEL, None)]
        kinds = [k + (None,) for k in kinds]
        for br in [int(v) for v in args.block_rows.split(",")]:
            for cap in [int(v) for v in args.level_caps.split(",")]:
                tag = f"{br or 'default'}, cap {'default' if cap < 0 else cap}"
                kinds += [(f"block_ilu0[{tag}]", P.BLOCK_ILU0, br, cap), (f"block_sgs[{tag}]", P.BLOCK_SGS, br, cap)]
        for label, kind, br, cap in kinds:
            for attempt in range(2):  # the second create shows the steady state (allocations cached)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                M = A.getPreconditioner(kind, br if br else None, cap) if kind is not None else None
                torch.cuda.synchronize()
                tc = time.perf_counter() - t0
                if attempt == 0 and M is not None:
                    tc_first = tc
                    M.close()
            x = torch.zeros_like(ones)
            host.bicgstab_dev(A, b, x, 3, 1e-30, M, stream)  # warm-up
            x.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st, it, res = host.bicgstab_dev(A, b, x, -1, 1e-8, M, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            apply_ms = 0.0
            if M is not None:
                y = torch.empty_like(ones)
                M.apply_dev(b, y, stream)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.reps):
                    M.apply_dev(b, y, stream)
                e1.record()
                torch.cuda.synchronize()
                apply_ms = e0.elapsed_time(e1) / args.reps
            extra = ""
            if kind in (P.BLOCK_ILU0, P.BLOCK_SGS):
                bounds = M.block_bounds()
                extra = f" blocks {len(bounds) - 1} (rows <= {int(np.diff(bounds).max())})"
            lv = M.levels() if M is not None else (0, 0)
            bytes_ = 2 * (nnz * 12 + (n + 1) * 4) + 5 * n * 8
            print(f"{name} f64 BiCGStab+{label}: create {tc * 1e3:.2f} ms (first {tc_first * 1e3 if M is not None else 0:.2f}), levels {lv},{extra} {it} iterations in {dt * 1e3:.2f} ms "
                  f"({dt / max(it, 1) * 1e3:.4f} ms/it), create+solve {(tc + dt) * 1e3:.2f} ms, apply {apply_ms * 1e3:.1f} us"
                  f" ({bytes_ / max(apply_ms, 1e-9) / 1e6:.0f} GB/s of 2 x SpMV bytes), max|x-1| {float((x - 1).abs().max()):.1e}", flush=True)


if __name__ == "__main__":
    main()
