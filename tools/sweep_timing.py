#!/usr/bin/env python3
"""Times one preconditioner apply (two triangular sweeps) per launch mode on one MI355X:
    python tools/sweep_timing.py [--n3 108] [--n2 1000]
Rows printed: matrix, kind, levels, ms per apply with level-scheduled launches and with the synchronisation-free sweeps."""
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
    ap.add_argument("--n3", type=int, default=108)
    ap.add_argument("--n2", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    smm.init(0)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    P = smm.SolverPreconditioner
    mats = []
    N = args.n3
    n, nnz = N ** 3, host.gen_stencil3d_nnz(N, N, N)
    ds, dp, dv = (torch.empty(n + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.int32, device=dev),
                  torch.empty(nnz, dtype=torch.float64, device=dev))
    host.gen_stencil3d_dev(N, N, N, 6.0, -1.3, -0.7, ds, dp, dv, np.float64, stream)
    mats.append((f"convdiff3d {N}^3", smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float64), n, nnz, (ds, dp, dv)))
    N = args.n2
    n, nnz = N * N, host.gen_poisson2d_nnz(N, N)
    ds, dp, dv = (torch.empty(n + 1, dtype=torch.int32, device=dev), torch.empty(nnz, dtype=torch.int32, device=dev),
                  torch.empty(nnz, dtype=torch.float64, device=dev))
    host.gen_poisson2d_dev(N, N, ds, dp, dv, np.float64, stream)
    mats.append((f"poisson2d {N}^2", smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float64), n, nnz, (ds, dp, dv)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, A, n, nnz, _keep in mats:
        rhs = torch.rand(n, dtype=torch.float64, device=dev)
        for kind in (P.SYMMETRIC_GAUS_SEIDEL, P.ILU0):
            M = A.getPreconditioner(kind)
            res = {}
            outs = {}
            for mode, label in ((host.SWEEP_LEVELS, "levels"), (host.SWEEP_SYNCFREE, "syncfree"), (host.SWEEP_SYNCFREE_XCD, "onexcd")):
                M.set_sweep(mode)
                x = torch.zeros(n, dtype=torch.float64, device=dev)
                M.apply_dev(rhs, x, stream)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.reps):
                    M.apply_dev(rhs, x, stream)
                e1.record()
                torch.cuda.synchronize()
                res[label] = e0.elapsed_time(e1) / args.reps
                outs[label] = x
            same = bool(torch.equal(outs["levels"], outs["syncfree"])) and bool(torch.equal(outs["levels"], outs["onexcd"]))
            bytes_ = 2 * (nnz * 12 + (n + 1) * 4) + 5 * n * 8
            print(f"{name} {kind.name}: levels {M.levels()}  level-scheduled {res['levels']:.3f} ms  sync-free {res['syncfree']:.3f} ms  one-XCD {res['onexcd']:.3f} ms "
                  f"({bytes_ / res['syncfree'] / 1e6:.0f} GB/s)  identical {same}", flush=True)


if __name__ == "__main__":
    main()
