#!/usr/bin/env python3
"""Run by tests/test_gpu_dist_native.py in a process of its own (the non-temporal policy and the march threshold are read from the environment
once).  GPU box: the row-partitioned ConjugateGradient (csrc/smm_dist.hip distCg) with its next direction formed inside the local block's 2.5-D SpMV
kernel -- the halo of r travelling instead of p's, every rank forming the halo of the new direction itself, the thin remote block behind it --
against the loop that forms p in distCgLazyP and against the eager loop: bit for bit, on small grids cut into 1 / 2 / 3 slabs (the local block forced to the index-free family) whose ranks are
threads joined by the host-callback communicator (SMM_HIP_NT_OUT=1, march and deferred-x thresholds lowered for this process): every iteration
count 0..19, convergence inside the loop, a start vector away from zero, fp32 / fp64; then against the oracle."""
import os
import sys

os.environ["SMM_HIP_NT_OUT"] = "1"
os.environ["SMM_HIP_MARCH_MIN_ROWS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import sparse_matrix_math_amd as smm  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from sparse_matrix_math_amd import generators as gen, host  # noqa: E402
from tests.test_gpu_dist_native import _solve  # noqa: E402

smm.init(0)
oracle = Oracle()
bad = 0
for dtype in (np.float64, np.float32):
    # (worlds: a slab's remote block is thin -- and the ranks vote for the fused form -- from 16 planes per rank; three ranks: whole planes each,
    # because the 2.5-D kernel wants a multiple of 16 bytes of rows and the partition by stored entries does not always give one)
    for name, csr, worlds in (("stencil 96x112x33", gen.stencil3d(96, 112, 33, dtype=dtype), (1, 2)), ("stencil 96x88x54", gen.stencil3d(96, 88, 54, dtype=dtype), (1, 2, 3))):
        start, pos, val = csr
        n = len(start) - 1
        b = gen.row_sums(start, val).astype(dtype)
        x0s = np.random.default_rng(5).uniform(-1, 1, n).astype(dtype)
        for world in worlds:
            fused_spmvs = 0
            cases = [(k, 0.0) for k in (0, 1, 2, 3, 7, 8, 9, 10, 16, 17, 19)] + [(-1, 1e-2 if dtype == np.float32 else 1e-6), (400, 3.0)]
            for maxit, eps in cases:
                got = {}
                for mode in ("fused", "lazy", "eager"):
                    host.set_cg_lazy_x_min_bytes(1 << 60 if mode == "eager" else 0)
                    host.set_cg_fuse_p(mode == "fused")
                    seen = {}
                    (st, it, res), x, _, _ = _solve(smm, csr, b, world, dtype, maxit, dtype(eps), solver="cg", x0_full=x0s, forms_seen=seen, lanes=(1,), bounds=[k * 18 * 96 * 88 for k in range(4)] if world == 3 else None)
                    got[mode] = (int(st), int(it), x.copy(), seen)
                ok = got["fused"][:2] == got["lazy"][:2] == got["eager"][:2] and np.array_equal(got["fused"][2], got["eager"][2]) and np.array_equal(got["lazy"][2], got["eager"][2])
                counts = [got["fused"][3][r][2] for r in range(world)]
                # every rank's SpMVs but the first formed the direction (a solve that stops early leaves launches behind it that return at once: counted all the same)
                if got["fused"][1] >= 2 and min(counts) < got["fused"][1] - 1:
                    ok = False
                if any(got[m][3][r][2] for m in ("lazy", "eager") for r in range(world)):
                    ok = False
                if world > 1 and any(got["fused"][3][r][0] == 0 for r in range(world)):
                    ok = False  # (a slab's remote block is thin)
                fused_spmvs += sum(counts)
                if not ok:
                    bad += 1
                    dmax = float(np.max(np.abs(got["fused"][2].astype(np.float64) - got["eager"][2])))
                    print(f"MISMATCH {name} {np.dtype(dtype).name} world {world} maxit {maxit} eps {eps}: fused {got['fused'][:2]} {got['fused'][3]} lazy {got['lazy'][:2]} eager {got['eager'][:2]} max|dx| {dmax:.3e}")
            st_o, x_o, it_o, _ = oracle.cg(csr, b, np.zeros(n, dtype=dtype), 7, 0.0)
            host.set_cg_lazy_x_min_bytes(0)
            host.set_cg_fuse_p(True)
            (st, it, res), x, _, _ = _solve(smm, csr, b, world, dtype, 7, dtype(0.0), solver="cg", lanes=(1,))
            err = float(np.max(np.abs(x - x_o)))
            tol = 5e-3 if dtype == np.float32 else 1e-10
            if int(st) != st_o or err > tol * max(1.0, float(np.max(np.abs(x_o)))):
                bad += 1
                print(f"ORACLE MISMATCH {name} {np.dtype(dtype).name} world {world}: {err:.3e}")
            print(f"{name:18s} {np.dtype(dtype).name} {world} rank(s): fused == deferred == eager for {len(cases)} solves ({fused_spmvs} SpMVs formed p); vs oracle {err:.2e}", flush=True)
host.set_cg_lazy_x_min_bytes(-1)
host.set_cg_fuse_p(True)
print("dist cg fuse check:", "ALL OK" if bad == 0 else f"{bad} BAD")
sys.exit(1 if bad else 0)
