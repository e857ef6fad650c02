#!/usr/bin/env python3
"""GPU box: ConjugateGradient with its next direction formed inside the 2.5-D SpMV kernel (MarchFuse) against the launch that forms it in
cgLazyXP and against the eager loop -- bit for bit -- on small grids (SMM_HIP_NT_OUT=1, march and deferred-x thresholds lowered for this
process): 3-D 7-point and 2-D 5-point stencils incl. partial tiles / planes, every iteration count 0..10, convergence inside the loop, x0
in place and apart, fp32 / fp64; then against the oracle."""
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

smm.init(0)
oracle = Oracle()
host.cg_resident(host.CG_RESIDENT_OFF)
bad = 0
for dtype in (np.float64, np.float32):
    for name, csr in (("stencil 96x112x33", gen.stencil3d(96, 112, 33, dtype=dtype)), ("stencil 128x64x9", gen.stencil3d(128, 64, 9, dtype=dtype)),
                      ("poisson2d 600x37", gen.poisson2d(600, 37, dtype=dtype))):
        start, pos, val = csr
        n = len(start) - 1
        A = smm.CSRMatrix(n, n, *csr)
        A.set_kernel(3, 1)
        kernel = A.kernel_desc()[0]
        b = gen.row_sums(start, val).astype(dtype)
        rng = np.random.default_rng(5)
        x0s = rng.uniform(-1, 1, n).astype(dtype)
        for maxit, eps in [(k, 0.0) for k in range(0, 20)] + [(-1, 1e-2 if dtype == np.float32 else 1e-6), (400, 3.0)]:
            for in_place in (True, False):
                got = {}
                for mode in ("fused", "lazy", "eager"):
                    host.set_cg_lazy_x_min_bytes(1 << 60 if mode == "eager" else 0)
                    host.set_cg_fuse_p(mode == "fused")
                    x = x0s.copy() if in_place else np.full(n, 7, dtype=dtype)
                    info = {}
                    st = smm.ConjugateGradient(A, b, x if in_place else x0s, x, maxit, dtype(eps), info=info)
                    got[mode] = (int(st), info["iterations"], x.copy())
                ok = got["fused"][:2] == got["lazy"][:2] == got["eager"][:2] and np.array_equal(got["fused"][2], got["eager"][2]) and np.array_equal(got["lazy"][2], got["eager"][2])
                if not ok:
                    bad += 1
                    d = float(np.max(np.abs(got["fused"][2].astype(np.float64) - got["eager"][2])))
                    print(f"MISMATCH {name} {np.dtype(dtype).name} maxit {maxit} eps {eps} in_place {in_place}: fused {got['fused'][:2]} lazy {got['lazy'][:2]} eager {got['eager'][:2]} max|dx| {d:.3e}")
        st_o, x_o, it_o, _ = oracle.cg(csr, b, np.zeros(n, dtype=dtype), 7, 0.0)
        host.set_cg_lazy_x_min_bytes(0)
        host.set_cg_fuse_p(True)
        x = np.zeros(n, dtype=dtype)
        st = smm.ConjugateGradient(A, b, x, x, 7, dtype(0.0))
        err = float(np.max(np.abs(x - x_o)))
        tol = 5e-3 if dtype == np.float32 else 1e-10  # (fp32: the oracle adds 10^5 products one after the other; the eager loop sits as far from it)
        if int(st) != st_o or err > tol * max(1.0, float(np.max(np.abs(x_o)))):
            bad += 1
            print(f"ORACLE MISMATCH {name} {np.dtype(dtype).name}: {err:.3e}")
        print(f"{name:20s} {np.dtype(dtype).name} kernel {kernel}: fused == deferred == eager for 22 x 2 solves; vs oracle {err:.2e}", flush=True)
        A.close()
print("cg fuse check:", "ALL OK" if bad == 0 else f"{bad} BAD")
sys.exit(1 if bad else 0)
