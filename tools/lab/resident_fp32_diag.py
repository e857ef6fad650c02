"""round 5 diagnostic: the single-launch BiCGStab in fp32 on 128^3 (16 rows per lane) -- where does the residual turn NaN, and does the loop agree?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import generators as gen, host

smm.init(0)
for grid, dtype in ((128, np.float32), (64, np.float32), (108, np.float64)):
    csr = gen.convdiff3d(grid, 0.3, dtype=dtype)
    n = len(csr[0]) - 1
    A = smm.CSRMatrix(n, n, *csr)
    A.set_kernel(3, 1)
    b = gen.row_sums(csr[0], csr[2]).astype(dtype)
    for pname in ("none", "jacobi"):
        M = A.getPreconditioner(smm.SolverPreconditioner.JACOBI) if pname == "jacobi" else None
        for maxit in (5, 20, 60, 150, -1):
            out = []
            for mode in (2, 0):
                host.bicgstab_resident(mode)
                x = np.zeros(n, dtype=dtype)
                info = {}
                st = smm.BiCGStab(A, b.copy(), x, maxit, 2e-2 if dtype == np.float32 else 1e-8, M, info=info)
                out.append((int(st), info["iterations"], float(info["resnorm"]), float(np.max(np.abs(x - 1)))))
            print(grid, np.dtype(dtype).name, pname, maxit, "single launch", out[0], "loop", out[1], flush=True)
