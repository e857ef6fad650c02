#!/usr/bin/env python3
"""One rank's ConjugateGradient iteration of BASELINE config 4 at 8 GPUs WITH a remote block: the 512 x 512 x 64 slab of tools/dist_cg_timing.py on
a single-rank communicator, the last plane's columns counted as remote (SMM_HIP_LAB_SELF_SPLIT=-P: A_rem then holds the entries a neighbouring
slab would own -- about two planes of rows out of 64 -- nothing travels).  Against the same loop with nothing remote.

usage: slab_cg_remote.py [nz] [dtype: f64|f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host
from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

smm.init(0)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
nx, ny, nz = 512, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
td = torch.float32 if dtype == np.float32 else torch.float64
n = nx * ny * nz; nnz = host.gen_stencil3d_nnz(nx, ny, nz)
ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=td, device=dev)
host.gen_stencil3d_dev(nx, ny, nz, 6.0, -1.0, -1.0, ds, dp, dv, dtype, stream)
torch.cuda.synchronize()
comm = NativeComm.single()
ones = torch.ones(n, dtype=td, device=dev); b = None
res = {}
legs = [int(v) for v in os.environ["LAB_LEGS"].split(",")] if os.environ.get("LAB_LEGS") else [0, 1, 2, 3, 4]  # (LAB_LEGS=2: one leg, for a profile)
P = nx * ny
for leg, (name, win, thin, fuse) in enumerate((("nothing remote, p formed in a launch of its own", 0, 1, 0),
                                               ("nothing remote, p formed inside the SpMV", 0, 1, 1),
                                               ("the last plane's columns remote, second launch over all rows, p in a launch of its own", -P, 0, 0),
                                               ("the last plane's columns remote, second launch over the rows with a remote entry, p in a launch of its own", -P, 1, 0),
                                               ("the last plane's columns remote, second launch over the rows with a remote entry, p formed inside the SpMV", -P, 1, 1))):
    if leg not in legs:
        continue
    os.environ["SMM_HIP_LAB_SELF_SPLIT"] = str(win)
    os.environ["SMM_HIP_THIN_REMOTE"] = str(thin)
    host.set_cg_fuse_p(bool(fuse))
    A = NativeDistMatrix(comm, n, [0, n], ds, dp, dv, dtype)
    os.environ.pop("SMM_HIP_LAB_SELF_SPLIT"); os.environ.pop("SMM_HIP_THIN_REMOTE")
    if b is None:
        b = torch.empty_like(ones); A.spmv(0, None, ones, b)
    bicg = os.environ.get("LAB_SOLVER") == "bicgstab"  # (LAB_SOLVER=bicgstab: the same slab through the row-partitioned BiCGStab loop, two SpMVs per iteration)
    solve = (lambda its: A.bicgstab(b, x, its, 0.0)) if bicg else (lambda its: A.cg(b, x, x, its, 0.0))
    x = torch.zeros_like(ones); solve(10)
    best = 1e9
    for _ in range(3):
        x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        st, it, r2 = solve(100); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    res[name] = x.clone()
    blocks = A.local_blocks()
    desc = " / ".join(blk.kernel_desc()[0] for blk in blocks)
    print(f"slab {nx}x{ny}x{nz} {np.dtype(dtype).name}{' BiCGStab' if bicg else ''}, {name}: {best / it * 1e6:.0f} us per iteration (100 iterations, best of 3); nnz A_loc {A.nnz_loc}, A_rem {A.nnz_rem}; kernels {desc}; SpMV forms (one launch, two) {A.matvec_forms()}, thin (rows, SpMVs) {A.thin_remote()}, SpMVs that formed p {A.cg_fused()}", flush=True)
    A.close()
keys = list(res)
for k in keys[1:]:
    print(f"max |x - x(nothing remote)| / max |x|, {k}: {float((res[k] - res[keys[0]]).abs().max()) / float(res[keys[0]].abs().max()):.3e}")
