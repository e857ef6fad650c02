#!/usr/bin/env python3
"""One rank's ConjugateGradient iteration of BASELINE config 4 at 8 GPUs: a 512 x 512 x 64 slab of the 7-point Laplacian (16.8 M rows, fp64
vectors of 134 MB) through the row-partitioned loop (csrc/smm_dist.hip distCg) on a single-rank communicator -- the direction formed inside the
SpMV (r06), x deferred (distCgLazyP), the eager loop -- and the single-GPU loop (cgDev) on the same matrix.  (A rank WITH a remote block:
tools/lab/slab_cg_remote.py.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host
from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

smm.init(0)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream().cuda_stream
nx, ny, nz = 512, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 64
for dtype in (np.float64, np.float32):
    td = torch.float32 if dtype == np.float32 else torch.float64
    n = nx * ny * nz; nnz = host.gen_stencil3d_nnz(nx, ny, nz)
    ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=td, device=dev)
    host.gen_stencil3d_dev(nx, ny, nz, 6.0, -1.0, -1.0, ds, dp, dv, dtype, stream)
    torch.cuda.synchronize()
    comm = NativeComm.single()
    A = NativeDistMatrix(comm, n, [0, n], ds, dp, dv, dtype)
    ones = torch.ones(n, dtype=td, device=dev); b = torch.empty_like(ones); A.spmv(0, None, ones, b)
    out = {}
    for name, knob, fuse in (("p formed inside the SpMV, x deferred", -1, True), ("deferred x", -1, False), ("eager", 1 << 60, False)):
        host.set_cg_lazy_x_min_bytes(knob)
        host.set_cg_fuse_p(fuse)
        x = torch.zeros_like(ones); A.cg(b, x, x, 10, 0.0)
        x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        st, it, res = A.cg(b, x, x, 100, 0.0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[name] = x.clone()
        print(f"slab {nx}x{ny}x{nz} {np.dtype(dtype).name} row-partitioned CG, {name}: {it} iterations in {dt*1e3:.1f} ms = {dt/it*1e6:.0f} us per iteration (A_loc on {A.local_blocks()[0].kernel_desc()[0]})", flush=True)
    host.set_cg_lazy_x_min_bytes(-1)
    host.set_cg_fuse_p(True)
    print("   deferred == eager bit for bit:", bool(torch.equal(out["deferred x"], out["eager"])), "; fused against eager, max |dx| / max |x|:",
          float((out["p formed inside the SpMV, x deferred"] - out["eager"]).abs().max() / out["eager"].abs().max()),
          "(at this size the fused launch's grid differs from the plain one's: other partial sums; on small grids the same bits, tests/dist_cg_fuse_check.py)")
    B = smm.CSRMatrix.from_device(n, n, ds, dp, dv, dtype)
    x = torch.zeros_like(ones); host.cg_dev(B, b, x, x, 10, 0.0, None, stream)
    x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    st, it, res2 = host.cg_dev(B, b, x, x, 100, 0.0, None, stream); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"   single-GPU loop on the same matrix: {dt/it*1e6:.0f} us per iteration; x equal to the row-partitioned one: {bool(torch.equal(x, out['eager']))}")
    A.close(); comm.close(); del B, A
