#!/usr/bin/env python3
"""27- / 19-point stencils: the three-window march kernel against the constant-diagonal gather kernel and the CSR stream (round 5).
   python tools/march3_timing.py [--grids 128,160,256] [--dtype f64]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import generators as gen, host

def time_spmv(A, x, y, stream, reps):
    for _ in range(3):
        A.spmv_dev(0, None, x, y, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        A.spmv_dev(0, None, x, y, stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grids", default="128,160,200,256")
    ap.add_argument("--points", default="27,19")
    ap.add_argument("--dtype", default="f64,f32")
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    smm.init(0)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    march_off = os.environ.get("SMM_HIP_CONST_MARCH") == "0"
    for dt in args.dtype.split(","):
        npd = np.float64 if dt == "f64" else np.float32
        td = torch.float64 if dt == "f64" else torch.float32
        for pts in (int(p) for p in args.points.split(",")):
            for g in (int(v) for v in args.grids.split(",")):
                csr = gen.stencil3d_wide(g, g, g, pts, dtype=npd)
                n = len(csr[0]) - 1
                d = [torch.from_numpy(a).to(dev) for a in csr]
                A = smm.CSRMatrix.from_device(n, n, d[0], d[1], d[2], npd)
                x = torch.rand(n, dtype=td, device=dev) - 0.5
                y = torch.empty_like(x)
                A.set_kernel(3, 1)
                name, nbytes = A.kernel_desc()
                us = time_spmv(A, x, y, stream, args.reps)
                ref = y.clone()
                A.set_kernel(2, 1)
                us_csr = time_spmv(A, x, y, stream, args.reps)
                same = bool(torch.equal(ref, y))
                print(f"{pts}-point {g}^3 {dt}: {name:32s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s on its {nbytes / 1e6:.0f} MB | CSR stream {us_csr:8.1f} us | bit-equal {same}"
                      + ("  (SMM_HIP_CONST_MARCH=0)" if march_off else ""), flush=True)
                A.close()
                del A, d, x, y, ref
                torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
