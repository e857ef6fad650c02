#!/usr/bin/env python3
"""r06 lab: the row-partitioned SpMV of one rank's share (1.25 M rows, stand-in B) by form, timed with events around 100 back-to-back
distributed SpMVs on a single-rank communicator (SMM_HIP_LAB_SELF_SPLIT: entries farther than `window` from the row count as remote).
usage: split_spmv_timing.py [rows] [max_offset] [window]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host
from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

smm.init(0)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
maxoff = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
window = int(sys.argv[3]) if len(sys.argv) > 3 else 37500
nnz = host.gen_banded_nnz(n, 25, 0x5EED, maxoff)
s0 = torch.cuda.current_stream().cuda_stream
ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=torch.float32, device=dev)
host.gen_banded_dev(n, 25, 0x5EED, maxoff, ds, dp, dv, np.float32, s0, diag_shift=1.0)
torch.cuda.synchronize()
xt = torch.rand(n, dtype=torch.float32, device=dev) + 0.5
y = torch.empty_like(xt)
comm = NativeComm.single()
lanes = os.environ.get("LAB_LANES", "2,1").split(",")


def matrix(win, split):
    os.environ["SMM_HIP_LAB_SELF_SPLIT"] = str(win)
    os.environ["SMM_HIP_SPLIT_SPMV"] = "1" if split else "0"
    D = NativeDistMatrix(comm, n, [0, n], ds, dp, dv, np.float32)
    for blk, L in zip(D.local_blocks(), lanes if win > 0 else ("2", "1")):
        if blk.nnz > 0:
            blk.set_kernel(3, int(L))
    os.environ.pop("SMM_HIP_LAB_SELF_SPLIT"); os.environ.pop("SMM_HIP_SPLIT_SPMV")
    return D


own = torch.cuda.Stream(device=dev)
st = own.cuda_stream
for name, D in (("unsplit (one block, one launch)", matrix(0, True)), ("A_loc / A_rem in TWO launches", matrix(window, False)), ("A_loc / A_rem in ONE launch", matrix(window, True))):
    for _ in range(5):
        D.spmv(0, None, xt, y, st)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(own):
            e0.record()
            for _ in range(100):
                D.spmv(0, None, xt, y, st)
            e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10.0)
    print(f"{name}: {best:.1f} us per SpMV (copy of x into the halo-extended vector included); forms {D.matvec_forms()}; nnz {D.nnz_loc} / {D.nnz_rem}", flush=True)

# the two blocks on their own, by lanes per row (what each half costs as a launch of its own)
from sparse_matrix_math_amd import _lib
lib = _lib.load()
D = matrix(window, False)
xe = torch.rand(D.ext_len, dtype=torch.float32, device=dev)
for which, blk in zip(("A_loc", "A_rem"), D.local_blocks()):
    line = f"{which} alone ({blk.nnz} entries, {blk.nnz / n:.1f} per row):"
    for L in (1, 2, 4):
        blk.set_kernel(3, L)
        def run():
            _lib.check(lib.smm_hip_spmv_dev_f32(blk._h, 0, None, host._dptr(xe), host._dptr(y), host._dptr(st)))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(own):
                e0.record()
                for _ in range(50):
                    run()
                e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 20.0)
        line += f"  L={L}: {best:.1f} us ({blk.kernel_desc()[0].replace('spmvPattern', '')})"
    print(line, flush=True)
