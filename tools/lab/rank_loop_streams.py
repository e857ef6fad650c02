#!/usr/bin/env python3
"""One rank's share of the benchmark matrix (1.25 M rows): microseconds per BiCGStab iteration of the row-partitioned loop on a
single-rank communicator and of the single-GPU loop.

usage: rank_loop_streams.py [rows] [max_offset] [window]
  window > 0 (r06): SMM_HIP_LAB_SELF_SPLIT -- entries farther than `window` columns from their row count as "remote": the rank's rows are
  split into A_loc / A_rem as they are in a many-GPU run (58 % / 42 % of the entries at 8 GPUs), nothing travels, and the word the
  one-launch SpMV waits for is raised from the communicator's stream behind the update.  Three row-partitioned legs then: the SpMV in ONE
  launch (csrc/smm_spmv_split.hip), in TWO launches (SMM_HIP_SPLIT_SPMV=0), and unsplit (no window: one block, one launch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host
from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

smm.init(0)
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
maxoff = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20  # (2^20: the benchmark matrix's band -- at 1.25 M rows most rows then lose diagonals at the edges; 2^16: full rows)
window = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nnz = host.gen_banded_nnz(n, 25, 0x5EED, maxoff)
s0 = torch.cuda.current_stream().cuda_stream
ds = torch.empty(n + 1, dtype=torch.int32, device=dev); dp = torch.empty(nnz, dtype=torch.int32, device=dev); dv = torch.empty(nnz, dtype=torch.float32, device=dev)
host.gen_banded_dev(n, 25, 0x5EED, maxoff, ds, dp, dv, np.float32, s0, diag_shift=1.0)
print(f"{n} rows, offsets below {maxoff}: {nnz} entries ({nnz / n:.1f} per row)")
torch.cuda.synchronize()
xt = torch.rand(n, dtype=torch.float32, device=dev) + 0.5
b = torch.empty_like(xt)
comm = NativeComm.single()
A = smm.CSRMatrix.from_device(n, n, ds, dp, dv, np.float32)
A.spmv_dev(0, None, xt, b, s0)
torch.cuda.synchronize()
own = torch.cuda.Stream(device=dev)


def dist_matrix(win, split, slots=False):
    os.environ["SMM_HIP_LAB_SELF_P2P"] = "1" if slots else "0"  # the scalars through the (one-rank) slots: the peer-to-peer transport's launches
    os.environ["SMM_HIP_LAB_SELF_SPLIT"] = str(win)
    os.environ["SMM_HIP_SPLIT_SPMV"] = "1" if split else "0"
    D = NativeDistMatrix(comm, n, [0, n], ds, dp, dv, np.float32)
    if win != 0 and os.environ.get("LAB_LANES"):  # e.g. LAB_LANES=2,2: pieces per row of A_loc, A_rem (PATTERN family forced)
        for blk, L in zip(D.local_blocks(), os.environ["LAB_LANES"].split(",")):
            blk.set_kernel(3, int(L))
    os.environ.pop("SMM_HIP_LAB_SELF_SPLIT")
    os.environ.pop("SMM_HIP_SPLIT_SPMV")
    return D


legs = [("row-partitioned, unsplit", dist_matrix(0, True))]
if window != 0:
    legs.append(("row-partitioned, A_loc / A_rem in ONE launch", dist_matrix(window, True)))
    legs.append(("row-partitioned, A_loc / A_rem in TWO launches", dist_matrix(window, False)))
    legs.append(("row-partitioned, ONE launch, scalars through the slots (the peer-to-peer transport's launches)", dist_matrix(window, True, slots=True)))
legs.append(("single-GPU", None))
results = {}
for name, st in (("own stream", own.cuda_stream),) if window != 0 else (("NULL stream", s0), ("own stream", own.cuda_stream)):
    for kind, D in legs:
        def solve(it):
            x = torch.zeros_like(xt); torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = D.bicgstab(b, x, it, 0.0, st) if D is not None else host.bicgstab_dev(A, b, x, it, 0.0, None, st)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, res, x
        solve(20)
        best = min(solve(20)[0] for _ in range(8))
        _, res, x = solve(20)
        results[kind] = x.cpu().numpy().tobytes()
        extra = ""
        if D is not None:
            one, two = D.matvec_forms()
            extra = f"; options {D.options['p2p_scalars']}; nnz A_loc {D.nnz_loc} ({100.0 * D.nnz_loc / max(1, D.nnz_loc + D.nnz_rem):.0f} %), A_rem {D.nnz_rem}; SpMVs in one launch {one}, in two {two}"
        print(f"{name}, {kind}: {best / 20 * 1e6:.1f} us per iteration (solves of 20 iterations, best of 8){extra}", flush=True)
if window != 0:
    a, c = results["row-partitioned, A_loc / A_rem in ONE launch"], results["row-partitioned, A_loc / A_rem in TWO launches"]
    print("x after 20 iterations, one launch == two launches bit for bit:", a == c)
