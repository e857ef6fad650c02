"""One rank of a row-partitioned solve BETWEEN PROCESSES on one GPU (started world-size times by tests/test_gpu_dist_native.py): every
rank builds the same small matrix on the host, takes its row range, creates the gloo rehearsal communicator (host callbacks; the
set-up collectives) and the distributed matrix, and runs the stand-alone distributed SpMV, BiCGStab (with / without Jacobi) and CG.
Rank 0 gathers the pieces and prints one line `P2P_WORKER {json}` with the assembled results as hex strings; the TEST compares them
with the oracle (this script never imports it).

Matrices:
  banded     -- gen.banded_random_spd(60000, k=12, max_offset=9000): every rank exchanges with its neighbours
  decoupled  -- a block-diagonal matrix of two such bands whose SECOND block is exactly the last rank's rows: that rank neither sends
                nor receives, and in a world >= 3 with relays it is picked as a relay (planRelays goes by ring distance alone) -- the
                relay-only rank of ADVICE r05
  grid       -- gen.stencil3d(96, 88, 54) cut into slabs of whole planes: the local blocks are served by the 2.5-D constant-diagonal kernel
                (the test lowers its thresholds through the environment; x is deferred from 0 bytes), the remote blocks are thin, and
                ConjugateGradient forms its next direction inside the SpMV -- r's halo travels, each rank forms the halo of p itself
usage: p2p_proc_worker.py MATRIX DTYPE [pattern]  (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the environment)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_matrix(kind, dtype, world):
    from sparse_matrix_math_amd import generators as gen
    from sparse_matrix_math_amd.distributed import partition_rows_by_nnz

    if kind == "banded":
        start, pos, val = gen.banded_random_spd(60000, k=12, seed=4, max_offset=9000, dtype=dtype)
        n = len(start) - 1
        return (start, pos, val), partition_rows_by_nnz(lambda i: int(start[i]), n, world)
    if kind == "grid":
        planes = 54
        assert planes % world == 0
        csr = gen.stencil3d(96, 88, planes, dtype=dtype)
        return csr, [k * (planes // world) * 96 * 88 for k in range(world + 1)]
    assert kind == "decoupled" and world >= 2
    s1, p1, v1 = gen.banded_random_spd(45000, k=12, seed=4, max_offset=7000, dtype=dtype)
    s2, p2, v2 = gen.banded_random_spd(15000, k=12, seed=9, max_offset=3000, dtype=dtype)
    n1, n2 = len(s1) - 1, len(s2) - 1
    start = np.concatenate([s1, s2[1:] + s1[-1]]).astype(np.int32)
    pos = np.concatenate([p1, p2 + n1]).astype(np.int32)
    val = np.concatenate([v1, v2]).astype(dtype)
    bounds = partition_rows_by_nnz(lambda i: int(s1[i]), n1, world - 1) + [n1 + n2]
    return (start, pos, val), bounds


def main():
    kind, dtype = sys.argv[1], np.dtype(sys.argv[2]).type
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist

    import sparse_matrix_math_amd as smm
    from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    smm.init(0)
    dev = torch.device("cuda:0")
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    (start, pos, val), bounds = build_matrix(kind, dtype, world)
    n = len(start) - 1
    lo, hi = bounds[rank], bounds[rank + 1]
    x_true = np.random.default_rng(3).uniform(0.5, 1.5, n).astype(dtype)
    import scipy.sparse as sp

    b_full = (sp.csr_matrix((val.astype(np.float64), pos, start), shape=(n, n)) @ x_true.astype(np.float64)).astype(dtype)
    comm = NativeComm.gloo(dist)
    comm.selftest()
    d_start = torch.from_numpy((start[lo:hi + 1] - start[lo]).astype(np.int32)).to(dev)
    d_pos = torch.from_numpy(pos[start[lo]:start[hi]].copy()).to(dev)
    d_val = torch.from_numpy(val[start[lo]:start[hi]].copy()).to(dev)
    A = NativeDistMatrix(comm, n, bounds, d_start, d_pos, d_val, dtype)
    if len(sys.argv) > 3 and sys.argv[3] == "pattern":
        # both local blocks in the row-mask encoding (what the solvers adopt for blocks of >= 2^20 entries): SpMVs with a halo then run as ONE
        # launch (csrc/smm_spmv_split.hip) -- here with the exchange's word raised by another PROCESS's pushes landing
        for blk, lanes in zip(A.local_blocks(), (2, 1)):
            if blk.nnz > 0:
                blk.set_kernel(3, lanes)
    if kind == "grid":
        from sparse_matrix_math_amd import host

        host.set_cg_lazy_x_min_bytes(0)
        A.local_blocks()[0].set_kernel(3, 1)
    b = torch.from_numpy(b_full[lo:hi].copy()).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    results = {}

    def gather(t):
        mine = t.detach().cpu().numpy().copy()
        box = [None] * world if rank == 0 else None
        dist.gather_object(mine, box, dst=0)
        return np.concatenate(box).tobytes().hex() if rank == 0 else None

    y = torch.empty(hi - lo, dtype=tdt, device=dev)
    A.spmv(0, None, b, y, stream)
    A.spmv(0, None, y, y.clone(), stream)  # (two distributed SpMVs back to back: the second push waits for the first one's acknowledgement)
    A.spmv(0, None, b, y, stream)
    torch.cuda.synchronize()
    results["y"] = gather(y)
    for name, solver, precond, max_it in (("bicgstab7", "bicgstab", None, 7), ("jacobi7", "bicgstab", smm.SolverPreconditioner.JACOBI, 7), ("cg9", "cg", None, 9),
                                         ("bicgstab40", "bicgstab", None, 40)):
        A.set_precond(precond)
        x = torch.zeros(hi - lo, dtype=tdt, device=dev)
        eps = 1e-30 if max_it < 40 else 1e-6
        res = A.cg(b, x, x, max_it, eps, stream) if solver == "cg" else A.bicgstab(b, x, max_it, eps, stream)
        torch.cuda.synchronize()
        every = [None] * world
        dist.all_gather_object(every, [int(res[0]), int(res[1]), float(res[2])])
        assert all(e == every[0] for e in every), every  # every rank reports the same status / iterations / residual
        results[name] = {"res": every[0], "x": gather(x)}
    options = [None] * world
    dist.all_gather_object(options, dict(A.options, halo_elements=A.halo_elements, nnz_rem=A.nnz_rem, matvec_forms=list(A.matvec_forms()), thin_remote=list(A.thin_remote()),
                                         cg_fused=A.cg_fused()))
    A.set_precond(None)
    A.close()
    comm.close()
    if rank == 0:
        print("P2P_WORKER " + json.dumps({"kind": kind, "dtype": np.dtype(dtype).name, "world": world, "bounds": [int(v) for v in bounds], "options": options,
                                          "results": results}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
