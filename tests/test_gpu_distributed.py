"""The multi-GPU driver's product kernels (distributed.HipOps: fused SpMV + stage kernels of csrc/smm_stepwise.hip) on ONE
GPU: the ranks are threads of this process joined by tests/dist_helpers.ThreadComm (RCCL cannot put two ranks on one
device); a single-rank run over a real 1-rank RCCL group covers the TorchComm path."""
import threading

import numpy as np
import pytest
from dist_helpers import ThreadComm

from sparse_matrix_math_amd import generators as gen

pytestmark = pytest.mark.gpu


def _solve_threads(smm, csr, b_full, world, dtype, max_it, eps, precond=None):
    import torch

    from sparse_matrix_math_amd.distributed import DistBiCGStab, HipOps, partition_rows_by_nnz, plan_halo, split_local_remote

    dev = torch.device("cuda:0")
    start, pos, val = csr
    n = len(start) - 1
    bounds = partition_rows_by_nnz(lambda i: int(start[i]), n, world)
    shared = ThreadComm.Shared(world)
    out = [None] * world
    errors = []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = bounds[rank], bounds[rank + 1]
            lstart = torch.from_numpy((start[lo:hi + 1] - start[lo]).astype(np.int32)).to(dev)
            lpos = torch.from_numpy(pos[start[lo]:start[hi]].copy()).to(dev)
            lval = torch.from_numpy(val[start[lo]:start[hi]].copy()).to(dev)
            comm = ThreadComm(shared, rank, sync=torch.cuda.synchronize)
            cmin = min(int(lpos.min()), lo) if lpos.numel() else lo
            cmax = max(int(lpos.max()) + 1, hi) if lpos.numel() else hi
            needs = comm.all_gather_pairs(cmin, cmax, torch, dev)
            sends, recvs = plan_halo(bounds, needs, rank)
            loc, rem = split_local_remote(torch, lstart, lpos, lval, lo, hi, cmin)
            ops = HipOps(torch, loc, rem, n, lo, hi, cmin, cmax, dtype, dev, precond=precond)
            solver = DistBiCGStab(ops, comm, cmin, sends, recvs)
            x = torch.zeros(hi - lo, dtype=torch.float32 if dtype == np.float32 else torch.float64, device=dev)
            b = torch.from_numpy(b_full[lo:hi].copy()).to(dev)
            res = solver.solve(b, x, max_it, eps, check_every=1 << 30)
            torch.cuda.synchronize()
            out[rank] = (res, lo, hi, x.cpu().numpy())
            ops.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    x = np.zeros(n, dtype=dtype)
    for res, lo, hi, xs in out:
        x[lo:hi] = xs
        assert res == out[0][0]
    return out[0][0], x


@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_virtual_ranks_match_oracle(smm, oracle, world, dtype):
    cases = {
        "banded": gen.banded_random_spd(60000, k=12, seed=4, max_offset=9000, dtype=dtype),
        "convdiff": gen.convdiff3d(20, 0.3, dtype=dtype),
    }
    for name, csr in cases.items():
        n = len(csr[0]) - 1
        x_true = np.random.default_rng(3).uniform(0.5, 1.5, n).astype(dtype)
        b = oracle.spmv(csr, 0, None, x_true)
        for max_it in (1, 6):
            (status, iters, res), x = _solve_threads(smm, csr, b, world, dtype, max_it, 1e-30)
            st_ref, x_ref, it_ref, res_ref = oracle.bicgstab(csr, b, np.zeros(n, dtype=dtype), max_it, 1e-30)
            assert status == st_ref == 0 and iters == it_ref == max_it, (name, world)
            tol = 3e-4 if dtype == np.float32 else 1e-10
            assert float(np.max(np.abs(x - x_ref))) <= tol * float(np.max(np.abs(x_ref))), (name, world, max_it)
            assert abs(res - res_ref) <= 50 * tol * max(res_ref, 1e-30) + 1e-30
        if world != 2 or name != "convdiff":
            continue  # one converged run per dtype is enough (every step of the thread communicator is a device sync)
        eps = 1e-3 if dtype == np.float32 else 1e-8
        (status, iters, res), x = _solve_threads(smm, csr, b, world, dtype, -1, eps)
        assert status == 0 and res <= eps
        np.testing.assert_allclose(x, x_true, rtol=1e-3 if dtype == np.float32 else 1e-7)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_virtual_ranks_preconditioned(smm, oracle, world):
    """BiCGStab preconditioned block-Jacobi by rank (each rank's Jacobi / ILU0 / SGS of its diagonal block) through the
    product kernels.  One rank = the single-GPU preconditioned solver bit for bit; Jacobi is the same preconditioner for any
    number of ranks; SGS / ILU0 on several ranks are weaker preconditioners that must still converge to the solution."""
    P = smm.SolverPreconditioner
    dtype = np.float64
    csr = gen.convdiff3d(16, 0.3, dtype=dtype)
    n = len(csr[0]) - 1
    x_true = np.random.default_rng(7).uniform(0.5, 1.5, n)
    b = oracle.spmv(csr, 0, None, x_true)
    A = smm.CSRMatrix(n, n, *csr)
    eps = 1e-9
    (st0, it_none, _), _ = _solve_threads(smm, csr, b, world, dtype, -1, eps)
    for kind in (P.JACOBI, P.ILU0, P.SYMMETRIC_GAUS_SEIDEL):
        (status, iters, res), x = _solve_threads(smm, csr, b, world, dtype, -1, eps, precond=kind)
        assert status == 0 and res <= eps, (kind, world)
        np.testing.assert_allclose(x, x_true, rtol=1e-6, err_msg=f"{kind} world {world}")
        # BiCGStab's count is erratic (Jacobi on a constant diagonal is only a scaling, yet 66 vs 88 iterations were seen):
        # only a gross failure of the preconditioned recurrence is caught here; ILU0 / SGS must actually help
        assert iters <= 2 * it_none, (kind, world, iters, it_none)
        if kind != P.JACOBI:
            assert iters < it_none, (kind, world, iters, it_none)
        xs = np.zeros(n)
        info = {}
        st = smm.BiCGStab(A, b.copy(), xs, -1, eps, A.getPreconditioner(kind), info=info)
        if world == 1:
            # the same kernels in the same order: only the dot products differ (stage-wise partial sums vs fused epilogues)
            assert int(st) == 0 and abs(info["iterations"] - iters) <= 1
            np.testing.assert_allclose(x, xs, rtol=1e-7)
        elif kind != P.JACOBI:
            assert iters >= info["iterations"] - 3  # block-Jacobi by rank does not beat the global preconditioner by more than noise


def test_single_rank_rccl_group_and_row_range_generator(smm, oracle):
    """TorchComm over a real (1-rank) RCCL communicator + the per-rank generator of the benchmark matrix"""
    import torch
    import torch.distributed as dist

    from sparse_matrix_math_amd import host
    from sparse_matrix_math_amd.distributed import build_hip_solver, partition_rows_by_nnz

    dev = torch.device("cuda:0")
    n, k, seed, mo, shift = 50000, 10, 77, 6000, 1.0
    want = gen.banded_random_spd(n, k, seed, mo, np.float32, shift)
    row_start = lambda i: host.gen_banded_row_start(n, k, seed, mo, i)  # noqa: E731
    assert [row_start(i) for i in (0, 1, 777, n)] == [int(want[0][i]) for i in (0, 1, 777, n)]
    bounds = partition_rows_by_nnz(row_start, n, 3)
    for g in range(3):
        lo, hi = bounds[g], bounds[g + 1]
        nnz = row_start(hi) - row_start(lo)
        d_start = torch.empty(hi - lo + 1, dtype=torch.int32, device=dev)
        d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
        d_val = torch.empty(nnz, dtype=torch.float32, device=dev)
        host.gen_banded_rows_dev(n, k, seed, mo, shift, lo, hi, d_start, d_pos, d_val, np.float32, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d_start.cpu().numpy(), want[0][lo:hi + 1] - want[0][lo])
        np.testing.assert_array_equal(d_pos.cpu().numpy(), want[1][want[0][lo]:want[0][hi]])
        np.testing.assert_array_equal(d_val.cpu().numpy(), want[2][want[0][lo]:want[0][hi]])
    import os
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tens = [torch.from_numpy(a).to(dev) for a in want]
        solver = build_hip_solver(torch, dist, tens[0], tens[1], tens[2], [0, n], n, np.float32, dev)
        x_true = np.random.default_rng(3).uniform(0.5, 1.5, n).astype(np.float32)
        b = oracle.spmv(want, 0, None, x_true)
        x = torch.zeros(n, dtype=torch.float32, device=dev)
        status, iters, res = solver.solve(torch.from_numpy(b).to(dev), x, 5, 0.0)
        st_ref, x_ref, it_ref, _ = oracle.bicgstab(want, b, np.zeros(n, dtype=np.float32), 5, 0.0)
        assert status == st_ref and iters == it_ref == 5
        assert float(np.max(np.abs(x.cpu().numpy() - x_ref))) <= 3e-4 * float(np.max(np.abs(x_ref)))
    finally:
        dist.destroy_process_group()


def test_bench_two_rank_rehearsal():
    """bench.py's N > 1 leg end to end in two processes (the driver launches it the same way with RCCL): here the ranks share
    the one GPU and SMM_BENCH_BACKEND=gloo stages the halos through the host, so this checks function, not speed: per-rank
    generation of the row-partitioned matrix, halo plan, the stage-wise BiCGStab with real inter-process exchanges and
    all-reduces, the max-over-ranks timing and the one JSON line of rank 0"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--rows", "600000", "--max-offset", "65536",
           "--steps", "20", "--warmup", "10", "--iters-per-solve", "10"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["scaling"] == "strong"
    assert line["unit"] == "iterations/s" and line["value"] > 0
    assert line["rehearsal_backend"] == "gloo"
    assert line["per_rank"]["halo_elements"] > 0  # a real exchange took place
    assert line["max_rel_err_vs_x_true"] < 1e-3
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
