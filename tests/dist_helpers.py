"""Test-only pieces for the distributed driver: a numpy stand-in for the local kernels (CPU, uses the oracle's SpMV) and a
thread-based communicator that lets several ranks share one process (and one GPU)."""
import threading

import numpy as np

from sparse_matrix_math_amd.distributed import (STAGE_ALPHA_APPLY, STAGE_ALPHA_LOCAL, STAGE_BETA_APPLY, STAGE_INIT_APPLY, STAGE_INIT_LOCAL,
                                               STAGE_OMEGA_APPLY, STAGE_OMEGA_LOCAL)


class NumpyOps:
    """Same contract as distributed.HipOps, computed on the CPU with the oracle's SpMV: checks the driver's partition / halo /
    staging logic without a GPU.  Vectors are torch CPU tensors (so gloo can move them); arithmetic is numpy on their memory."""

    def __init__(self, torch, oracle, loc, rem, n_global, own_lo, own_hi, cmin, cmax_excl, dtype, precond=None):
        self.torch, self.oracle = torch, oracle
        self.precond = precond  # None | "jacobi" | "sgs" | "ilu0": applied to this rank's diagonal block (block-Jacobi by rank)
        self.has_precond = precond is not None
        self.dtype = np.dtype(dtype)
        tdt = torch.float32 if self.dtype == np.float32 else torch.float64
        self.n_global, self.n_local = n_global, own_hi - own_lo
        self.own_offset = own_lo - cmin
        ext_len = max(1, cmax_excl - cmin)
        self.csr = {"loc": tuple(t.numpy() for t in loc), "rem": tuple(t.numpy() for t in rem)}
        self.x_ext, self.p_ext, self.s_ext = (torch.zeros(ext_len, dtype=tdt) for _ in range(3))
        self.r, self.r0, self.ap, self.as_ = (torch.zeros(self.n_local, dtype=tdt) for _ in range(4))
        self.sums = torch.zeros(4, dtype=tdt)
        self.partial = np.zeros(2, dtype=self.dtype)
        self.c = {"rr0": 0.0, "alpha": 0.0, "omega": 0.0, "beta": 0.0, "res": 0.0}
        self.done, self.iters = 0, 0
        if self.has_precond:
            self.scratch = torch.zeros(self.n_local, dtype=tdt)
            if precond == "jacobi":
                err, self.diag = oracle.jacobi_setup(self.csr["loc"])
                assert err == 0
            elif precond == "ilu0":
                err, self.lu = oracle.ilu0_factorize(self.csr["loc"])
                assert err == 0

    def precond_apply(self, src, dst):
        rhs = src.numpy().copy()
        if self.precond == "jacobi":
            out = self.oracle.jacobi_apply(self.diag, rhs)
        elif self.precond == "sgs":
            err, out = self.oracle.sgs_apply(self.csr["loc"], rhs)
            assert err == 0
        else:
            err, out = self.oracle.ilu0_apply(self.csr["loc"], self.lu, rhs)
            assert err == 0
        dst.numpy()[:] = out

    def dot_into(self, a, b, k):
        self.sums.numpy()[k] = np.dot(a.numpy(), b.numpy())

    def own(self, ext):
        return ext[self.own_offset:self.own_offset + self.n_local]

    def copy_into_ext(self, ext, own_values):
        self.own(ext).copy_(own_values)

    def spmv(self, which, op, lhs, x, out, dot_mode, w1):
        if self.done:
            return
        o = self.oracle.spmv(self.csr[which], op, None if lhs is None else lhs.numpy().copy(), x.numpy())
        out.numpy()[:] = o
        if dot_mode == 1:
            self.partial[0] = np.dot(o, w1.numpy())
        elif dot_mode == 2:
            self.partial[0] = np.dot(o, o)
            self.partial[1] = np.dot(o, w1.numpy())

    def stage(self, stage, x_own, eps):
        t = self.dtype.type
        r, r0, ap, as_ = (v.numpy() for v in (self.r, self.r0, self.ap, self.as_))
        p, s, sums, x = self.own(self.p_ext).numpy(), self.own(self.s_ext).numpy(), self.sums.numpy(), x_own.numpy()
        if stage == STAGE_INIT_LOCAL:
            r0[:] = r
            p[:] = r
            sums[0] = np.dot(r, r)
        elif stage == STAGE_INIT_APPLY:
            self.c["rr0"], self.done, self.iters = sums[0], 0, 0
        elif self.done:
            return
        elif stage == STAGE_ALPHA_LOCAL:
            sums[0] = self.partial[0]
        elif stage == STAGE_ALPHA_APPLY:
            self.c["alpha"] = t(self.c["rr0"]) / sums[0]
            s[:] = -t(self.c["alpha"]) * ap + r
        elif stage == STAGE_OMEGA_LOCAL:
            sums[:2] = self.partial
        elif stage == STAGE_OMEGA_APPLY:
            self.c["omega"] = sums[1] / sums[0]
            a, w = t(self.c["alpha"]), t(self.c["omega"])
            x[:] = a * p + (w * s + x)
            r[:] = -w * as_ + s
            sums[0], sums[1] = np.dot(r, r), np.dot(r, r0)
        elif stage == STAGE_BETA_APPLY:
            res = np.sqrt(sums[0])
            self.c["beta"] = (sums[1] * t(self.c["alpha"])) / (t(self.c["rr0"]) * t(self.c["omega"]))
            self.c["rr0"], self.c["res"] = sums[1], res
            self.iters += 1
            if not res > eps:
                self.done = 1
                return
            p[:] = t(self.c["beta"]) * (-t(self.c["omega"]) * ap + p) + r

    def cg_stage(self, stage, xcur_own, x_own, eps):
        t = self.dtype.type
        r, ap = self.r.numpy(), self.ap.numpy()
        p, sums = self.own(self.p_ext).numpy(), self.sums.numpy()
        if stage == 1:  # CG_INIT_LOCAL
            p[:] = r
            sums[0] = np.dot(r, r)
        elif stage == 2:  # CG_INIT_APPLY
            self.c["rr0"] = self.c["res"] = sums[0]
            self.iters, self.done, self.status = 0, 0, 2
            if t(eps) * t(eps) > sums[0]:
                self.done, self.status = 1, 0
        elif self.done:
            return
        elif stage == 3:  # CG_ALPHA_LOCAL
            sums[0] = self.partial[0]
        elif stage == 4:  # CG_ALPHA_APPLY
            a = t(self.c["rr0"]) / sums[0]
            x_own.numpy()[:] = a * p + xcur_own.numpy()
            r[:] = -a * ap + r
            sums[0] = np.dot(r, r)
        elif stage == 5:  # CG_BETA_APPLY
            self.iters += 1
            self.c["res"] = sums[0]
            if t(eps) * t(eps) > sums[0]:
                self.done, self.status = 1, 0
                return
            beta = sums[0] / t(self.c["rr0"])
            self.c["rr0"] = sums[0]
            p[:] = beta * p + r

    def cg_status(self):
        return self.status

    def result(self):
        return self.done, self.iters, float(self.c["res"])


class ThreadComm:
    """ranks = threads of one process: all-reduce and halo exchange through shared memory and a barrier.  Lets a test drive
    distributed.HipOps for several ranks on ONE GPU (RCCL refuses two ranks on the same device)."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.ext = [None] * world
            self.cmin = [0] * world

    def __init__(self, shared, rank, sync=None):
        self.shared, self.rank, self.world, self.sync = shared, rank, shared.world, sync

    def _sync(self):
        if self.sync:
            self.sync()  # device-wide: the other threads' kernels on this GPU must be finished before their data is read

    def all_reduce_sum(self, t):
        sh = self.shared
        self._sync()
        sh.slots[self.rank] = t.clone()
        sh.barrier.wait()
        total = sh.slots[0].clone()
        for q in range(1, self.world):
            total += sh.slots[q]
        sh.barrier.wait()
        t.copy_(total)
        self._sync()

    def all_gather_pairs(self, a, b, torch, device):
        sh = self.shared
        sh.slots[self.rank] = (a, b)
        sh.barrier.wait()
        out = list(sh.slots)
        sh.barrier.wait()
        return out

    def exchange(self, ext, cmin, sends, recvs):
        sh = self.shared
        self._sync()
        sh.ext[self.rank], sh.cmin[self.rank] = ext, cmin
        sh.barrier.wait()
        for q, lo, hi in recvs:
            ext[lo - cmin:hi - cmin].copy_(sh.ext[q][lo - sh.cmin[q]:hi - sh.cmin[q]])
        self._sync()
        sh.barrier.wait()
        return []
