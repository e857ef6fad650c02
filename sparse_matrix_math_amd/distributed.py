"""Row-partitioned multi-GPU BiCGStab: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference is a single-process CPU library (no collectives); this is the MI355X-native scale-out
of its BiCGStab loop (ref:2191-2283):

  * rank g owns a contiguous range of rows [bounds[g], bounds[g+1]) of A, balanced by nonzeros, and the matching slices of
    every vector;
  * before each SpMV the x-vector HALO is exchanged point to point: a rank needs x only on the column range its rows touch
    ([cmin, cmax]); the parts of that range owned by other ranks are received straight into a halo-extended buffer
    [left halo | owned | right halo] (`batch_isend_irecv`).  For a banded / stencil matrix that is a few MB from one or two
    neighbours; for a matrix with global coupling it degenerates into an all-gather built from the same primitives;
  * the local rows are split once into A_loc (columns this rank owns) and A_rem (halo columns): A_loc p runs while the halo
    is in flight, A_rem p is added when it has landed (rMultAdd in place), with the dot products fused into that last launch;
  * the scalars of the recurrence are completed with an all-reduce of 1-2 numbers at the three reduction points of an
    iteration; every scalar stays in HBM (csrc/smm_stepwise.hip), the host never reads one inside the loop.

The driver (`DistBiCGStab`) is written against two small interfaces -- local kernels (`ops`) and collectives (`comm`) -- so
the partition / halo / staging logic is exercised on CPU by tests/test_distributed_gloo.py (gloo, world_size 2) with
stand-in local kernels.  The product implementation of `ops` is `HipOps` (libsmm_hip.so); there is no CPU implementation
in this package.
"""
import ctypes

import numpy as np

STAGE_INIT_LOCAL, STAGE_INIT_APPLY, STAGE_ALPHA_LOCAL, STAGE_ALPHA_APPLY, STAGE_OMEGA_LOCAL, STAGE_OMEGA_APPLY, STAGE_BETA_APPLY = range(1, 8)
OP_ASSIGN, OP_ADD, OP_SUB = 0, 1, 2


# ---------------------------------------------------------------------------------------------------------------------
# partition and halo plan (pure Python: no device, no communication)
# ---------------------------------------------------------------------------------------------------------------------
def partition_rows_by_nnz(row_start, n, world):
    """bounds[0..world]: contiguous row ranges with ~equal nonzeros.  row_start(i) = start[i] of the full matrix."""
    total = row_start(n)
    bounds = [0]
    for g in range(1, world):
        target = total * g // world
        lo, hi = bounds[-1], n
        while lo < hi:  # smallest row with row_start(row) >= target
            mid = (lo + hi) // 2
            if row_start(mid) >= target:
                hi = mid
            else:
                lo = mid + 1
        bounds.append(lo)
    bounds.append(n)
    return bounds


def plan_halo(bounds, needs, rank):
    """needs[q] = (cmin, cmax_exclusive) column range rank q's rows touch.  Returns (sends, recvs): lists of
    (peer, lo, hi) global column ranges this rank sends from its owned slice / receives into its halo."""
    own_lo, own_hi = bounds[rank], bounds[rank + 1]
    recvs, sends = [], []
    for q in range(len(bounds) - 1):
        if q == rank:
            continue
        lo, hi = max(needs[rank][0], bounds[q]), min(needs[rank][1], bounds[q + 1])
        if lo < hi:
            recvs.append((q, lo, hi))
        lo, hi = max(needs[q][0], own_lo), min(needs[q][1], own_hi)
        if lo < hi:
            sends.append((q, lo, hi))
    return sends, recvs


def split_local_remote(torch, start, positions, values, own_lo, own_hi, cmin):
    """Split the local rows into A_loc (global column in [own_lo, own_hi), renumbered from own_lo) and A_rem (all other
    columns, renumbered from cmin, the first column of the halo-extended buffer).  Order inside a row is preserved."""
    nloc = start.numel() - 1
    lens = (start[1:] - start[:-1]).to(torch.int64)
    rows_of = torch.repeat_interleave(torch.arange(nloc, device=start.device, dtype=torch.int64), lens)
    is_loc = (positions >= own_lo) & (positions < own_hi)
    out = []
    for mask, shift in ((is_loc, own_lo), (~is_loc, cmin)):
        counts = torch.bincount(rows_of[mask], minlength=nloc)
        st = torch.zeros(nloc + 1, dtype=torch.int32, device=start.device)
        st[1:] = torch.cumsum(counts, 0).to(torch.int32)
        out.append((st, (positions[mask] - shift).to(torch.int32).contiguous(), values[mask].contiguous()))
    return out[0], out[1]


# ---------------------------------------------------------------------------------------------------------------------
# collectives
# ---------------------------------------------------------------------------------------------------------------------
class TorchComm:
    """torch.distributed: "nccl" (RCCL over xGMI) on the GPUs, "gloo" in the CPU tests"""

    def __init__(self, dist, group=None):
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # gloo has no point-to-point or all-gather on device tensors: a multi-process REHEARSAL of the GPU path on a box without
        # RCCL peers (several ranks sharing one GPU, bench.py with SMM_BENCH_BACKEND=gloo) stages those through host memory
        self.staged = dist.get_backend(group) == "gloo"

    def all_reduce_sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_gather_pairs(self, a, b, torch, device):
        mine = torch.tensor([a, b], dtype=torch.int64, device="cpu" if self.staged else device)
        if self.world == 1:
            return [(a, b)]
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine, group=self.group)
        return [(int(t[0]), int(t[1])) for t in out]

    def _exchange_staged(self, ext, cmin, sends, recvs):
        class _Landing:
            def __init__(self, req, dst, buf):
                self.req, self.dst, self.buf = req, dst, buf

            def wait(self):
                self.req.wait()
                if self.dst is not None:
                    self.dst.copy_(self.buf)

        out = []
        for q, lo, hi in recvs:
            dst = ext[lo - cmin:hi - cmin]
            buf = dst.new_empty(dst.shape, device="cpu")
            out.append(_Landing(self.dist.irecv(buf, q, group=self.group), dst, buf))
        for q, lo, hi in sends:
            buf = ext[lo - cmin:hi - cmin].cpu()
            out.append(_Landing(self.dist.isend(buf, q, group=self.group), None, buf))
        return out

    def exchange(self, ext, cmin, sends, recvs):
        """start the halo exchange of `ext` (the halo-extended vector whose element 0 is global column cmin); returns a
        list of requests to wait on"""
        if not sends and not recvs:
            return []
        if self.staged and ext.is_cuda:
            return self._exchange_staged(ext, cmin, sends, recvs)
        # the same few vectors are exchanged every iteration: build their P2POp lists (slice views + op objects) once
        cache = getattr(self, "_p2p_cache", None)
        if cache is None:
            cache = self._p2p_cache = {}
        key = (ext.data_ptr(), cmin, id(sends), id(recvs))
        ops = cache.get(key)
        if ops is None:
            ops = []
            for q, lo, hi in recvs:
                ops.append(self.dist.P2POp(self.dist.irecv, ext[lo - cmin:hi - cmin], q, self.group))
            for q, lo, hi in sends:
                ops.append(self.dist.P2POp(self.dist.isend, ext[lo - cmin:hi - cmin], q, self.group))
            cache[key] = ops
        return self.dist.batch_isend_irecv(ops)


# ---------------------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------------------
class DistBiCGStab:
    """BiCGStab (ref:2191-2283) on a row-partitioned matrix.

    ops must provide:  spmv(which, op, lhs, x, out, dot_mode, w1)  with which in {"loc", "rem"};  stage(stage, x, eps);
    result() -> (done, iterations, resnorm);  attributes p_ext, s_ext, x_ext (halo-extended vectors), r, ap, as_, r0
    (handles understood by spmv), sums (tensor of >= 2 scalars the collectives reduce), own_offset, n_local.
    """

    def __init__(self, ops, comm, cmin, sends, recvs):
        self.ops, self.comm = ops, comm
        self.cmin, self.sends, self.recvs = cmin, sends, recvs

    def _matvec(self, ext, own_view, out, op_first, lhs_first, dot_mode, w1):
        """out = op(lhs, A ext): A_loc on the owned slice while the halo is in flight, then A_rem added in place with the
        dot products fused into that launch"""
        if getattr(self.ops, "rem_empty", False) and not self.sends and not self.recvs:
            self.ops.spmv("loc", op_first, lhs_first, own_view, out, dot_mode, w1)  # nothing lives on another rank
            return
        reqs = self.comm.exchange(ext, self.cmin, self.sends, self.recvs)
        self.ops.spmv("loc", op_first, lhs_first, own_view, out, 0, None)
        for r in reqs:
            r.wait()
        second = OP_SUB if op_first == OP_SUB else OP_ADD
        self.ops.spmv("rem", second, out, ext, out, dot_mode, w1)

    def solve(self, b, x_own, max_iterations, eps, check_every=8):
        """x_own: this rank's slice of x (in/out).  Returns (status, iterations, resnorm) -- identical on every rank."""
        ops, comm = self.ops, self.comm
        n_global = ops.n_global
        max_iterations = min(max_iterations, n_global)  # ref:2200
        if max_iterations == -1:
            max_iterations = n_global  # ref:2201-2203
        if hasattr(ops, "begin_solve"):
            ops.begin_solve()
        # views used every iteration, made once (the loop below is host-side latency critical at 8 GPUs)
        sums1, sums2 = ops.sums[:1], ops.sums[:2]
        x_view, p_own, s_own = ops.own(ops.x_ext), ops.own(ops.p_ext), ops.own(ops.s_ext)
        # Preconditioned form (ref:2209-2224, 2234-2235, 2250-2251): every A v is followed by M^-1.  M is BLOCK-JACOBI BY RANK:
        # each rank applies its preconditioner (Jacobi / ILU0 / SGS) of its own diagonal block A_loc, no communication; with
        # one rank it is the single-GPU preconditioner, with more ranks a weaker one (iteration counts differ).
        pre = bool(getattr(ops, "has_precond", False))
        # r = b - A x (ref:2215)
        ops.copy_into_ext(ops.x_ext, x_own)
        if pre:
            self._matvec(ops.x_ext, x_view, ops.scratch, OP_SUB, b, 0, None)
            ops.precond_apply(ops.scratch, ops.r)  # ref:2217-2224
        else:
            self._matvec(ops.x_ext, x_view, ops.r, OP_SUB, b, 0, None)
        ops.stage(STAGE_INIT_LOCAL, x_own, eps)  # r0 = p = r, local r.r0
        comm.all_reduce_sum(sums1)
        ops.stage(STAGE_INIT_APPLY, x_own, eps)
        planned = max(1, max_iterations)  # do { } while: the body always runs once (ref:2232, 2277)
        done = 0
        for it in range(planned):
            if it and it % check_every == 0:
                done, _, _ = ops.result()
                if done:
                    break
            if pre:
                self._matvec(ops.p_ext, p_own, ops.scratch, OP_ASSIGN, None, 0, None)  # ref:2234
                ops.precond_apply(ops.scratch, ops.ap)  # ap = M^-1 A p, ref:2235
                ops.dot_into(ops.ap, ops.r0, 0)  # local ap.r0, ref:2243
            else:
                self._matvec(ops.p_ext, p_own, ops.ap, OP_ASSIGN, None, 1, ops.r0)  # ap = A p, local ap.r0
                ops.stage(STAGE_ALPHA_LOCAL, x_own, eps)
            comm.all_reduce_sum(sums1)
            ops.stage(STAGE_ALPHA_APPLY, x_own, eps)  # alpha, s
            if pre:
                self._matvec(ops.s_ext, s_own, ops.scratch, OP_ASSIGN, None, 0, None)  # ref:2250
                ops.precond_apply(ops.scratch, ops.as_)  # as = M^-1 A s, ref:2251
                ops.dot_into(ops.as_, ops.as_, 0)  # local as.as and as.s, ref:2259-2261
                ops.dot_into(ops.as_, s_own, 1)
            else:
                self._matvec(ops.s_ext, s_own, ops.as_, OP_ASSIGN, None, 2, s_own)  # as = A s, as.as, as.s
                ops.stage(STAGE_OMEGA_LOCAL, x_own, eps)
            comm.all_reduce_sum(sums2)
            ops.stage(STAGE_OMEGA_APPLY, x_own, eps)  # omega, x, r, local ||r||^2 and r.r0
            comm.all_reduce_sum(sums2)
            ops.stage(STAGE_BETA_APPLY, x_own, eps)  # res, beta, p
        done, iterations, resnorm = ops.result()
        status = 2 if iterations > max_iterations else 0  # ref:2279-2282
        return status, iterations, resnorm


CG_INIT_LOCAL, CG_INIT_APPLY, CG_ALPHA_LOCAL, CG_ALPHA_APPLY, CG_BETA_APPLY = range(1, 6)


class DistCG(DistBiCGStab):
    """ConjugateGradient (ref:2316-2398) on a row-partitioned matrix (BASELINE config 4: CG with the dot products completed by
    an RCCL all-reduce).  ops additionally provides cg_stage(stage, xcur, x, eps) and cg_status()."""

    def solve(self, b, x0_own, x_own, max_iterations, eps, check_every=8):
        """x0_own: initial guess; x_own: result (may be the same tensor).  Returns (status, iterations, resnorm2)."""
        ops, comm = self.ops, self.comm
        if hasattr(ops, "begin_solve"):
            ops.begin_solve()
        if max_iterations == -1:
            max_iterations = ops.n_global  # ref:2345-2347 (no clamp otherwise)
        sums1 = ops.sums[:1]
        x_view, p_own = ops.own(ops.x_ext), ops.own(ops.p_ext)
        ops.copy_into_ext(ops.x_ext, x0_own)
        self._matvec(ops.x_ext, x_view, ops.r, OP_SUB, b, 0, None)  # r = b - A x0, ref:2337
        ops.cg_stage(CG_INIT_LOCAL, x0_own, x_own, eps)  # p = r, local r.r
        comm.all_reduce_sum(sums1)
        ops.cg_stage(CG_INIT_APPLY, x0_own, x_own, eps)  # eps^2 > ||r||^2 -> done, x untouched (ref:2342-2344)
        for it in range(max(0, max_iterations)):
            if it % check_every == 0:
                done, _, _ = ops.result()
                if done:
                    break
            self._matvec(ops.p_ext, p_own, ops.ap, OP_ASSIGN, None, 1, p_own)  # Ap = A p, local p.Ap
            ops.cg_stage(CG_ALPHA_LOCAL, x0_own, x_own, eps)
            comm.all_reduce_sum(sums1)
            ops.cg_stage(CG_ALPHA_APPLY, x0_own if it == 0 else x_own, x_own, eps)  # alpha, x, r, local ||r||^2 (ref:2351, 2395)
            comm.all_reduce_sum(sums1)
            ops.cg_stage(CG_BETA_APPLY, x0_own, x_own, eps)  # test, beta, p
        _, iterations, resnorm2 = ops.result()
        return ops.cg_status(), iterations, resnorm2


# ---------------------------------------------------------------------------------------------------------------------
# product implementation of the local kernels: libsmm_hip.so
# ---------------------------------------------------------------------------------------------------------------------
class HipOps:
    """Local kernels on one MI355X through the C ABI (csrc/smm_stepwise.hip, smm_spmv.hip).  Vectors are torch tensors
    (device memory management only); all launches go to torch's current stream so RCCL and the kernels are ordered."""

    def __init__(self, torch, loc, rem, n_global, own_lo, own_hi, cmin, cmax_excl, np_dtype, device, precond=None):
        from . import _lib, host

        self.torch, self.host, self.lib = torch, host, _lib.load()
        self.check = _lib.check
        self.np_dtype = np.dtype(np_dtype)
        self.suf = host._suffix(np_dtype)
        tdt = torch.float32 if self.np_dtype == np.float32 else torch.float64
        self.n_global, self.n_local = n_global, own_hi - own_lo
        self.own_offset = own_lo - cmin
        ext_len = max(1, cmax_excl - cmin)
        self.A_loc = host.CSRMatrix.from_device(self.n_local, self.n_local, *loc, np_dtype)
        self.A_rem = host.CSRMatrix.from_device(self.n_local, ext_len, *rem, np_dtype)
        self.x_ext = torch.zeros(ext_len, dtype=tdt, device=device)
        self.p_ext = torch.zeros(ext_len, dtype=tdt, device=device)
        self.s_ext = torch.zeros(ext_len, dtype=tdt, device=device)
        self.sums = torch.zeros(4, dtype=tdt, device=device)
        self.rem_empty = self.A_rem.nnz == 0
        self._cached_stream = None
        self.ws = ctypes.c_void_p()
        self.check(getattr(self.lib, f"smm_hip_bicgstab_ws_create_{self.suf}")(self.n_local, ctypes.byref(self.ws)))
        self.check(self.lib.smm_hip_bicgstab_ws_bind(self.ws, host._dptr(self.own(self.p_ext)), host._dptr(self.own(self.s_ext)), host._dptr(self.sums)))
        ptrs = [ctypes.c_void_p() for _ in range(6)]
        self.check(self.lib.smm_hip_bicgstab_ws_pointers(self.ws, *[ctypes.byref(p) for p in ptrs]))
        self.r, self.r0, self.ap, self.as_, self.partials, _ = (p.value for p in ptrs)
        self._spmv = getattr(self.lib, f"smm_hip_spmv_fused_dev_{self.suf}")
        self._stage = getattr(self.lib, f"smm_hip_bicgstab_ws_stage_{self.suf}")
        self._result = getattr(self.lib, f"smm_hip_bicgstab_ws_result_{self.suf}")
        self._cg_stage = getattr(self.lib, f"smm_hip_cg_ws_stage_{self.suf}")
        # block-Jacobi preconditioning: this rank's preconditioner of its diagonal block A_loc (square, local column numbers)
        self.has_precond = precond is not None and int(precond) != 0
        if self.has_precond:
            self.M = host.Preconditioner(self.A_loc, precond)
            self.scratch = torch.zeros(max(1, self.n_local), dtype=tdt, device=device)
            self._apply = getattr(self.lib, f"smm_hip_precond_apply_dev_{self.suf}")
            self._dot = getattr(self.lib, f"smm_hip_dot_dev_{self.suf}")
            self._elem = self.np_dtype.itemsize

    def own(self, ext):
        return ext[self.own_offset:self.own_offset + self.n_local]

    def copy_into_ext(self, ext, own_values):
        self.own(ext).copy_(own_values)

    def precond_apply(self, src, dst):
        d = self.host._dptr
        self.check(self._apply(self.M._h, d(src), d(dst), self._stream()))

    def dot_into(self, a, b, k):
        """sums[k] = a . b over this rank's rows (fixed-order partial sums, no atomics)"""
        d = self.host._dptr
        self.check(self._dot(self.n_local, d(a), d(b), ctypes.c_void_p(self.sums.data_ptr() + k * self._elem), self._stream()))

    def begin_solve(self):
        self._cached_stream = ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def _stream(self):
        return self._cached_stream if self._cached_stream is not None else ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def spmv(self, which, op, lhs, x, out, dot_mode, w1):
        A = self.A_loc if which == "loc" else self.A_rem
        d = self.host._dptr
        self.check(self._spmv(A._h, op, d(lhs), d(x), d(out), dot_mode, d(w1), d(self.partials), self._stream()))

    def stage(self, stage, x_own, eps):
        self.check(self._stage(self.ws, stage, self.host._dptr(x_own), self.np_dtype.type(eps), self._stream()))

    def cg_stage(self, stage, xcur_own, x_own, eps):
        d = self.host._dptr
        self.check(self._cg_stage(self.ws, stage, d(xcur_own), d(x_own), self.np_dtype.type(eps), self._stream()))

    def cg_status(self):
        st = ctypes.c_int()
        self.check(self.lib.smm_hip_cg_ws_status(self.ws, self._stream(), ctypes.byref(st)))
        return st.value

    def result(self):
        done, it = ctypes.c_int(), ctypes.c_int()
        res = (ctypes.c_float if self.suf == "f32" else ctypes.c_double)()
        self.check(self._result(self.ws, self._stream(), ctypes.byref(done), ctypes.byref(it), ctypes.byref(res)))
        if self.has_precond:  # a triangular sweep that ran into its escape bound must not pass as a result (NaNs otherwise)
            self.check(self.lib.smm_hip_precond_take_error(self.M._h, self._stream()))
        return done.value, it.value, res.value

    def close(self):
        if self.ws:
            self.lib.smm_hip_bicgstab_ws_destroy(self.ws)
            self.ws = ctypes.c_void_p()


def build_hip_solver(torch, dist, start, positions, values, bounds, n_global, np_dtype, device, group=None, solver="bicgstab", precond=None):
    """start/positions/values: this rank's rows (local start[], GLOBAL columns) as device tensors.  Collective.
    solver: "bicgstab" (DistBiCGStab) or "cg" (DistCG).  precond (BiCGStab only): a SolverPreconditioner kind (JACOBI / ILU0 /
    SYMMETRIC_GAUS_SEIDEL) applied block-Jacobi by rank."""
    comm = TorchComm(dist, group)
    rank = comm.rank
    own_lo, own_hi = bounds[rank], bounds[rank + 1]
    if positions.numel():
        cmin = min(int(positions.min()), own_lo)
        cmax_excl = max(int(positions.max()) + 1, own_hi)
    else:
        cmin, cmax_excl = own_lo, own_hi
    needs = comm.all_gather_pairs(cmin, cmax_excl, torch, device)
    sends, recvs = plan_halo(bounds, needs, rank)
    loc, rem = split_local_remote(torch, start, positions, values, own_lo, own_hi, cmin)
    if precond is not None and solver == "cg":
        raise ValueError("the distributed CG has no preconditioned form (the reference's PCG takes IC0 only, ref:2414)")
    ops = HipOps(torch, loc, rem, n_global, own_lo, own_hi, cmin, cmax_excl, np_dtype, device, precond=precond)
    solver = (DistCG if solver == "cg" else DistBiCGStab)(ops, comm, cmin, sends, recvs)
    solver.halo_elements = sum(hi - lo for _, lo, hi in recvs)
    return solver


# ---------------------------------------------------------------------------------------------------------------------
# native driver: the whole loop behind the C ABI (csrc/smm_dist.hip) -- Python only sets it up
# ---------------------------------------------------------------------------------------------------------------------
class NativeComm:
    """smm_hip_comm: RCCL over xGMI (`rccl`), host callbacks carried by torch.distributed gloo (`gloo`: several ranks may share a
    GPU; a rehearsal, never a measurement), arbitrary Python callbacks (`host`), or a single rank (`single`)."""

    def __init__(self, handle, keep=()):
        self._h = handle
        self._keep = keep  # callback objects must outlive the communicator

    @classmethod
    def single(cls):
        from . import _lib

        h = ctypes.c_void_p()
        _lib.check(_lib.load().smm_hip_comm_create_self(ctypes.byref(h)))
        return cls(h)

    @classmethod
    def rccl(cls, dist, group=None):
        """rank 0 draws the RCCL unique id, torch.distributed carries it to the other ranks (any backend)"""
        from . import _lib

        lib = _lib.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ident = ctypes.create_string_buffer(128)
        box = [None]
        if rank == 0:
            # a failure here (librccl not found ...) must reach every rank: the others are about to wait in the broadcast
            status = lib.smm_hip_comm_unique_id(ident)
            box = [ident.raw] if status == 0 else [RuntimeError(lib.smm_hip_last_error().decode("utf-8", "replace"))]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if isinstance(box[0], Exception):
            raise box[0]
        h = ctypes.c_void_p()
        _lib.check(lib.smm_hip_comm_create_rccl(rank, world, ctypes.create_string_buffer(box[0], 128), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def host(cls, rank, world, allreduce, sendrecv):
        """allreduce(np_array) sums in place over the ranks; sendrecv(sends, recvs) with lists of (peer, np.uint8 array)"""
        from . import _lib

        np_of = {0: np.float32, 1: np.float64, 2: np.int64}

        def _ar(_user, buf, count, dtype):
            try:
                ct = np.ctypeslib.as_ctypes_type(np_of[dtype])
                allreduce(np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ct)), (count,)))
                return 0
            except Exception:  # noqa: BLE001 -- a Python exception must not unwind through the C frames
                import traceback

                traceback.print_exc()
                return 1

        def _sr(_user, ns, speer, sbuf, sbytes, nr, rpeer, rbuf, rbytes):
            try:
                def view(ptr, n):
                    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_ubyte)), (n,)) if n else np.empty(0, np.uint8)

                sends = [(speer[i], view(sbuf[i], sbytes[i])) for i in range(ns)]
                recvs = [(rpeer[i], view(rbuf[i], rbytes[i])) for i in range(nr)]
                sendrecv(sends, recvs)
                return 0
            except Exception:  # noqa: BLE001
                import traceback

                traceback.print_exc()
                return 1

        cb1, cb2 = _lib.HOST_ALLREDUCE_FN(_ar), _lib.HOST_SENDRECV_FN(_sr)
        h = ctypes.c_void_p()
        _lib.check(_lib.load().smm_hip_comm_create_host(int(rank), int(world), cb1, cb2, None, ctypes.byref(h)))
        return cls(h, keep=(cb1, cb2))

    @classmethod
    def gloo(cls, dist, group=None):
        import torch

        def allreduce(a):
            dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM, group=group)

        def sendrecv(sends, recvs):
            reqs = [dist.irecv(torch.from_numpy(buf), src=peer, group=group) for peer, buf in recvs]
            reqs += [dist.isend(torch.from_numpy(buf), dst=peer, group=group) for peer, buf in sends]
            for r in reqs:
                r.wait()

        return cls.host(dist.get_rank(group), dist.get_world_size(group), allreduce, sendrecv)

    def info(self):
        from . import _lib

        r, w, k = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().smm_hip_comm_info(self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(k)))
        n = ctypes.c_int()
        _lib.check(_lib.load().smm_hip_comm_rccl_ranks(self._h, ctypes.byref(n)))  # ncclCommCount: 0 unless this is an RCCL communicator
        return {"rank": r.value, "world": w.value, "kind": {0: "self", 1: "rccl", 2: "host"}[k.value], "rccl_ranks": n.value}

    def selftest(self):
        from . import _lib

        _lib.check(_lib.load().smm_hip_comm_selftest(self._h))

    def close(self):
        if self._h:
            from . import _lib

            _lib.load().smm_hip_comm_destroy(self._h)
            self._h = ctypes.c_void_p()


class _BlockView:
    """non-owning view of one of a distributed matrix's local blocks (enough of host.CSRMatrix for host.Preconditioner / set_kernel)"""

    def __init__(self, handle, owner, np_dtype):
        from . import host

        self._h, self._owner = handle, owner
        self.dtype = np.dtype(np_dtype)
        self._suf = host._suffix(np_dtype)
        host.CSRMatrix._read_info(self)

    def set_kernel(self, family=0, lanes_per_row=0):
        from . import _lib

        _lib.check(_lib.load().smm_hip_csr_set_kernel(self._h, int(family), int(lanes_per_row)))

    def get_kernel(self):
        from . import _lib

        fam, lanes = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().smm_hip_csr_get_kernel(self._h, ctypes.byref(fam), ctypes.byref(lanes)))
        return fam.value, lanes.value

    def pattern_info(self):
        from . import _lib

        enc, k = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().smm_hip_csr_pattern_info(self._h, ctypes.byref(enc), ctypes.byref(k)))
        return enc.value, k.value

    def kernel_desc(self):
        from . import host

        return host.CSRMatrix.kernel_desc(self)


class NativeDistMatrix:
    """smm_hip_dist_csr: this rank's rows of a row-partitioned matrix + the native solvers on it.  Collective."""

    def __init__(self, comm, n_global, bounds, d_start, d_positions, d_values, np_dtype):
        from . import _lib, host

        self.comm, self.lib, self.check, self.host = comm, _lib.load(), _lib.check, host
        self.np_dtype = np.dtype(np_dtype)
        self.suf = host._suffix(np_dtype)
        self.n_global = int(n_global)
        self._h = ctypes.c_void_p()
        b = (ctypes.c_int * len(bounds))(*[int(v) for v in bounds])
        self.check(getattr(self.lib, f"smm_hip_dist_csr_create_dev_{self.suf}")(comm._h, self.n_global, b, host._dptr(d_start), host._dptr(d_positions),
                                                                               host._dptr(d_values), ctypes.byref(self._h)))
        n, e, o, h = (ctypes.c_int() for _ in range(4))
        nl, nr = ctypes.c_longlong(), ctypes.c_longlong()
        self.check(self.lib.smm_hip_dist_csr_info(self._h, ctypes.byref(n), ctypes.byref(e), ctypes.byref(o), ctypes.byref(h), ctypes.byref(nl), ctypes.byref(nr)))
        self.n_local, self.ext_len, self.own_offset, self.halo_elements, self.nnz_loc, self.nnz_rem = n.value, e.value, o.value, h.value, nl.value, nr.value
        k = ctypes.c_int()
        self.check(self.lib.smm_hip_dist_csr_halo_chunks(self._h, ctypes.byref(k)))
        self.halo_chunks = k.value  # pieces the halo travels in (SMM_HIP_HALO_CHUNKS at create time, agreed by all ranks; 1 = one exchange per SpMV)
        p2p, relays, first, share = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        self.check(self.lib.smm_hip_dist_csr_options(self._h, ctypes.byref(p2p), ctypes.byref(relays), ctypes.byref(first), ctypes.byref(share)))
        # how the halo and the scalars travel (smm_hip.h, smm_hip_dist_csr_options): agreed by all ranks when the matrix was created
        # p2p: the halo travels peer to peer; p2p_scalars: the dot products go through the per-rank slots (also true in the hybrid, where the halo stays
        # with the communicator's send / receive)
        self.options = {"p2p": p2p.value == 1, "p2p_scalars": p2p.value != 0, "relays": relays.value, "halo_first": bool(first.value), "direct_share": share.value}
        self._M = None

    def matvec_forms(self):
        """(SpMVs with a halo run as ONE launch, as TWO launches) so far -- csrc/smm_spmv_split.hip"""
        one, two = ctypes.c_longlong(), ctypes.c_longlong()
        self.check(self.lib.smm_hip_dist_csr_matvec_forms(self._h, ctypes.byref(one), ctypes.byref(two)))
        return one.value, two.value

    def thin_remote(self):
        """(rows listed as holding a remote entry -- 0: the remote block is not thin --, SpMVs whose second half ran over those rows only)"""
        rows, count = ctypes.c_int(), ctypes.c_longlong()
        self.check(self.lib.smm_hip_dist_csr_thin_remote(self._h, ctypes.byref(rows), ctypes.byref(count)))
        return rows.value, count.value

    def cg_fused(self):
        """SpMVs of cg() that formed the next direction themselves (MarchFuse in the row-partitioned loop)"""
        count = ctypes.c_longlong()
        self.check(self.lib.smm_hip_dist_csr_cg_fused(self._h, ctypes.byref(count)))
        return count.value

    def split_wait_ms(self, reset=True):
        """total milliseconds workgroup 0 of the one-launch SpMVs waited for the halo's word after its local half (device clock)"""
        ms = ctypes.c_double()
        self.check(self.lib.smm_hip_dist_csr_split_wait(self._h, ctypes.byref(ms), 1 if reset else 0))
        return ms.value

    def local_blocks(self):
        a, r = ctypes.c_void_p(), ctypes.c_void_p()
        self.check(self.lib.smm_hip_dist_csr_local_block(self._h, ctypes.byref(a), ctypes.byref(r)))
        return _BlockView(a, self, self.np_dtype), _BlockView(r, self, self.np_dtype)

    def set_precond(self, kind):
        """block-Jacobi by rank: JACOBI / ILU0 / SGS of this rank's diagonal block (None: no preconditioner)"""
        if self._M is not None:
            self._M.close()
        self._M = None
        if kind is not None and int(kind) != 0:
            self._M = self.host.Preconditioner(self.local_blocks()[0], kind)

    def spmv(self, op, d_lhs, d_x, d_out, stream=None):
        d = self.host._dptr
        self.check(getattr(self.lib, f"smm_hip_dist_spmv_dev_{self.suf}")(self._h, int(op), d(d_lhs), d(d_x), d(d_out), d(stream)))

    def bicgstab(self, d_b, d_x, max_iterations, eps, stream=None):
        d = self.host._dptr
        st, it = ctypes.c_int(), ctypes.c_int()
        res = (ctypes.c_float if self.suf == "f32" else ctypes.c_double)()
        self.check(getattr(self.lib, f"smm_hip_dist_bicgstab_dev_{self.suf}")(self._h, d(d_b), d(d_x), int(max_iterations), self.np_dtype.type(eps),
                                                                             self.host._mh(self._M), d(stream), ctypes.byref(st), ctypes.byref(it), ctypes.byref(res)))
        return st.value, it.value, res.value

    def cg(self, d_b, d_x0, d_x, max_iterations, eps, stream=None):
        d = self.host._dptr
        st, it = ctypes.c_int(), ctypes.c_int()
        res = (ctypes.c_float if self.suf == "f32" else ctypes.c_double)()
        self.check(getattr(self.lib, f"smm_hip_dist_cg_dev_{self.suf}")(self._h, d(d_b), d(d_x0), d(d_x), int(max_iterations), self.np_dtype.type(eps), d(stream),
                                                                       ctypes.byref(st), ctypes.byref(it), ctypes.byref(res)))
        return st.value, it.value, res.value

    def close(self):
        if self._M is not None:
            self._M.close()
            self._M = None
        if self._h:
            self.lib.smm_hip_dist_csr_destroy(self._h)
            self._h = ctypes.c_void_p()


# ---------------------------------------------------------------------------------------------------------------------
# bench.py leg for N > 1: the same 10M-row banded matrix, rows split over the ranks (strong scaling)
# ---------------------------------------------------------------------------------------------------------------------
def bench_bicgstab(args, rank, world, dev, np_dtype, t_dtype):
    """One rank of `bench.py --gpus N`: the 10M-row banded matrix range-partitioned by nonzeros (strong scaling).  Driver:
    "native" (default) -- the loop of csrc/smm_dist.hip behind the C ABI, RCCL communicator created from a unique id carried by
    torch.distributed; "python" -- DistBiCGStab above over torch.distributed collectives (the stage-wise kernels)."""
    import time

    import torch
    import torch.distributed as dist

    from . import host

    n = args.rows
    driver = getattr(args, "dist_driver", "native")
    staged = dist.get_backend() == "gloo"
    stream = torch.cuda.current_stream().cuda_stream
    row_start = lambda i: host_gen_row_start(args, i)  # noqa: E731
    bounds = partition_rows_by_nnz(row_start, n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    nnz_local = row_start(hi) - row_start(lo)
    d_start = torch.empty(hi - lo + 1, dtype=torch.int32, device=dev)
    d_pos = torch.empty(nnz_local, dtype=torch.int32, device=dev)
    d_val = torch.empty(nnz_local, dtype=t_dtype, device=dev)
    host.gen_banded_rows_dev(n, args.band_k, args.seed, args.max_offset, args.diag_shift, lo, hi, d_start, d_pos, d_val, np_dtype, stream)
    torch.cuda.synchronize()
    # b = A x_true, x_true uniform in [0.5, 1.5) generated per rank from (seed, rank)
    x_true = torch.rand(hi - lo, dtype=t_dtype, device=dev, generator=torch.Generator(device=dev).manual_seed(args.seed + rank)) + 0.5
    b = torch.empty(hi - lo, dtype=t_dtype, device=dev)
    x = torch.zeros(hi - lo, dtype=t_dtype, device=dev)
    if driver == "native":
        try:
            comm = NativeComm.gloo(dist) if staged else NativeComm.rccl(dist)
        except RuntimeError as e:  # raised on EVERY rank (see NativeComm.rccl): fall back together, loudly
            import sys

            print(f"rank {rank}: native communicator unavailable ({e}); using the Python driver over torch.distributed", file=sys.stderr)
            driver = "python"
    if driver == "native":
        # every collective the loop uses, once, with bounded waits (SMM_HIP_COMM_TIMEOUT_S): a rank whose peers never join gets
        # SMM_HIP_ERR_COMM here instead of hanging, and the process ends with a non-zero status (a fresh exit, never a re-exec)
        try:
            comm.selftest()
        except Exception as e:  # noqa: BLE001
            raise SystemExit(f"rank {rank}: the communicator's self-test failed: {e}")
        A = NativeDistMatrix(comm, n, bounds, d_start, d_pos, d_val, np_dtype)
        del d_pos, d_val
        A.spmv(OP_ASSIGN, None, x_true, b, stream)
        halo, ext_len = A.halo_elements, A.ext_len
        one_launch = A.nnz_rem == 0 and A.halo_elements == 0
        comm_info = comm.info()

        def solve(it):
            return A.bicgstab(b, x, it, 0.0, stream)
    else:
        solver = build_hip_solver(torch, dist, d_start, d_pos, d_val, bounds, n, np_dtype, dev)
        del d_pos, d_val
        ops = solver.ops
        ops.copy_into_ext(ops.x_ext, x_true)
        solver._matvec(ops.x_ext, ops.own(ops.x_ext), b, OP_ASSIGN, None, 0, None)
        halo, ext_len = solver.halo_elements, ops.x_ext.numel()
        one_launch = getattr(ops, "rem_empty", False) and not solver.sends and not solver.recvs
        comm_info = {"rank": rank, "world": world, "kind": "torch.distributed/" + dist.get_backend(), "rccl_ranks": 0}

        def solve(it):
            return solver.solve(b, x, it, 0.0, check_every=1 << 30)

    def run(total):
        done, last = 0, None
        while done < total:
            it = min(args.iters_per_solve, total - done)
            x.zero_()
            status, iters, resnorm = solve(it)
            if iters != it or not np.isfinite(resnorm) or resnorm <= 0:
                raise SystemExit(f"rank {rank}: BiCGStab ran {iters} of {it} iterations (resnorm {resnorm}): the timed region is invalid")
            done += iters
            last = resnorm
        return done, last

    def failed(e):
        # a communication failure in the N > 1 leg (a bounded wait expired, a collective failed): this rank leaves with a status of its own, at once
        # and without tearing anything down (the peers run into THEIR bounded waits); never a re-exec of a process that has touched the GPU
        import os
        import sys

        print(f"rank {rank}: the row-partitioned solve failed: {e}", file=sys.stderr, flush=True)
        os._exit(3)

    try:
        if args.warmup > 0:
            run(args.warmup)
    except Exception as e:  # noqa: BLE001
        failed(e)
    # THE timed region (`value`, `ms_per_step`): K iterations between two barriers, no instrumentation inside.  The live SpMV timing -- HIP
    # events around every SpMV launch -- is not free in this loop: measured on one rank's share of the benchmark matrix (1.25 M rows,
    # profiles/r05/rank_loop_gaps.txt) every instrumented launch has ~6 us of idle stream on either side, 316 us per iteration against
    # 295 without the events (7 %; the share grows with the number of ranks).  The events therefore run in a SECOND pass of the same K
    # iterations right behind the timed one; `roofline` and `exposed_comm_ms` come from that pass, `ms_per_step_instrumented` says what it cost.
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        iters, resnorm = run(args.steps)
    except Exception as e:  # noqa: BLE001
        failed(e)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    host.profile_enable(True)
    host.profile_read(reset=True)
    host.profile_read_waits(reset=True)
    dist.barrier()
    torch.cuda.synchronize()
    forms_before = A.matvec_forms() if driver == "native" else (0, 0)
    if driver == "native":
        A.split_wait_ms(reset=True)
    t1 = time.perf_counter()
    try:
        iters2, _ = run(args.steps)
    except Exception as e:  # noqa: BLE001
        failed(e)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed_instrumented = time.perf_counter() - t1
    spmv_ms, spmv_launches = host.profile_read(reset=True)
    # what the halo exchanges cost beyond the local block that ran beside them (events on the caller's and the communicator's stream;
    # zero pairs: the staged / single-rank communicators exchange on the caller's own stream)
    exposed_ms, exchanges = host.profile_read_waits(reset=True)
    # the one-launch SpMV waits INSIDE the kernel (no event pair on the streams): what its workgroup 0 waited for the halo's word in this pass
    split_wait_ms = A.split_wait_ms(reset=True) if driver == "native" else 0.0
    host.profile_enable(False)
    iters_measured = max(iters2, 1)
    err = ((x - x_true).abs() / x_true).max().reshape(1)
    if staged:
        err = err.cpu()
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    nnz_total = row_start(n)
    s_bytes = np.dtype(np_dtype).itemsize
    # one matvec = the A_loc launch + the A_rem launch; algorithmic bytes of this rank's slice (SURVEY.md section 8d formula)
    b_local = nnz_local * (s_bytes + 4) + (hi - lo + 1) * 4 + ext_len * s_bytes + (hi - lo) * s_bytes
    # which family served the two local blocks: AUTO may have moved a large block to the index-free PATTERN family (its first SpMV
    # verifies every entry); the roofline object is priced with CSR bytes only when the CSR kernels ran -- otherwise the bytes really
    # moved (no positions[], 8 bytes of mask per row) are reported beside it and `frac` uses THEM
    families, b_true, kernels = None, b_local, ["spmvTileKernel"]
    if driver == "native":
        blk_loc, blk_rem = A.local_blocks()
        families = {"A_loc": blk_loc.get_kernel() + blk_loc.pattern_info()[:1], "A_rem": blk_rem.get_kernel() + blk_rem.pattern_info()[:1]}
        kernels = [blk_loc.kernel_desc()[0]] + ([] if one_launch else [blk_rem.kernel_desc()[0]])  # the library's own answer (smm_hip_csr_kernel_desc)
        b_true = ext_len * s_bytes + (hi - lo) * s_bytes
        for name, nnz_blk in (("A_loc", A.nnz_loc), ("A_rem", A.nnz_rem)):
            fam, lanes, enc = families[name]
            if fam == 3 and enc == 3 and lanes == 1:  # constant diagonals: the row masks only
                b_true += (hi - lo) * 8
            elif fam == 3 and enc == 2:  # a 16-bit code per entry
                b_true += nnz_blk * (s_bytes + 2) + (hi - lo + 1) * 4
            elif fam == 3:  # row masks + values
                b_true += nnz_blk * s_bytes + (hi - lo) * 8 + (hi - lo + 1) * 4
            else:
                b_true += nnz_blk * (s_bytes + 4) + (hi - lo + 1) * 4
    per_matvec = 1 if one_launch else 2
    matvecs = max(spmv_launches // per_matvec, 1)
    split_forms = None
    if driver == "native" and not one_launch:
        # SpMVs with a halo run as ONE launch (csrc/smm_spmv_split.hip: local half, wait for the exchange's word, remote half) or as two
        forms_after = A.matvec_forms()
        d_one, d_two = forms_after[0] - forms_before[0], forms_after[1] - forms_before[1]
        split_forms = {"one_launch": d_one, "two_launches": d_two}
        if d_one + d_two > 0:
            matvecs = d_one + d_two
            per_matvec = 1 if d_two == 0 else 2
        if d_one > 0 and d_two == 0:
            kernels = ["spmvPatternSplitKernel"]
    matvec_s = spmv_ms * 1e-3 / matvecs
    achieved = b_true / matvec_s / 1e9
    return {
        # the same definition as the one-GPU line's `roofline`: the kernels the timed region runs, priced with the bytes THEIR layouts move
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                     "kernel": " + ".join(kernels) + (" (one matvec of rank 0 = ONE launch over A_loc and A_rem)" if kernels == ["spmvPatternSplitKernel"] else
                                                    " (one matvec of rank 0 = its A_loc launch" + (" + its A_rem launch)" if per_matvec == 2 else ")")),
                     "algorithmic_bytes_per_launch": b_true, "csr_bytes_per_launch": b_local, "avg_launch_ms": matvec_s * 1e3,
                     "launches": matvecs, "matvec_forms": split_forms, "families": families,
                     "measured_in": "a second pass of the same K iterations right behind the timed region (the events cost this loop ~12 us per SpMV launch: kept out of `value`)",  # (family, lanes per row, PATTERN encoding) of each block
                     "note": "bytes = what the kernels that ran really move per matvec of rank 0 (a block AUTO moved to the PATTERN family has no positions[]); "
                             "csr_bytes_per_launch is the SURVEY 8d formula; the CSR kernel's own fraction is the one-GPU line's roofline_csr"},
        "elapsed": elapsed,
        "iters": iters,
        "ms_per_step_instrumented": elapsed_instrumented / iters_measured * 1e3,
        "nnz": nnz_total,
        "resnorm": float(resnorm),
        "max_rel_err_vs_x_true": float(err.item()),
        "rccl_ranks": comm_info["rccl_ranks"],  # the communicator's size as RCCL reports it (ncclCommCount); 0: not an RCCL communicator
        "distributed": {"driver": driver, "comm": comm_info["kind"], "comm_ranks": comm_info["world"],
                        "kernels_per_iteration": 8 if driver == "native" else 13, "allreduces_per_iteration": 3, "halo_exchanges_per_iteration": 2},
        # the first thing to read in a multi-GPU line that scales worse than hoped: milliseconds per BiCGStab iteration (two exchanges) that
        # rank 0's A_rem waited for its halo AFTER A_loc had ended -- the exchange's share that no compute covered
        "exposed_comm_ms": (exposed_ms + split_wait_ms) / iters_measured,
        "exposed_comm": {"total_ms": exposed_ms + split_wait_ms, "one_launch_wait_ms": split_wait_ms, "exchanges": exchanges, "ms_per_exchange": exposed_ms / max(exchanges, 1),
                         "note": "rank 0; two launches: events, end of A_loc on the solver's stream -> end of the halo exchange on the communicator's stream, clipped at 0; "
                                 "one launch: what workgroup 0 of the SpMV waited for the exchange's word after its local half (device clock)"},
        "halo_chunks": A.halo_chunks if driver == "native" else 1,
        "dist_options": A.options if driver == "native" else None,
        "per_rank": {"rows": hi - lo, "nnz": nnz_local, "halo_elements": halo,
                     "spmv_launch_ms_rank0": spmv_ms / max(spmv_launches, 1), "spmv_launches_rank0": spmv_launches},
    }


def host_gen_row_start(args, row):
    from . import _lib

    return int(_lib.load().smm_hip_gen_banded_row_start(int(args.rows), int(args.band_k), int(args.seed), int(args.max_offset), int(row)))
