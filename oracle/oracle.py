"""ctypes wrapper of the CPU oracle (oracle/_build/libsmm_oracle.so) and, where it was built, of the real
reference behind oracle/_ref/libsmm_ref.so.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
Nothing under sparse_matrix_math_amd/ imports this module.  See oracle/smm_oracle.h for the parity status of
each function (pinned vs unpinned).
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "_build", "libsmm_oracle.so")
ORACLE_FMA_LIB = os.path.join(_HERE, "_build", "libsmm_oracle_fma.so")
REF_LIB = os.path.join(_HERE, "_ref", "libsmm_ref.so")

PRECOND_NONE, PRECOND_JACOBI, PRECOND_ILU0, PRECOND_SGS = 0, 1, 2, 3
PRECOND_BLOCK_ILU0, PRECOND_BLOCK_SGS = 5, 6
OP_ASSIGN, OP_ADD, OP_SUB = 0, 1, 2

_SUF = {np.dtype(np.float32): ("f32", c_float), np.dtype(np.float64): ("f64", c_double)}


def build(ref=True):
    """gcc-compile the restatement; compile the reference harness too when /root/reference is mounted."""
    subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
    if ref and os.path.exists("/root/reference/include/sparse_matrix_math.h"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else c_void_p(0)


def block_diagonal_part(csr, bounds):
    """(A with every entry that couples two row blocks removed, mask of the kept entries); bounds: nblocks + 1 row numbers"""
    start, pos, val = csr
    n = len(start) - 1
    bounds = np.asarray(bounds)
    rowof = np.repeat(np.arange(n), np.diff(start))
    keep = (np.searchsorted(bounds, rowof, side="right") - 1) == (np.searchsorted(bounds, pos, side="right") - 1)
    s2 = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rowof[keep], minlength=n), out=s2[1:])
    return (s2, pos[keep].copy(), val[keep].copy()), keep


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class Oracle:
    """The C restatement.  fma=True loads the SMM_WITH_STD_FMA flavour."""

    def __init__(self, fma=False):
        path = ORACLE_FMA_LIB if fma else ORACLE_LIB
        if not os.path.exists(path):
            build(ref=False)
        self.path = path
        self.lib = ctypes.CDLL(path)
        self.lib.smm_oracle_omp_max_threads.restype = c_int
        for suf, ct in (("f32", c_float), ("f64", c_double)):
            for name in ("dot", "nrm2sq", "dot_tbbshape", "omp_dot"):
                getattr(self.lib, f"smm_oracle_{name}_{suf}").restype = ct

    def _f(self, name, dtype):
        suf, ct = _SUF[np.dtype(dtype)]
        return getattr(self.lib, f"smm_oracle_{name}_{suf}"), ct

    def spmv(self, csr, op, lhs, x, omp=False):
        start, pos, val = csr
        rows = len(start) - 1
        out = np.zeros(rows, dtype=val.dtype)
        fn, _ = self._f("omp_spmv" if omp else "spmv", val.dtype)
        fn(c_int(rows), _p(start), _p(pos), _p(val), c_int(op), _p(lhs), _p(x), _p(out))
        return out

    def spmv_inplace(self, csr, op, lhs_out, x):
        """out aliases lhs (the in-place sub-cases of test/cpp/csr.cpp:295-300)"""
        start, pos, val = csr
        fn, _ = self._f("spmv", val.dtype)
        fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), c_int(op), _p(lhs_out), _p(x), _p(lhs_out))
        return lhs_out

    def dot(self, a, b, kind="dot"):
        fn, ct = self._f(kind, a.dtype)
        return a.dtype.type(fn(c_int(a.size), _p(a), _p(b)))

    def cg(self, csr, b, x0, maxit, eps, omp=False):
        start, pos, val = csr
        rows = len(start) - 1
        x = x0.copy()
        fn, ct = self._f("omp_cg" if omp else "cg", val.dtype)
        it, res = c_int(), ct()
        st = fn(c_int(rows), _p(start), _p(pos), _p(val), _p(b), _p(x0), _p(x), c_int(maxit), ct(eps), ctypes.byref(it), ctypes.byref(res))
        return st, x, it.value, res.value

    def bicgstab(self, csr, b, x0, maxit, eps, precond=PRECOND_NONE, precond_values=None, omp=False):
        start, pos, val = csr
        rows = len(start) - 1
        x = x0.copy()
        bb = b.copy()
        it = c_int()
        if omp:
            fn, ct = self._f("omp_bicgstab", val.dtype)
            res = ct()
            st = fn(c_int(rows), _p(start), _p(pos), _p(val), _p(bb), _p(x), c_int(maxit), ct(eps), ctypes.byref(it), ctypes.byref(res))
        else:
            fn, ct = self._f("bicgstab", val.dtype)
            res = ct()
            st = fn(c_int(rows), _p(start), _p(pos), _p(val), _p(bb), _p(x), c_int(maxit), ct(eps), c_int(precond), _p(precond_values),
                    ctypes.byref(it), ctypes.byref(res))
        return st, x, it.value, res.value

    def bicgsymmetric(self, csr, b, x0, maxit, eps):
        start, pos, val = csr
        x = x0.copy()
        bb = b.copy()
        fn, ct = self._f("bicgsymmetric", val.dtype)
        it = c_int()
        st = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(bb), _p(x), c_int(maxit), ct(eps), ctypes.byref(it))
        return st, x, it.value

    def sgs_apply(self, csr, rhs):
        start, pos, val = csr
        x = np.zeros_like(rhs)
        fn, _ = self._f("sgs_apply", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(rhs), _p(x))
        return err, x

    def jacobi_setup(self, csr):
        start, pos, val = csr
        diag = np.zeros(len(start) - 1, dtype=val.dtype)
        fn, _ = self._f("jacobi_setup", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(diag))
        return err, diag

    def jacobi_apply(self, diag, rhs):
        x = np.zeros_like(rhs)
        fn, _ = self._f("jacobi_apply", rhs.dtype)
        fn(c_int(rhs.size), _p(diag), _p(rhs), _p(x))
        return x

    def ilu0_factorize(self, csr):
        start, pos, val = csr
        lu = np.zeros_like(val)
        fn, _ = self._f("ilu0_factorize", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(lu))
        return err, lu

    def ilu0_apply(self, csr, lu, rhs):
        start, pos, val = csr
        x = np.zeros_like(rhs)
        fn, _ = self._f("ilu0_apply", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(lu), _p(rhs), _p(x))
        return err, x

    def ic0_factorize(self, csr):
        start, pos, val = csr
        ic = np.zeros_like(val)
        fn, _ = self._f("ic0_factorize", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(ic))
        return err, ic

    def ic0_apply(self, csr, ic, rhs):
        start, pos, val = csr
        x = np.zeros_like(rhs)
        fn, _ = self._f("ic0_apply", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(ic), _p(rhs), _p(x))
        return err, x

    def pcg_ic0(self, csr, ic, b, x0, maxit, eps):
        start, pos, val = csr
        x = x0.copy()
        fn, ct = self._f("pcg_ic0", val.dtype)
        it, res = c_int(), ct()
        st = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(ic), _p(b), _p(x0), _p(x), c_int(maxit), ct(eps), ctypes.byref(it),
                ctypes.byref(res))
        return st, x, it.value, res.value

    # ---- block preconditioners: the global algorithm on the block-diagonal part of A (bounds: nblocks + 1 row numbers) ----
    def block_ilu0_factorize(self, csr, bounds):
        start, pos, val = csr
        bounds = _c(bounds, np.int32)
        lu = np.zeros_like(val)
        fn, _ = self._f("block_ilu0_factorize", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), c_int(len(bounds) - 1), _p(bounds), _p(lu))
        return err, lu

    def block_ilu0_apply(self, csr, bounds, lu, rhs):
        start, pos, val = csr
        bounds = _c(bounds, np.int32)
        x = np.zeros_like(rhs)
        fn, _ = self._f("block_ilu0_apply", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(lu), c_int(len(bounds) - 1), _p(bounds), _p(rhs), _p(x))
        return err, x

    def block_sgs_apply(self, csr, bounds, rhs):
        start, pos, val = csr
        bounds = _c(bounds, np.int32)
        x = np.zeros_like(rhs)
        fn, _ = self._f("block_sgs_apply", val.dtype)
        err = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), c_int(len(bounds) - 1), _p(bounds), _p(rhs), _p(x))
        return err, x

    def block_level_cut(self, csr, bounds, cap):
        """(mask of the stored entries the level-capped block preconditioners keep, deepest level + 1); cap <= 0: no cap"""
        start, pos, _ = csr
        bounds = _c(bounds, np.int32)
        keep = np.zeros(len(pos), dtype=np.uint8)
        self.lib.smm_oracle_block_level_cut.restype = c_int
        deepest = self.lib.smm_oracle_block_level_cut(c_int(len(start) - 1), _p(start), _p(pos), c_int(len(bounds) - 1), _p(bounds), c_int(int(cap)), _p(keep))
        return keep.astype(bool), deepest

    def bicgstab_block(self, csr, b, x0, maxit, eps, precond, bounds, precond_values=None):
        start, pos, val = csr
        bounds = _c(bounds, np.int32)
        x = x0.copy()
        bb = b.copy()
        it = c_int()
        fn, ct = self._f("bicgstab_block", val.dtype)
        res = ct()
        st = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(bb), _p(x), c_int(maxit), ct(eps), c_int(precond), _p(precond_values),
                c_int(len(bounds) - 1), _p(bounds), ctypes.byref(it), ctypes.byref(res))
        return st, x, it.value, res.value

    def bicgstab_block_of(self, csr, mcsr, b, x0, maxit, eps, precond, bounds, precond_values=None):
        """bicgstab_block with the preconditioner built from another matrix (mcsr; precond_values on ITS pattern)"""
        start, pos, val = csr
        ms, mp, mv = mcsr
        bounds = _c(bounds, np.int32)
        x = x0.copy()
        bb = b.copy()
        it = c_int()
        fn, ct = self._f("bicgstab_block_of", val.dtype)
        res = ct()
        st = fn(c_int(len(start) - 1), _p(start), _p(pos), _p(val), _p(ms), _p(mp), _p(mv), _p(bb), _p(x), c_int(maxit), ct(eps), c_int(precond),
                _p(precond_values), c_int(len(bounds) - 1), _p(bounds), ctypes.byref(it), ctypes.byref(res))
        return st, x, it.value, res.value

    def level_cut_matrix(self, csr, bounds, cap):
        """(the matrix the level-cut block preconditioners are built from: A's in-block entries that the cut keeps; mask of those
        entries over A's pattern; deepest level + 1).  cap 0 = the block-diagonal part."""
        start, pos, val = csr
        keep, deepest = self.block_level_cut(csr, bounds, cap)
        n = len(start) - 1
        rowof = np.repeat(np.arange(n), np.diff(start))
        s2 = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(rowof[keep], minlength=n), out=s2[1:])
        return (s2, pos[keep].copy(), val[keep].copy()), keep, deepest

    def set_spmv_form(self, rows_in_lock_step):
        """1, 2 or 4 rows walked in lock step by the OpenMP port's SpMV (same bits every way; tools/cpu_port_ab.py)"""
        self.lib.smm_oracle_omp_set_spmv_form(int(rows_in_lock_step))

    def omp_threads(self):
        return self.lib.smm_oracle_omp_max_threads()

    def set_threads(self, n):
        self.lib.smm_oracle_omp_set_threads(c_int(n))


class Reference:
    """The real reference header (build container only).  `available()` is False on the GPU box."""

    @staticmethod
    def available():
        return os.path.exists(REF_LIB)

    def __init__(self):
        self.lib = ctypes.CDLL(REF_LIB)
        for suf, ct in (("f32", c_float), ("f64", c_double)):
            getattr(self.lib, f"ref_csr_create_{suf}").restype = c_void_p
            getattr(self.lib, f"ref_dot_{suf}").restype = ct
        self.lib.ref_load_mtx_f64.argtypes = [c_char_p, POINTER(c_int), POINTER(c_int), c_int, c_void_p, c_void_p, c_void_p]

    def _f(self, name, dtype):
        suf, ct = _SUF[np.dtype(dtype)]
        return getattr(self.lib, f"ref_{name}_{suf}"), ct

    class _Csr:
        def __init__(self, ref, csr, cols=None):
            start, pos, val = csr
            self.ref, self.dtype, self.rows = ref, val.dtype, len(start) - 1
            fn, _ = ref._f("csr_create", val.dtype)
            self.h = c_void_p(fn(c_int(self.rows), c_int(self.rows if cols is None else cols), _p(start), _p(pos), _p(val)))

        def __enter__(self):
            return self

        def __exit__(self, *a):
            fn, _ = self.ref._f("csr_destroy", self.dtype)
            fn(self.h)

    def csr(self, csr, cols=None):
        return Reference._Csr(self, csr, cols)

    def spmv(self, m, op, lhs, x):
        out = np.zeros(m.rows, dtype=m.dtype)
        fn, _ = self._f("spmv", m.dtype)
        fn(m.h, c_int(op), _p(lhs), _p(x), _p(out))
        return out

    def spmv_inplace(self, m, op, lhs_out, x):
        fn, _ = self._f("spmv", m.dtype)
        fn(m.h, c_int(op), _p(lhs_out), _p(x), _p(lhs_out))
        return lhs_out

    def dot(self, a, b):
        fn, ct = self._f("dot", a.dtype)
        return a.dtype.type(fn(c_int(a.size), _p(a), _p(b)))

    def cg(self, m, b, x0, maxit, eps):
        x = x0.copy()
        fn, ct = self._f("cg", m.dtype)
        st = fn(m.h, _p(b), _p(x0), _p(x), c_int(maxit), ct(eps))
        return st, x

    def cg_inplace(self, m, b, x, maxit, eps):
        fn, ct = self._f("cg", m.dtype)
        st = fn(m.h, _p(b), _p(x), _p(x), c_int(maxit), ct(eps))
        return st, x

    def bicgstab(self, m, b, x0, maxit, eps, precond=PRECOND_NONE, diag=None):
        x = x0.copy()
        bb = b.copy()
        fn, ct = self._f("bicgstab", m.dtype)
        st = fn(m.h, _p(bb), _p(x), c_int(maxit), ct(eps), c_int(precond), _p(diag))
        return st, x

    def bicgsymmetric(self, m, b, x0, maxit, eps):
        x = x0.copy()
        bb = b.copy()
        fn, ct = self._f("bicgsymmetric", m.dtype)
        st = fn(m.h, _p(bb), _p(x), c_int(maxit), ct(eps))
        return st, x

    def bicgstab_sgs_of(self, m, m_precond, b, x0, maxit, eps):
        """BiCGStab<SGSPreconditioner> on m with the SGS preconditioner of ANOTHER matrix (its block-diagonal part)"""
        x = x0.copy()
        bb = b.copy()
        fn, ct = self._f("bicgstab_sgs_of", m.dtype)
        st = fn(m.h, m_precond.h, _p(bb), _p(x), c_int(maxit), ct(eps))
        return st, x

    def sgs_apply(self, m, rhs):
        x = np.zeros_like(rhs)
        fn, _ = self._f("sgs_apply", m.dtype)
        err = fn(m.h, _p(rhs), _p(x))
        return err, x

    def ic0(self, m, nnz, rhs):
        ic = np.zeros(nnz, dtype=m.dtype)
        x = np.zeros_like(rhs)
        fn, _ = self._f("ic0", m.dtype)
        err = fn(m.h, _p(ic), _p(rhs), _p(x))
        return err, ic, x

    def pcg_ic0(self, m, b, x0, maxit, eps):
        x = x0.copy()
        fn, ct = self._f("pcg_ic0", m.dtype)
        st = fn(m.h, _p(b), _p(x0), _p(x), c_int(maxit), ct(eps))
        return st, x

    def load_mtx(self, path, cap=1 << 22):
        rows, cols = c_int(), c_int()
        start = np.zeros(cap, dtype=np.int32)
        pos = np.zeros(cap, dtype=np.int32)
        val = np.zeros(cap, dtype=np.float64)
        nnz = self.lib.ref_load_mtx_f64(path.encode(), ctypes.byref(rows), ctypes.byref(cols), c_int(cap), _p(start), _p(pos), _p(val))
        if nnz < 0:
            raise RuntimeError(f"reference loader failed on {path}: {nnz}")
        return rows.value, cols.value, start[: rows.value + 1].copy(), pos[:nnz].copy(), val[:nnz].copy()
