"""Host-side mirror of the reference's public API for the hot path, on top of the C ABI (include/smm_hip.h).

Names, argument order and error behaviour follow include/sparse_matrix_math.h of vasil-pashov/sparse_matrix_math:
CSRMatrix.rMult / rMultAdd / rMultSub (ref:1501-1515), getPreconditioner (ref:1643-1651), ConjugateGradient
(ref:2316-2398, IC0 overload ref:2414-2505), BiCGStab (ref:2191-2303), BiCGSymmetric (ref:2021-2102), SolverStatus
(ref:2010-2014), SolverPreconditioner (ref:1002-1006).  Vectors are numpy arrays in host memory, like the
reference's raw T* arguments; the `*_dev` helpers take device pointers (ints or objects with .data_ptr()) for
callers that keep their data in HBM (bench.py, the multi-GPU driver).

Everything here runs on the GPU through libsmm_hip.so.  There is no CPU implementation in this package.
"""
import ctypes
import enum

import numpy as np

from . import _lib
from ._lib import check


class SolverStatus(enum.IntEnum):  # ref:2010-2014
    SUCCESS = 0
    DIVERGED = 1
    MAX_ITERATIONS_REACHED = 2


class SolverPreconditioner(enum.IntEnum):
    """ref:1002-1006 has NONE, SYMMETRIC_GAUS_SEIDEL (sic) and ILU0; JACOBI, IC0 and the BLOCK_ forms (ILU0 / SGS of the
    block-diagonal part of A, one wavefront per block) are additions.  Values are the SMM_PRECOND_* codes of the C ABI."""
    NONE = 0
    JACOBI = 1
    ILU0 = 2
    SYMMETRIC_GAUS_SEIDEL = 3
    IC0 = 4
    BLOCK_ILU0 = 5
    BLOCK_SGS = 6


OP_ASSIGN, OP_ADD, OP_SUB = 0, 1, 2
SPMV_AUTO, SPMV_VECTOR, SPMV_STREAM, SPMV_PATTERN = 0, 1, 2, 3
SWEEP_AUTO, SWEEP_LEVELS, SWEEP_SYNCFREE, SWEEP_SYNCFREE_XCD = 0, 1, 2, 3

_SUFFIX = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}
_CT = {"f32": ctypes.c_float, "f64": ctypes.c_double}


def _suffix(dtype):
    try:
        return _SUFFIX[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"only float32 and float64 are supported (the reference's float/double), got {dtype}")


def _fn(base, suf):
    return getattr(_lib.load(), f"{base}_{suf}")


def _host(a, dtype, name, n=None, writable=False):
    if not isinstance(a, np.ndarray) or a.dtype != np.dtype(dtype) or not a.flags.c_contiguous:
        raise TypeError(f"{name} must be a C-contiguous numpy array of {np.dtype(dtype)}")
    if writable and not a.flags.writeable:
        raise TypeError(f"{name} must be writable")
    if n is not None and a.size < n:
        raise ValueError(f"{name} has {a.size} elements, needs {n}")
    return a.ctypes.data_as(ctypes.c_void_p)


def _dptr(t):
    """device pointer from an int, None, or anything with .data_ptr() (torch tensors)"""
    if t is None:
        return ctypes.c_void_p(0)
    if hasattr(t, "data_ptr"):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(int(t))


def init(device=0):
    check(_lib.load().smm_hip_init(int(device)))


def device_info():
    name = ctypes.create_string_buffer(256)
    cus = ctypes.c_int()
    mem = ctypes.c_size_t()
    check(_lib.load().smm_hip_device_info(name, 256, ctypes.byref(cus), ctypes.byref(mem)))
    return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": mem.value}


def partials_count():
    return _lib.load().smm_hip_partials_count()


def finish_len():
    return _lib.load().smm_hip_finish_len()


def finish_totals_offset():
    return _lib.load().smm_hip_finish_totals_offset()


def uses_std_fma():
    return bool(_lib.load().smm_hip_uses_std_fma())


def synchronize(stream=None):
    check(_lib.load().smm_hip_stream_synchronize(_dptr(stream)))


def profile_enable(on=True):
    check(_lib.load().smm_hip_profile_enable(1 if on else 0))


def profile_read(reset=True):
    """(summed SpMV kernel milliseconds, launches) measured with HIP events on the launch stream"""
    ms, n = ctypes.c_double(), ctypes.c_longlong()
    check(_lib.load().smm_hip_profile_read(ctypes.byref(ms), ctypes.byref(n), 1 if reset else 0))
    return ms.value, n.value


def set_march_min_rows(const_diagonals_rows=-1, values_read_rows=-1):
    """from how many rows grid-shaped matrices run the 2.5-D kernels (-1: the default); applies to matrices analysed afterwards"""
    check(_lib.load().smm_hip_set_march_min_rows(int(const_diagonals_rows), int(values_read_rows)))


def profile_read_waits(reset=True):
    """(exposed ms, exchanges): what the halo exchanges of the row-partitioned SpMVs cost BEYOND the local block that ran beside them"""
    ms, n = ctypes.c_double(), ctypes.c_longlong()
    check(_lib.load().smm_hip_profile_read_waits(ctypes.byref(ms), ctypes.byref(n), 1 if reset else 0))
    return ms.value, n.value


class Preconditioner:
    """`int apply(const T* rhs, T* x) const` (ref:1173-1235).  Created by CSRMatrix.getPreconditioner."""

    def __init__(self, matrix, kind, block_rows=None, level_cap=None, partition=None):
        self.matrix = matrix  # keeps the matrix alive (the reference holds a const CSRMatrix&)
        self.kind = SolverPreconditioner(kind)
        self._h = ctypes.c_void_p()
        if block_rows is None and level_cap is None and partition is None:
            check(_lib.load().smm_hip_precond_create(matrix._h, int(kind), ctypes.byref(self._h)))
        elif level_cap is None and partition is None:  # BLOCK_ILU0 / BLOCK_SGS with a chosen block size
            check(_lib.load().smm_hip_precond_create_block(matrix._h, int(kind), int(block_rows), ctypes.byref(self._h)))
        else:  # ... a chosen level cut (0 = none; None / -1 = the default) and partition (None / 0 = auto, 1 = contiguous rows, 2 = grid bricks)
            check(_lib.load().smm_hip_precond_create_block_ex(matrix._h, int(kind), int(block_rows or 0), -1 if level_cap is None else int(level_cap),
                                                               int(partition or 0), ctypes.byref(self._h)))

    def block_record_bytes(self):
        """BLOCK_ kinds: bytes one apply reads per row besides the vectors: (lower-sweep record, upper-sweep record, row-order entries)"""
        v = [ctypes.c_int() for _ in range(3)]
        check(_lib.load().smm_hip_precond_block_record_bytes(self._h, *[ctypes.byref(c) for c in v]))
        return tuple(c.value for c in v)

    def block_rows(self):
        """BLOCK_ kinds: (the rows block by block -- order[bounds[b] : bounds[b+1]] are block b's rows, the identity for contiguous
        blocks --, the brick's extent along the grid axes or (0, 0, 0))"""
        n = self.matrix.rows
        order = np.zeros(n, dtype=np.int32)
        brick = (ctypes.c_int * 3)()
        check(_lib.load().smm_hip_precond_block_rows(self._h, order.ctypes.data_as(ctypes.c_void_p), n, ctypes.cast(brick, ctypes.c_void_p)))
        return order, tuple(brick)

    def level_cap(self):
        """BLOCK_ kinds: the level cut this handle was built with (0 = none)"""
        c = ctypes.c_int()
        check(_lib.load().smm_hip_precond_block_level_cap(self._h, ctypes.byref(c)))
        return c.value

    def block_bounds(self):
        """BLOCK_ kinds: the nblocks + 1 row numbers at which the rows were cut"""
        n = ctypes.c_int()
        check(_lib.load().smm_hip_precond_block_count(self._h, ctypes.byref(n)))
        out = np.zeros(n.value + 1, dtype=np.int32)
        check(_lib.load().smm_hip_precond_block_bounds(self._h, out.ctypes.data_as(ctypes.c_void_p), out.size))
        return out

    def apply(self, rhs, x):
        suf = self.matrix._suf
        n = self.matrix.rows
        check(_fn("smm_hip_precond_apply", suf)(self._h, _host(rhs, self.matrix.dtype, "rhs", n), _host(x, self.matrix.dtype, "x", n, True)))
        return 0

    def apply_dev(self, d_rhs, d_x, stream=None):
        check(_fn("smm_hip_precond_apply_dev", self.matrix._suf)(self._h, _dptr(d_rhs), _dptr(d_x), _dptr(stream)))

    def apply_spmv(self, v, x):
        """x = M^-1 (A v) (ref:2234-2235); the BLOCK_ kinds form A v inside the apply's launch"""
        suf = self.matrix._suf
        n = self.matrix.rows
        check(_fn("smm_hip_precond_apply_spmv", suf)(self._h, _host(v, self.matrix.dtype, "v", n), _host(x, self.matrix.dtype, "x", n, True)))
        return 0

    def apply_spmv_dev(self, d_v, d_x, stream=None):
        check(_fn("smm_hip_precond_apply_spmv_dev", self.matrix._suf)(self._h, _dptr(d_v), _dptr(d_x), _dptr(stream)))

    def take_error(self, stream=None):
        """synchronises `stream`; raises when a triangular sweep applied on it failed to finish (apply_dev cannot report it)"""
        check(_lib.load().smm_hip_precond_take_error(self._h, _dptr(stream)))

    def values(self):
        """factor values: diag (JACOBI) or the ILU0 / IC0 / BLOCK_ILU0 values on A's pattern"""
        count = self.matrix.rows if self.kind == SolverPreconditioner.JACOBI else self.matrix.nnz
        out = np.empty(count, dtype=self.matrix.dtype)
        check(_fn("smm_hip_precond_values", self.matrix._suf)(self._h, _host(out, self.matrix.dtype, "out"), count))
        return out

    def set_sweep(self, mode):
        """SWEEP_AUTO / SWEEP_LEVELS / SWEEP_SYNCFREE: how the triangular sweeps are launched (same numbers either way)"""
        check(_lib.load().smm_hip_precond_set_sweep(self._h, int(mode)))

    def levels(self):
        kind, lo, up = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(_lib.load().smm_hip_precond_info(self._h, ctypes.byref(kind), ctypes.byref(lo), ctypes.byref(up)))
        return lo.value, up.value

    def close(self):
        if self._h:
            _lib.load().smm_hip_precond_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CSRMatrix:
    """Device-resident CSRMatrix<T> with the reference's layout (ref:1243-1259): values[nnz], positions[nnz]
    (ascending per row), start[rows+1]."""

    def __init__(self, rows, cols, start, positions, values):
        values = np.ascontiguousarray(values)
        self.dtype = values.dtype
        self._suf = _suffix(self.dtype)
        start = np.ascontiguousarray(start, dtype=np.int32)
        positions = np.ascontiguousarray(positions, dtype=np.int32)
        if start.size != rows + 1:
            raise ValueError("start must have rows+1 entries")
        if positions.size < start[-1] or values.size < start[-1]:
            raise ValueError("positions/values shorter than start[rows]")
        self._h = ctypes.c_void_p()
        self._keep = None
        check(_fn("smm_hip_csr_create", self._suf)(int(rows), int(cols), _host(start, np.int32, "start"), _host(positions, np.int32, "positions"),
                                                   _host(values, self.dtype, "values"), ctypes.byref(self._h)))
        self._read_info()

    @classmethod
    def from_device(cls, rows, cols, d_start, d_positions, d_values, dtype):
        """Wrap arrays already in HBM (no copy).  The arrays are kept referenced by the returned object."""
        self = cls.__new__(cls)
        self.dtype = np.dtype(dtype)
        self._suf = _suffix(self.dtype)
        self._h = ctypes.c_void_p()
        self._keep = (d_start, d_positions, d_values)
        check(_fn("smm_hip_csr_create_dev", self._suf)(int(rows), int(cols), _dptr(d_start), _dptr(d_positions), _dptr(d_values), ctypes.byref(self._h)))
        self._read_info()
        return self

    def _read_info(self):
        r, c, n, d, f = (ctypes.c_int() for _ in range(5))
        check(_lib.load().smm_hip_csr_info(self._h, ctypes.byref(r), ctypes.byref(c), ctypes.byref(n), ctypes.byref(d), ctypes.byref(f)))
        self.rows, self.cols, self.nnz, self.first_active_start = r.value, c.value, n.value, f.value

    # reference getters (ref:1351-1364)
    def getDenseRowCount(self):
        return self.rows

    def getDenseColCount(self):
        return self.cols

    def getNonZeroCount(self):
        return self.nnz

    def set_kernel(self, family=SPMV_AUTO, lanes_per_row=0):
        check(_lib.load().smm_hip_csr_set_kernel(self._h, int(family), int(lanes_per_row)))

    def get_kernel(self):
        fam, lanes = ctypes.c_int(), ctypes.c_int()
        check(_lib.load().smm_hip_csr_get_kernel(self._h, ctypes.byref(fam), ctypes.byref(lanes)))
        return fam.value, lanes.value

    def autotune(self):
        check(_lib.load().smm_hip_csr_autotune(self._h))
        return self.get_kernel()

    def _spmv(self, op, lhs, mult, out):
        fn = _fn("smm_hip_spmv", self._suf)
        plhs = _host(lhs, self.dtype, "lhs", self.rows) if op != OP_ASSIGN else ctypes.c_void_p(0)
        check(fn(self._h, op, plhs, _host(mult, self.dtype, "mult", self.cols), _host(out, self.dtype, "out", self.rows, True)))

    def rMult(self, mult, res):  # ref:1501-1505
        self._spmv(OP_ASSIGN, None, mult, res)

    def rMultAdd(self, lhs, mult, out):  # ref:1507-1510
        self._spmv(OP_ADD, lhs, mult, out)

    def rMultSub(self, lhs, mult, out):  # ref:1512-1515
        self._spmv(OP_SUB, lhs, mult, out)

    def spmv_dev(self, op, d_lhs, d_x, d_out, stream=None):
        check(_fn("smm_hip_spmv_dev", self._suf)(self._h, int(op), _dptr(d_lhs), _dptr(d_x), _dptr(d_out), _dptr(stream)))

    def tile_info(self):
        """(tiles, nonzeros per tile, rows per tile, 1 if spmvTileKernel serves the launches) of the STREAM family's tile table"""
        v = [ctypes.c_int() for _ in range(4)]
        check(_lib.load().smm_hip_csr_tile_info(self._h, *[ctypes.byref(c) for c in v]))
        return tuple(c.value for c in v)

    def pattern_info(self):
        """(encoding, distinct offsets) of the PATTERN family for this matrix: encoding 0 none / not analysed, 1 row masks, 2 entry
        codes, 3 row masks + constant diagonals (no values[] read)"""
        enc, k = ctypes.c_int(), ctypes.c_int()
        check(_lib.load().smm_hip_csr_pattern_info(self._h, ctypes.byref(enc), ctypes.byref(k)))
        return enc.value, k.value

    def kernel_desc(self):
        """(kernel name without template arguments, bytes one launch of it moves by ITS data layout) for the next SpMV of this matrix"""
        buf = ctypes.create_string_buffer(64)
        nbytes = ctypes.c_longlong()
        check(_lib.load().smm_hip_csr_kernel_desc(self._h, buf, 64, ctypes.byref(nbytes)))
        return buf.value.decode(), nbytes.value

    def pattern_allow_const(self, allow):
        """False: a matrix with constant diagonals keeps reading values[] (measurements); same bits either way"""
        check(_lib.load().smm_hip_csr_pattern_allow_const(self._h, 1 if allow else 0))

    def spmv_fused_dev(self, op, d_lhs, d_x, d_out, dot_mode, d_w1, d_partials, stream=None, finish=False):
        """SpMV with the dot products of the fresh out[] in its epilogue (dot_mode 1: out.w1; 2: out.out and out.w1).  finish=False:
        d_partials receives 2 x partials_count() per-workgroup sums; finish=True: d_partials is a finishing buffer of finish_len()
        elements (zeroed once) and the totals land at finish_totals_offset() + {0, 1}"""
        name = "smm_hip_spmv_fused_finish_dev" if finish else "smm_hip_spmv_fused_dev"
        check(_fn(name, self._suf)(self._h, int(op), _dptr(d_lhs), _dptr(d_x), _dptr(d_out), int(dot_mode), _dptr(d_w1), _dptr(d_partials), _dptr(stream)))

    def getPreconditioner(self, kind, block_rows=None, level_cap=None, partition=None):  # ref:1643-1651; the keyword arguments: BLOCK_ kinds only (None = default)
        return Preconditioner(self, kind, block_rows, level_cap, partition)

    def close(self):
        if self._h:
            _lib.load().smm_hip_csr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dot(a, b):
    """Vector<T>::operator* (ref:305-328)"""
    suf = _suffix(a.dtype)
    out = _CT[suf]()
    n = a.size
    check(_fn("smm_hip_dot", suf)(n, _host(a, a.dtype, "a"), _host(b, a.dtype, "b", n), ctypes.byref(out)))
    return a.dtype.type(out.value)


def dot_dev(n, d_a, d_b, d_result, dtype, stream=None):
    check(_fn("smm_hip_dot_dev", _suffix(dtype))(int(n), _dptr(d_a), _dptr(d_b), _dptr(d_result), _dptr(stream)))


def _mh(M):
    return M._h if M is not None else ctypes.c_void_p(0)


def ConjugateGradient(a, b, x0, x, maxIterations, eps, M=None, info=None):
    """ref:2316-2398 (M = IC0 preconditioner: ref:2414-2505).  x may be x0.  Returns SolverStatus; `info`, when a
    dict, receives iterations and resnorm2."""
    suf = a._suf
    st, it, res = ctypes.c_int(), ctypes.c_int(), _CT[suf]()
    check(_fn("smm_hip_cg", suf)(a._h, _host(b, a.dtype, "b", a.rows), _host(x0, a.dtype, "x0", a.rows), _host(x, a.dtype, "x", a.rows, True),
                                 int(maxIterations), a.dtype.type(eps), _mh(M), ctypes.byref(st), ctypes.byref(it), ctypes.byref(res)))
    if info is not None:
        info.update(iterations=it.value, resnorm2=res.value)
    return SolverStatus(st.value)


def BiCGStab(a, b, x, maxIterations, eps, M=None, info=None):
    """ref:2191-2303.  x is the initial guess and receives the result."""
    suf = a._suf
    st, it, res = ctypes.c_int(), ctypes.c_int(), _CT[suf]()
    check(_fn("smm_hip_bicgstab", suf)(a._h, _host(b, a.dtype, "b", a.rows), _host(x, a.dtype, "x", a.rows, True), int(maxIterations),
                                       a.dtype.type(eps), _mh(M), ctypes.byref(st), ctypes.byref(it), ctypes.byref(res)))
    if info is not None:
        info.update(iterations=it.value, resnorm=res.value)
    return SolverStatus(st.value)


def BiCGSymmetric(a, b, x, maxIterations, eps, info=None):
    """ref:2021-2102"""
    suf = a._suf
    st, it = ctypes.c_int(), ctypes.c_int()
    check(_fn("smm_hip_bicgsymmetric", suf)(a._h, _host(b, a.dtype, "b", a.rows), _host(x, a.dtype, "x", a.rows, True), int(maxIterations),
                                            a.dtype.type(eps), ctypes.byref(st), ctypes.byref(it)))
    if info is not None:
        info.update(iterations=it.value)
    return SolverStatus(st.value)


def cg_dev(a, d_b, d_x0, d_x, maxIterations, eps, M=None, stream=None):
    """device-pointer CG; returns (SolverStatus, iterations, resnorm2).  Synchronises `stream`."""
    suf = a._suf
    st, it, res = ctypes.c_int(), ctypes.c_int(), _CT[suf]()
    check(_fn("smm_hip_cg_dev", suf)(a._h, _dptr(d_b), _dptr(d_x0), _dptr(d_x), int(maxIterations), a.dtype.type(eps), _mh(M), _dptr(stream),
                                     ctypes.byref(st), ctypes.byref(it), ctypes.byref(res)))
    return SolverStatus(st.value), it.value, res.value


CG_RESIDENT_OFF, CG_RESIDENT_AUTO, CG_RESIDENT_REQUIRE = 0, 1, 2


def cg_resident(mode=-1):
    """sets (0 off, 1 auto, 2 require) or only queries (-1) the register-resident CG path; returns the previous mode"""
    return int(_lib.load().smm_hip_cg_resident(int(mode)))


def set_cg_lazy_x_min_bytes(nbytes):
    """test / measurement knob: bytes per vector from which CG defers its x update (negative: the default, 64 MB)"""
    check(_lib.load().smm_hip_set_cg_lazy_x_min_bytes(int(nbytes)))


def set_cg_fuse_p(on):
    """test / measurement knob: False keeps CG from forming its next direction inside the 2.5-D SpMV kernel"""
    check(_lib.load().smm_hip_set_cg_fuse_p(1 if on else 0))


def bicgstab_resident(mode=-1):
    """sets (0 off, 1 auto, 2 require) or only queries (-1) the single-launch BiCGStab path; returns the previous mode"""
    return int(_lib.load().smm_hip_bicgstab_resident(int(mode)))


def bicgstab_dev(a, d_b, d_x, maxIterations, eps, M=None, stream=None):
    """device-pointer BiCGStab; returns (SolverStatus, iterations, resnorm).  Synchronises `stream`."""
    suf = a._suf
    st, it, res = ctypes.c_int(), ctypes.c_int(), _CT[suf]()
    check(_fn("smm_hip_bicgstab_dev", suf)(a._h, _dptr(d_b), _dptr(d_x), int(maxIterations), a.dtype.type(eps), _mh(M), _dptr(stream),
                                           ctypes.byref(st), ctypes.byref(it), ctypes.byref(res)))
    return SolverStatus(st.value), it.value, res.value


# ---- device-side generators (csrc/smm_gen.hip) ------------------------------------------------------------
def gen_banded_nnz(n, k=25, seed=0x5EED, max_offset=1 << 20):
    return int(_lib.load().smm_hip_gen_banded_nnz(int(n), int(k), int(seed), int(max_offset)))


def gen_poisson2d_nnz(nx, ny):
    return int(_lib.load().smm_hip_gen_poisson2d_nnz(int(nx), int(ny)))


def gen_stencil3d_nnz(nx, ny, nz):
    return int(_lib.load().smm_hip_gen_stencil3d_nnz(int(nx), int(ny), int(nz)))


def gen_banded_dev(n, k, seed, max_offset, d_start, d_positions, d_values, dtype, stream=None, diag_shift=1.0):
    check(_fn("smm_hip_gen_banded_dev", _suffix(dtype))(int(n), int(k), int(seed), int(max_offset), np.dtype(dtype).type(diag_shift), _dptr(d_start), _dptr(d_positions), _dptr(d_values), _dptr(stream)))


def gen_banded_row_start(n, k, seed, max_offset, row):
    """start[row] of the full banded matrix in closed form (no device needed)"""
    return int(_lib.load().smm_hip_gen_banded_row_start(int(n), int(k), int(seed), int(max_offset), int(row)))


def gen_banded_rows_dev(n, k, seed, max_offset, diag_shift, row_begin, row_end, d_start, d_positions, d_values, dtype, stream=None):
    """rows [row_begin, row_end) of the banded matrix: local start[], GLOBAL columns (what one rank owns)"""
    check(_fn("smm_hip_gen_banded_rows_dev", _suffix(dtype))(int(n), int(k), int(seed), int(max_offset), np.dtype(dtype).type(diag_shift),
                                                               int(row_begin), int(row_end), _dptr(d_start), _dptr(d_positions), _dptr(d_values),
                                                               _dptr(stream)))


def gen_poisson2d_dev(nx, ny, d_start, d_positions, d_values, dtype, stream=None):
    check(_fn("smm_hip_gen_poisson2d_dev", _suffix(dtype))(int(nx), int(ny), _dptr(d_start), _dptr(d_positions), _dptr(d_values), _dptr(stream)))


def gen_stencil3d_dev(nx, ny, nz, diag, lo, hi, d_start, d_positions, d_values, dtype, stream=None):
    t = np.dtype(dtype).type
    check(_fn("smm_hip_gen_stencil3d_dev", _suffix(dtype))(int(nx), int(ny), int(nz), t(diag), t(lo), t(hi), _dptr(d_start), _dptr(d_positions),
                                                            _dptr(d_values), _dptr(stream)))
