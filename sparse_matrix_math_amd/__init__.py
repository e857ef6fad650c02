"""sparse_matrix_math_amd -- MI355X (gfx950) implementation of the CSR SpMV + Krylov hot path of
vasil-pashov/sparse_matrix_math (SMM::CSRMatrix::rMult*, SMM::ConjugateGradient, SMM::BiCGStab).

The compute lives in hand-written HIP kernels behind the C ABI of include/smm_hip.h (libsmm_hip.so, built in-tree
under sparse_matrix_math_amd/lib by __graft_entry__.build()).  This package is the thin host side: ctypes binding,
the reference's API names, and the synthetic workload generators.  No CPU fallback exists.
"""
from . import generators  # noqa: F401
from .host import (  # noqa: F401
    OP_ADD,
    OP_ASSIGN,
    OP_SUB,
    SPMV_AUTO,
    SPMV_PATTERN,
    SPMV_STREAM,
    SPMV_VECTOR,
    BiCGStab,
    BiCGSymmetric,
    ConjugateGradient,
    CSRMatrix,
    Preconditioner,
    SolverPreconditioner,
    SolverStatus,
    bicgstab_dev,
    cg_dev,
    device_info,
    dot,
    dot_dev,
    init,
    synchronize,
    uses_std_fma,
)
from ._lib import SmmHipError  # noqa: F401

__version__ = "0.1.0"
