"""The SMM_WITH_STD_FMA flavour (ref:31-35: _smm_fma = std::fma) of the library, libsmm_hip_fma.so, against the oracle built the
same way: the row-sequential SpMV and the level-scheduled sweeps are bit-identical fma chains."""
import ctypes

import numpy as np
import pytest

from sparse_matrix_math_amd import _lib
from sparse_matrix_math_amd import generators as gen

pytestmark = pytest.mark.gpu


def test_fma_library_matches_fma_oracle(oracle, oracle_fma):
    _lib._share_hip_runtime_with_torch()
    lib = ctypes.CDLL(_lib.library_path(fma=True))
    lib.smm_hip_last_error.restype = ctypes.c_char_p
    assert lib.smm_hip_uses_std_fma() == 1
    assert lib.smm_hip_init(0) == 0, lib.smm_hip_last_error()
    P = ctypes.c_void_p
    for dtype, suf in ((np.float32, "f32"), (np.float64, "f64")):
        csr = gen.banded_random_spd(5000, k=7, seed=21, max_offset=900, dtype=dtype)  # 15 nnz/row -> one lane per row
        start, pos, val = csr
        n = len(start) - 1
        rng = np.random.default_rng(3)
        x = rng.uniform(-1, 1, n).astype(dtype)
        lhs = rng.uniform(-1, 1, n).astype(dtype)
        h = P()
        ptr = lambda a: a.ctypes.data_as(P)  # noqa: E731
        assert getattr(lib, f"smm_hip_csr_create_{suf}")(n, n, ptr(start), ptr(pos), ptr(val), ctypes.byref(h)) == 0
        out = np.zeros(n, dtype=dtype)
        assert getattr(lib, f"smm_hip_spmv_{suf}")(h, 2, ptr(lhs), ptr(x), ptr(out)) == 0, lib.smm_hip_last_error()
        want = oracle_fma.spmv(csr, 2, lhs, x)
        np.testing.assert_array_equal(out, want)
        assert not np.array_equal(out, oracle.spmv(csr, 2, lhs, x))  # and it really is the other rounding
        M = P()
        assert lib.smm_hip_precond_create(h, 3, ctypes.byref(M)) == 0  # SGS
        sx = np.zeros(n, dtype=dtype)
        assert getattr(lib, f"smm_hip_precond_apply_{suf}")(M, ptr(lhs), ptr(sx)) == 0
        np.testing.assert_array_equal(sx, oracle_fma.sgs_apply(csr, lhs)[1])
        # solver: fixed iterations against the fma oracle
        b = gen.row_sums(start, val)
        xs = np.zeros(n, dtype=dtype)
        st, it = ctypes.c_int(), ctypes.c_int()
        ct = ctypes.c_float if dtype == np.float32 else ctypes.c_double
        fn = getattr(lib, f"smm_hip_bicgstab_{suf}")
        fn.argtypes = [P, P, P, ctypes.c_int, ct, P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), P]
        x_true = rng.uniform(0.5, 1.5, n).astype(dtype)
        b = oracle_fma.spmv(csr, 0, None, x_true)
        assert fn(h, ptr(b.copy()), ptr(xs), 5, ct(0.0), None, ctypes.byref(st), ctypes.byref(it), None) == 0
        st_ref, x_ref, it_ref, _ = oracle_fma.bicgstab(csr, b, np.zeros(n, dtype=dtype), 5, 0.0)
        assert st.value == st_ref and it.value == it_ref == 5
        assert float(np.max(np.abs(xs - x_ref))) <= (3e-4 if dtype == np.float32 else 1e-10) * float(np.max(np.abs(x_ref)))
        lib.smm_hip_precond_destroy(M)
        lib.smm_hip_csr_destroy(h)
