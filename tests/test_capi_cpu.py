"""CPU-side checks of the boundary: the shared library loads, exports every symbol include/smm_hip.h declares, and
refuses to compute without a GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from sparse_matrix_math_amd import _lib
from sparse_matrix_math_amd import generators as gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "smm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smm_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) > 50
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/smm_hip.h but not exported by libsmm_hip.so"
    assert sorted(_lib.exported_symbols()) == declared, "ctypes prototype table out of sync with the header"


def test_fma_flavour_exports_too():
    path = _lib.library_path(fma=True)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    lib.smm_hip_uses_std_fma.restype = ctypes.c_int
    assert lib.smm_hip_uses_std_fma() == 1
    assert _lib.load().smm_hip_uses_std_fma() == 0
    for name in header_functions():
        assert hasattr(lib, name)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    """the product path must fail loudly, never compute on the CPU"""
    import sparse_matrix_math_amd as smm

    with pytest.raises(smm.SmmHipError) as e:
        smm.init(0)
    assert e.value.code == _lib.SMM_HIP_ERR_NO_DEVICE
    start, pos, val = gen.poisson2d(4)
    with pytest.raises(smm.SmmHipError) as e:
        smm.CSRMatrix(16, 16, start, pos, val)
    assert e.value.code == _lib.SMM_HIP_ERR_NO_DEVICE
    with pytest.raises(smm.SmmHipError):
        smm.dot(np.ones(4), np.ones(4))


def test_product_never_imports_oracle():
    """nothing under sparse_matrix_math_amd/ or include/ may reference oracle/"""
    for base in ("sparse_matrix_math_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "/lib" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "smm_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_closed_form_nnz_matches_generators():
    lib = _lib.load()
    for nx, ny in ((1, 1), (3, 5), (32, 32), (7, 2)):
        assert lib.smm_hip_gen_poisson2d_nnz(nx, ny) == len(gen.poisson2d(nx, ny)[1])
    for dims in ((1, 1, 1), (3, 4, 5), (12, 12, 12), (2, 9, 1)):
        assert lib.smm_hip_gen_stencil3d_nnz(*dims) == len(gen.stencil3d(*dims)[1])
    for n, k, seed, mo in ((2000, 25, 0x5EED, 1 << 20), (10, 25, 1, 1 << 20), (500, 3, 7, 50), (1, 5, 3, 100), (0, 5, 3, 100)):
        assert lib.smm_hip_gen_banded_nnz(n, k, seed, mo) == len(gen.banded_random_spd(n, k, seed, mo)[1])


def test_generator_properties():
    start, pos, val = gen.banded_random_spd(3000, k=25, seed=0x5EED, max_offset=1 << 20, dtype=np.float32)
    n = 3000
    import scipy.sparse as sp

    A = sp.csr_matrix((val.astype(np.float64), pos, start), shape=(n, n))
    assert (abs(A - A.T)).max() == 0  # symmetric by construction
    d = A.diagonal()
    off = abs(A).sum(axis=1).A1 - abs(d)
    assert np.all(d > off)  # strictly diagonally dominant -> SPD
    for r in range(0, n, 97):
        row = pos[start[r]:start[r + 1]]
        assert np.all(np.diff(row) > 0)  # ascending columns, as the reference's layout requires (ref:1247-1249)
    assert len(gen.band_offsets(10_000_000)) == 25
