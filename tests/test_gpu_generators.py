"""device generators (csrc/smm_gen.hip) == numpy generators, bit for bit"""
import numpy as np
import pytest

from sparse_matrix_math_amd import generators as gen

pytestmark = pytest.mark.gpu


def run_dev(smm, kind, args, dtype, nnz, rows, **kw):
    import torch

    dev = torch.device("cuda:0")
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    d_start = torch.full((rows + 1,), -1, dtype=torch.int32, device=dev)
    d_pos = torch.full((max(nnz, 1),), -1, dtype=torch.int32, device=dev)
    d_val = torch.zeros(max(nnz, 1), dtype=tdt, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    getattr(smm.host, f"gen_{kind}_dev")(*args, d_start, d_pos, d_val, dtype, stream, **kw)
    torch.cuda.synchronize()
    return d_start.cpu().numpy(), d_pos.cpu().numpy()[:nnz], d_val.cpu().numpy()[:nnz]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_banded(smm, dtype):
    for n, k, seed, mo, shift in ((2000, 25, 0x5EED, 1 << 20, 1.0), (50000, 25, 0x5EED, 1 << 20, 0.01), (10, 25, 1, 1 << 20, 1.0), (500, 3, 7, 50, 0.5),
                                  (1, 5, 3, 100, 1.0)):
        want = gen.banded_random_spd(n, k, seed, mo, dtype, shift)
        got = run_dev(smm, "banded", (n, k, seed, mo), dtype, len(want[1]), n, diag_shift=shift)
        for w, g in zip(want, got):
            np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_stencils(smm, dtype):
    for nx, ny in ((1, 1), (3, 5), (32, 32), (7, 2), (100, 100)):
        want = gen.poisson2d(nx, ny, dtype)
        got = run_dev(smm, "poisson2d", (nx, ny), dtype, len(want[1]), nx * ny)
        for w, g in zip(want, got):
            np.testing.assert_array_equal(g, w)
    for dims in ((1, 1, 1), (3, 4, 5), (12, 12, 12), (2, 9, 1), (30, 20, 10)):
        for diag, lo, hi in ((6.0, -1.0, -1.0), (6.0, -1.3, -0.7)):
            want = gen.stencil3d(*dims, diag, lo, hi, dtype)
            got = run_dev(smm, "stencil3d", (*dims, diag, lo, hi), dtype, len(want[1]), dims[0] * dims[1] * dims[2])
            for w, g in zip(want, got):
                np.testing.assert_array_equal(g, w)
