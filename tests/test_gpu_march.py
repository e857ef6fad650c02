"""The 2.5-D form of the constant-diagonal SpMV (spmvPatternConstMarchKernel, csrc/smm_spmv_march.hip): grid-shaped matrices of at
least 2^21 (values read: 6 x 2^20) rows march along the far direction with the plane's window of x in LDS.  Same products in the same order as the reference's
row loop (ref:1484-1499): every comparison below is bit for bit against the oracle."""
import numpy as np
import pytest

from oracle.oracle import OP_ADD, OP_ASSIGN, OP_SUB
from sparse_matrix_math_amd import generators as gen

pytestmark = pytest.mark.gpu
PATTERN, CONST = 3, 3


@pytest.fixture(autouse=True)
def _march_from_two_million_rows(smm):
    """production serves grids from 2^21 rows (constant diagonals) and 6 x 2^20 / 2^24 rows (values read, fp64 / fp32) with these
    kernels -- where they start to win (profiles/r04/march_threshold.txt); the parity tests run all of them on grids of 2.1 M rows to keep the
    oracle's side cheap"""
    smm.host.set_march_min_rows(1 << 21, 1 << 21)
    yield
    smm.host.set_march_min_rows(-1, -1)


def _stencil(smm, torch, nx, ny, nz, dtype, diag=6.0, lo=-1.25, hi=-0.75):
    dev = torch.device("cuda:0")
    td = torch.float32 if dtype == np.float32 else torch.float64
    n = nx * ny * nz
    stream = torch.cuda.current_stream().cuda_stream
    if nz > 1:
        nnz = smm.host.gen_stencil3d_nnz(nx, ny, nz)
    else:
        nnz = smm.host.gen_poisson2d_nnz(nx, ny)
    d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
    d_val = torch.empty(nnz, dtype=td, device=dev)
    if nz > 1:
        smm.host.gen_stencil3d_dev(nx, ny, nz, diag, lo, hi, d_start, d_pos, d_val, dtype, stream)
    else:
        smm.host.gen_poisson2d_dev(nx, ny, d_start, d_pos, d_val, dtype, stream)
    A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, dtype)
    return A, (d_start, d_pos, d_val), n, td, dev, stream


# (nx, ny, nz): a cube; planes that are no whole number of tiles (96 x 112 = 5.25 tiles) with an odd number of them; ONE plane whose every
# offset is near (a 2-D grid 1000 wide: halo 1000 of the 1024 a lane can hold in fp64)
GRIDS = [(128, 128, 128), (96, 112, 201), (1000, 2200, 1), (1024, 2100, 1)]  # (the last: the halo at the fp64 cap of 1024)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("grid", GRIDS)
def test_march_kernel_matches_oracle(smm, oracle, dtype, grid):
    import torch

    nx, ny, nz = grid
    A, (d_start, d_pos, d_val), n, td, dev, stream = _stencil(smm, torch, nx, ny, nz, dtype)
    A.set_kernel(PATTERN, 1)
    assert A.pattern_info()[0] == CONST
    name, nbytes = A.kernel_desc()
    assert name == "spmvPatternConstMarchKernel", name
    assert nbytes == n + 2 * n * np.dtype(dtype).itemsize  # one byte of mask per row (five near offsets: at most seven in all) + x + out
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand(n, dtype=td, device=dev, generator=g) - 0.5
    lhs = torch.rand(n, dtype=td, device=dev, generator=g) - 0.5
    y = torch.empty(n, dtype=td, device=dev)
    csr = (d_start.cpu().numpy(), d_pos.cpu().numpy(), d_val.cpu().numpy())
    xh, lh = x.cpu().numpy(), lhs.cpu().numpy()
    for op in (OP_ASSIGN, OP_ADD, OP_SUB):
        y.fill_(float("nan"))
        A.spmv_dev(op, lhs if op != OP_ASSIGN else None, x, y, stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(y.cpu().numpy(), oracle.spmv(csr, op, lh, xh))
    # in place (out aliases lhs, ref:1507-1515) and the fused dot products finished inside the launch
    z = lhs.clone()
    A.spmv_dev(OP_SUB, z, x, z, stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(z.cpu().numpy(), oracle.spmv(csr, OP_SUB, lh, xh))
    fin = torch.zeros(smm.host.finish_len(), dtype=td, device=dev)
    A.spmv_fused_dev(OP_ASSIGN, None, x, y, 2, x, fin, stream, finish=True)
    torch.cuda.synchronize()
    ref = oracle.spmv(csr, OP_ASSIGN, None, xh)
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    off = smm.host.finish_totals_offset()
    totals = fin.cpu().numpy().astype(np.float64)
    r64, x64 = ref.astype(np.float64), xh.astype(np.float64)
    tol = 1e-10 if dtype == np.float64 else 2e-5
    assert abs(totals[off] - float(r64 @ r64)) <= tol * float(r64 @ r64)
    assert abs(totals[off + 1] - float(r64 @ x64)) <= tol * float(np.abs(r64 * x64).sum())
    # with the constant-diagonal encoding off the same matrix is served by the kernels that read values[] (r04: their march form): same
    # products, same order, same bits
    A.pattern_allow_const(False)
    assert A.kernel_desc()[0] == ("spmvPatternMasksMarchKernel" if nz >= 8 else "spmvPatternWaveKernel")
    A.spmv_dev(OP_ASSIGN, None, x, y, stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    A.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_march_kernel_on_a_slab_that_ends_inside_a_plane(smm, oracle, dtype):
    """the rows of a row-partitioned grid are a slab that starts and ends anywhere: the leading 2 108 536 rows (and columns) of a
    128 x 128 x 129 stencil -- the last plane is a partial one -- still march (the offsets are linear in the row index; the masks carry
    the boundaries), bit for bit with the oracle"""
    import torch

    from sparse_matrix_math_amd import generators as gen

    start, pos, val = gen.stencil3d(128, 128, 129, 6.0, -1.25, -0.75, dtype=dtype)
    n = 128 * 128 * 129 - 5000
    keep = pos[: start[n]] < n
    rows_of = np.repeat(np.arange(n), np.diff(start[: n + 1]))
    s2 = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows_of[keep], minlength=n), out=s2[1:])
    csr = (s2, pos[: start[n]][keep].copy(), val[: start[n]][keep].copy())
    A = smm.CSRMatrix(n, n, *csr)
    A.set_kernel(PATTERN, 1)
    assert A.pattern_info()[0] == CONST and A.kernel_desc()[0] == "spmvPatternConstMarchKernel"
    rng = np.random.default_rng(6)
    x, lhs = rng.uniform(-0.5, 0.5, n).astype(dtype), rng.uniform(-0.5, 0.5, n).astype(dtype)
    out = np.zeros(n, dtype=dtype)
    A.rMult(x, out)
    np.testing.assert_array_equal(out, oracle.spmv(csr, OP_ASSIGN, None, x))
    A.rMultSub(lhs, x, out)
    np.testing.assert_array_equal(out, oracle.spmv(csr, OP_SUB, lhs, x))
    A.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_masks_march_kernel_with_varying_coefficients(smm, oracle, dtype):
    """spmvPatternMasksMarchKernel: the march for stencils whose diagonals VARY (values[] read through a wave-private LDS slice, x
    through the plane's window and the lane's registers, row starts from one start[] per 64 rows + a prefix sum of the masks'
    popcounts).  The 128^3 convection-diffusion operator with spatially varying coefficients (2.1 M rows, every diagonal varies) -- and a
    2-D one-plane matrix, which must stay with the wave kernel --, all three ops, in place, the fused dot products: bit for bit against the oracle (ref:1484-1499)."""
    import torch

    from sparse_matrix_math_amd import generators as gen

    dev = torch.device("cuda:0")
    td = torch.float32 if dtype == np.float32 else torch.float64
    stream = torch.cuda.current_stream().cuda_stream
    cases = {"convdiff_varying_128": gen.convdiff3d_varying(128, 0.3, dtype=dtype)}
    start, pos, val = gen.poisson2d(1024, 2100, dtype=dtype)  # one plane: every offset near (the halo at the fp64 cap); values perturbed so that no diagonal is constant
    val = (val * (1 + 0.25 * np.sin(np.arange(len(val)) * 0.37))).astype(dtype)
    cases["poisson2d_varying_1024x2100"] = (start, pos, val)
    for name, csr in cases.items():
        n = len(csr[0]) - 1
        A = smm.CSRMatrix(n, n, *csr)
        A.set_kernel(PATTERN, 1)
        assert A.pattern_info()[0] == 1, name  # row masks + values[]
        kernel, nbytes = A.kernel_desc()
        s = np.dtype(dtype).itemsize
        if name.startswith("convdiff"):
            assert kernel == "spmvPatternMasksMarchKernel", (name, kernel)
            assert nbytes == len(csr[1]) * s + n + (n // 64 + 1) * 4 + 2 * n * s  # (one byte of mask per row)
        else:  # ONE plane: nothing could be requested ahead -- the wave kernel keeps such matrices (measured: it is faster there)
            assert kernel == "spmvPatternWaveKernel", (name, kernel)
        rng = np.random.default_rng(12)
        x, lhs = rng.uniform(-0.5, 0.5, n).astype(dtype), rng.uniform(-0.5, 0.5, n).astype(dtype)
        out = np.zeros(n, dtype=dtype)
        for op, call in ((OP_ASSIGN, lambda: A.rMult(x, out)), (OP_ADD, lambda: A.rMultAdd(lhs, x, out)), (OP_SUB, lambda: A.rMultSub(lhs, x, out))):
            out[:] = np.nan
            call()
            np.testing.assert_array_equal(out, oracle.spmv(csr, op, lhs, x), err_msg=f"{name} op {op}")
        z = lhs.copy()
        A.rMultSub(z, x, z)
        np.testing.assert_array_equal(z, oracle.spmv(csr, OP_SUB, lhs, x))
        dx, dy = torch.from_numpy(x).to(dev), torch.empty(n, dtype=td, device=dev)
        fin = torch.zeros(smm.host.finish_len(), dtype=td, device=dev)
        A.spmv_fused_dev(OP_ASSIGN, None, dx, dy, 2, dx, fin, stream, finish=True)
        torch.cuda.synchronize()
        ref = oracle.spmv(csr, OP_ASSIGN, None, x)
        np.testing.assert_array_equal(dy.cpu().numpy(), ref)
        off = smm.host.finish_totals_offset()
        totals = fin.cpu().numpy().astype(np.float64)
        r64, x64 = ref.astype(np.float64), x.astype(np.float64)
        tol = 1e-10 if dtype == np.float64 else 2e-5
        assert abs(totals[off] - float(r64 @ r64)) <= tol * float(r64 @ r64)
        assert abs(totals[off + 1] - float(r64 @ x64)) <= tol * float(np.abs(r64 * x64).sum())
        A.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("grid,points", [((128, 128, 128), 27), ((96, 112, 201), 19), ((160, 100, 140), 27)])
def test_three_window_march_for_clustered_far_offsets(smm, oracle, dtype, grid, points):
    """19- and 27-point stencils (HPCG's matrix shape): the far offsets come in clusters around -P and +P, so the planes below and above are
    LDS windows too (spmvPatternConstMarch3Kernel, four rotating windows, one barrier per plane).  Every operation bit for bit against the
    oracle; planes that are no whole number of tiles, an odd number of planes, the fused dot products."""
    import torch

    nx, ny, nz = grid
    csr = gen.stencil3d_wide(nx, ny, nz, points, dtype=dtype)
    n = len(csr[0]) - 1
    dev = torch.device("cuda:0")
    td = torch.float32 if dtype == np.float32 else torch.float64
    stream = torch.cuda.current_stream().cuda_stream
    d = [torch.from_numpy(a).to(dev) for a in csr]
    A = smm.CSRMatrix.from_device(n, n, d[0], d[1], d[2], dtype)
    A.set_kernel(PATTERN, 1)
    assert A.pattern_info() == (CONST, points)
    name, nbytes = A.kernel_desc()
    assert name == "spmvPatternConstMarch3Kernel", name
    assert nbytes == n * 4 + 2 * n * np.dtype(dtype).itemsize
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.rand(n, dtype=td, device=dev, generator=g) - 0.5
    lhs = torch.rand(n, dtype=td, device=dev, generator=g) - 0.5
    y = torch.empty(n, dtype=td, device=dev)
    xh, lh = x.cpu().numpy(), lhs.cpu().numpy()
    for op in (OP_ASSIGN, OP_ADD, OP_SUB):
        y.fill_(float("nan"))
        A.spmv_dev(op, lhs if op != OP_ASSIGN else None, x, y, stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(y.cpu().numpy(), oracle.spmv(csr, op, lh, xh))
    z = lhs.clone()
    A.spmv_dev(OP_SUB, z, x, z, stream)  # in place (out aliases lhs, ref:1507-1515)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(z.cpu().numpy(), oracle.spmv(csr, OP_SUB, lh, xh))
    fin = torch.zeros(smm.host.finish_len(), dtype=td, device=dev)
    A.spmv_fused_dev(OP_ASSIGN, None, x, y, 2, x, fin, stream, finish=True)
    torch.cuda.synchronize()
    ref = oracle.spmv(csr, OP_ASSIGN, None, xh)
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    tot = fin.cpu().numpy()[smm.host.finish_totals_offset():smm.host.finish_totals_offset() + 2]
    rtol = 1e-4 if dtype == np.float32 else 1e-11
    np.testing.assert_allclose(tot, [np.dot(ref.astype(np.float64), ref), np.dot(ref.astype(np.float64), xh)], rtol=rtol, atol=rtol * n)
    # the gather kernel (march off for this handle's size class is not switchable per handle: compare with the CSR stream at one lane instead)
    A.set_kernel(2, 1)
    y2 = torch.empty_like(y)
    A.spmv_dev(OP_ASSIGN, None, x, y2, stream)
    torch.cuda.synchronize()
    assert torch.equal(y2, y)


def test_march_lds_is_raised_again_for_a_larger_halo(smm):
    """Two grids in ONE process, the second with the larger halo (nx = 800, then nx = 1024; 13.1 M rows each, fp64, production thresholds):
    the masks march needs 42 KB of dynamic LDS for the first and 49 KB -- with its 16.6 KB of static LDS more than the 64 KB a launch gets
    by default -- for the second.  r04 set hipFuncAttributeMaxDynamicSharedMemorySize ONCE per instantiation, to the first qualifying
    launch's size, so the second matrix got a failed launch; now the largest size granted so far is tracked and raised on demand (VERDICT
    r04 item 5b).  Both kernels (constant diagonals, values read) against the CSR stream at one lane per row, bit for bit."""
    import torch

    smm.host.set_march_min_rows(-1, -1)  # production thresholds
    for nx, ny, nz in ((800, 16, 1024), (1024, 16, 800)):
        A, _, n, td, dev, stream = _stencil(smm, torch, nx, ny, nz, np.float64)
        x = torch.rand(n, dtype=td, device=dev, generator=torch.Generator(device=dev).manual_seed(nx)) - 0.5
        A.set_kernel(2, 1)
        y_csr = torch.empty_like(x)
        A.spmv_dev(0, None, x, y_csr, stream)
        A.set_kernel(PATTERN, 1)
        assert A.kernel_desc()[0] == "spmvPatternConstMarchKernel"
        y = torch.full_like(x, float("nan"))
        A.spmv_dev(0, None, x, y, stream)
        torch.cuda.synchronize()
        assert torch.equal(y, y_csr), (nx, "const")
        A.pattern_allow_const(False)
        assert A.kernel_desc()[0] == "spmvPatternMasksMarchKernel"
        y.fill_(float("nan"))
        A.spmv_dev(0, None, x, y, stream)
        torch.cuda.synchronize()
        assert torch.equal(y, y_csr), (nx, "masks")
        del A, x, y, y_csr
        torch.cuda.empty_cache()


def test_march_in_cg_at_scale(smm, oracle):
    """ConjugateGradient on a 160^3 Laplacian (4.1 M rows, fp64) with AUTO: the solver adopts PATTERN / CONST, the SpMV is the march kernel;
    x after 25 iterations against the oracle (fixed iterations: eps = 0)"""
    import torch

    N = 160
    A, (d_start, d_pos, d_val), n, td, dev, stream = _stencil(smm, torch, N, N, N, np.float64, 6.0, -1.0, -1.0)
    ones = torch.ones(n, dtype=td, device=dev)
    b = torch.empty(n, dtype=td, device=dev)
    A.spmv_dev(OP_ASSIGN, None, ones, b, stream)
    x = torch.zeros(n, dtype=td, device=dev)
    st, it, _res = smm.host.cg_dev(A, b, x, x, 25, 0.0, None, stream)
    torch.cuda.synchronize()
    assert A.get_kernel() == (PATTERN, 1) and A.kernel_desc()[0] == "spmvPatternConstMarchKernel"
    csr = (d_start.cpu().numpy(), d_pos.cpu().numpy(), d_val.cpu().numpy())
    st_ref, x_ref, it_ref, _ = oracle.cg(csr, b.cpu().numpy(), np.zeros(n), 25, 0.0)
    assert int(st) == st_ref and it == it_ref == 25
    np.testing.assert_allclose(x.cpu().numpy(), x_ref, rtol=0, atol=1e-10 * float(np.abs(x_ref).max()))
    A.close()


def test_march_kernels_fuzz_below_the_production_threshold():
    """tools/march_fuzz.py in a process of its own (SMM_HIP_MARCH_MIN_ROWS=1 must be in the environment before the library reads it): 48
    random grid-shaped matrices of 20 K - 300 K rows -- partial tiles, partial last planes, one or both far offsets (single or in clusters), one-plane bands, random
    holes in every diagonal, constant and varying values, fp32 / fp64 -- every SpMV (three ops, in place) bit-identical to the oracle"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "march_fuzz.py"), "48"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "march fuzz: ALL OK" in r.stdout
    assert r.stdout.count("spmvPatternConstMarchKernel ok") >= 10 and r.stdout.count("spmvPatternMasksMarchKernel ok") >= 3, r.stdout[-3000:]
    assert r.stdout.count("spmvPatternConstMarch3Kernel ok") >= 4, r.stdout[-3000:]
    # the masks march takes two sub-steps per tile at these sizes (fp64: below 1e8 rows); the four-sub-step form of the big fp64 grids, forced
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "march_fuzz.py"), "32"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SMM_HIP_MASKS_MARCH_Q="4"))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "march fuzz: ALL OK" in r.stdout and r.stdout.count("spmvPatternMasksMarchKernel ok") >= 2, r.stdout[-3000:]
