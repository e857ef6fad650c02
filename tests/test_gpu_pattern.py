"""The opt-in PATTERN SpMV family (SMM_SPMV_PATTERN, smm_spmv_pattern.hip): positions[] replaced by a shared offset list
and one mask per row.  It must give the reference's numbers bit for bit (one lane per row = the reference's order,
ref:1484-1499), agree bit for bit with the STREAM family at equal lanes per row, and refuse matrices without a pattern."""
import numpy as np
import pytest

from oracle.oracle import OP_ADD, OP_ASSIGN, OP_SUB
from sparse_matrix_math_amd import generators as gen

pytestmark = pytest.mark.gpu
PATTERN, STREAM = 3, 2


def matrices(dtype):
    return {
        "poisson2d_37x23": gen.poisson2d(37, 23, dtype=dtype),
        "poisson2d_200": gen.poisson2d(200, dtype=dtype),
        "convdiff3d_17": gen.convdiff3d(17, dtype=dtype),
        "stencil3d_40x9x21": gen.stencil3d(40, 9, 21, dtype=dtype),
        "banded_20k_k25": gen.banded_random_spd(20_000, 25, 0x5EED, 4096, dtype=dtype),
        "banded_3k_k5": gen.banded_random_spd(3000, 5, 7, 100, dtype=dtype),
        "tiny": gen.poisson2d(2, dtype=dtype),
    }


def run(A, op, lhs, x, rows, dtype):
    out = np.zeros(rows, dtype=dtype)
    {OP_ASSIGN: lambda: A.rMult(x, out), OP_ADD: lambda: A.rMultAdd(lhs, x, out), OP_SUB: lambda: A.rMultSub(lhs, x, out)}[op]()
    return out


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_pattern_matches_oracle_and_stream(smm, oracle, dtype):
    rng = np.random.default_rng(3)
    for name, csr in matrices(dtype).items():
        rows = len(csr[0]) - 1
        A = smm.CSRMatrix(rows, rows, *csr)
        x = rng.uniform(-1, 1, rows).astype(dtype)
        lhs = rng.uniform(-1, 1, rows).astype(dtype)
        for op in (OP_ASSIGN, OP_ADD, OP_SUB):
            ref = oracle.spmv(csr, op, lhs, x)
            for lanes in (1, 2, 4, 8):
                A.set_kernel(STREAM, lanes)
                want = run(A, op, lhs, x, rows, dtype)
                A.set_kernel(PATTERN, lanes)
                assert A.get_kernel() == (PATTERN, lanes)
                got = run(A, op, lhs, x, rows, dtype)
                # rows inside the last LDS-capacity's worth of nonzeros are summed by one lane each whatever `lanes` is (the
                # "direct" tail path); the two families have different capacities, so only the rows before both tails must agree
                # bit for bit when lanes > 1 -- the tail rows agree within the re-ordering bound
                body = csr[0][1:] <= csr[0][-1] - 8200 if lanes > 1 else np.ones(rows, dtype=bool)
                np.testing.assert_array_equal(got[body], want[body], err_msg=f"{name} op {op} lanes {lanes}")
                mag = np.zeros(rows)
                np.add.at(mag, np.repeat(np.arange(rows), np.diff(csr[0])), np.abs(csr[2].astype(np.float64) * x[csr[1]]))
                assert np.all(np.abs(got.astype(np.float64) - want) <= 64 * np.finfo(dtype).eps * (mag + np.abs(lhs))), (name, op, lanes)
                if lanes == 1:
                    np.testing.assert_array_equal(got, ref, err_msg=f"{name} op {op}")


def test_pattern_solver(smm, oracle):
    """whole BiCGStab / CG solves through the PATTERN kernel, fused dot epilogues included.  The SpMV results are the STREAM
    family's bit for bit, but the fused dot products are summed per workgroup and the two families deal rows to workgroups
    differently, so the solves agree to rounding, not bitwise: same status, same iteration count, same solution to 1e-5"""
    csr = gen.banded_random_spd(30_000, 25, 0x5EED, 8192, dtype=np.float32)
    n = len(csr[0]) - 1
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, n).astype(np.float32)
    res = {}
    for family in (STREAM, PATTERN):
        A = smm.CSRMatrix(n, n, *csr)
        A.set_kernel(family, 1)
        x = np.zeros(n, dtype=np.float32)
        info = {}
        st = smm.BiCGStab(A, b, x, 40, 1e-6, info=info)
        x2 = np.zeros(n, dtype=np.float32)
        info2 = {}
        st2 = smm.ConjugateGradient(A, b, np.zeros(n, dtype=np.float32), x2, 60, 1e-6, info=info2)
        res[family] = (int(st), info.get("iterations"), x, int(st2), info2.get("iterations"), x2)
        # and it IS a solution: ||b - A x|| / ||b|| from the oracle's SpMV
        r = oracle.spmv(csr, OP_SUB, b, x)
        assert np.linalg.norm(r) <= 2e-5 * np.linalg.norm(b), family
    assert res[STREAM][0] == res[PATTERN][0] == 0 and res[STREAM][1] == res[PATTERN][1]
    np.testing.assert_allclose(res[STREAM][2], res[PATTERN][2], rtol=0, atol=1e-5)
    assert res[STREAM][3] == res[PATTERN][3] == 0 and abs(res[STREAM][4] - res[PATTERN][4]) <= 1
    np.testing.assert_allclose(res[STREAM][5], res[PATTERN][5], rtol=0, atol=1e-5)


MASKS, CODES = 1, 2


def test_pattern_refuses_matrices_without_one(smm):
    # more than 65536 distinct offsets (two random columns per row): neither encoding fits
    rng = np.random.default_rng(4)
    n = 200_000
    pos = np.sort(rng.integers(0, n, size=(n, 2)), axis=1)
    pos[:, 1] = np.where(pos[:, 1] == pos[:, 0], (pos[:, 0] + 1) % n, pos[:, 1])
    pos = np.sort(pos, axis=1).astype(np.int32).ravel()
    start = (2 * np.arange(n + 1)).astype(np.int32)
    A = smm.CSRMatrix(n, n, start, pos, rng.uniform(-1, 1, 2 * n))
    before = A.get_kernel()
    with pytest.raises(smm.SmmHipError):
        A.set_kernel(PATTERN, 0)
    assert A.get_kernel() == before and A.pattern_info() == (0, 0)
    with pytest.raises(smm.SmmHipError):  # (the refusal is remembered)
        A.set_kernel(PATTERN, 1)
    # empty matrix
    E = smm.CSRMatrix(5, 5, np.zeros(6, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0))
    with pytest.raises(smm.SmmHipError):
        E.set_kernel(PATTERN, 1)


def test_masks_give_way_to_codes(smm, oracle):
    """what the row masks cannot describe goes to the dictionary encoding, same bits"""
    rng = np.random.default_rng(12)
    # a stencil with ONE entry moved off the pattern in a row the sampling does not look at: the full verification of the masks finds it
    start, pos, val = gen.poisson2d(300)
    pos = pos.copy()
    row = 44_444
    e = start[row]  # first entry of the row (column row - 300): move it one to the left, still ascending
    assert pos[e] == row - 300
    pos[e] -= 1
    cases = {
        "stencil_with_a_stray_entry": ((start, pos, val), 90_000, 6),
        "random_columns": (gen.random_rows(500, 500, 1, 9, seed=4), 500, None),   # hundreds of offsets
        "rows_of_80_to_90_entries": (gen.random_rows(100, 4000, 80, 90, seed=2), 4000, None),
    }
    for name, (csr, cols, k) in cases.items():
        rows = len(csr[0]) - 1
        A = smm.CSRMatrix(rows, cols, *csr)
        A.set_kernel(PATTERN, 1)
        enc, found = A.pattern_info()
        assert enc == CODES and (k is None or found == k), name
        rowof = np.repeat(np.arange(rows), np.diff(csr[0]))
        assert found == len(np.unique(csr[1].astype(np.int64) - rowof))
        x = rng.uniform(-1, 1, cols)
        out = np.zeros(rows)
        A.rMult(x, out)
        np.testing.assert_array_equal(out, oracle.spmv(csr, OP_ASSIGN, None, x), err_msg=name)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_codes_match_oracle_and_stream(smm, oracle, dtype):
    """the dictionary encoding (> 64 offsets): every lane count and operation against STREAM bit for bit, one lane per row against the
    oracle bit for bit (ref:1484-1499); dictionaries that live in LDS (<= 2048 offsets) and in global memory"""
    rng = np.random.default_rng(8)
    big = gen.random_rows(6000, 6000, 3, 40, seed=21, dtype=dtype)
    cases = {
        "banded_65_diagonals": (gen.banded_random_spd(6000, 32, 0x5EED, 2500, dtype=dtype), 6000),
        "banded_33_diagonals_wide": (gen.banded_random_spd(20_000, 16, 3, 9000, dtype=dtype), 20_000),  # <= 64 offsets: the masks take it
        "random_6000": (big, 6000),                                               # ~ 11 000 offsets: dictionary in global memory
        "ragged_with_empty_rows": (gen.random_rows(3000, 2500, 0, 130, seed=5, dtype=dtype, empty_every=7), 2500),
        "70_entries_in_the_only_row": (gen.random_rows(1, 400, 70, 70, seed=1, dtype=dtype), 400),
        # ~ 1900 offsets (a dictionary in LDS close to its limit) with rows of 60-64 entries (the largest tiles): the kernel's LDS budget
        "long_rows_large_lds_dictionary": (gen.random_rows(1000, 1000, 60, 64, seed=17, dtype=dtype), 1000),
    }
    for name, (csr, cols) in cases.items():
        rows = len(csr[0]) - 1
        csr = (csr[0].astype(np.int32), csr[1], csr[2])
        A = smm.CSRMatrix(rows, cols, *csr)
        x = rng.uniform(-1, 1, cols).astype(dtype)
        lhs = rng.uniform(-1, 1, rows).astype(dtype)
        A.set_kernel(PATTERN, 1)
        enc, k = A.pattern_info()
        if name == "banded_33_diagonals_wide":
            assert enc == MASKS and k == 33
            continue
        assert enc == CODES, name
        if name == "random_6000":
            assert k > 2048
        for op in (OP_ASSIGN, OP_ADD, OP_SUB):
            ref = oracle.spmv(csr, op, lhs, x)
            for lanes in (1, 2, 4, 8):
                A.set_kernel(STREAM, lanes)
                want = run(A, op, lhs, x, rows, dtype)
                A.set_kernel(PATTERN, lanes)
                got = run(A, op, lhs, x, rows, dtype)
                body = csr[0][1:] <= csr[0][-1] - 8200 if lanes > 1 else np.ones(rows, dtype=bool)
                np.testing.assert_array_equal(got[body], want[body], err_msg=f"{name} op {op} lanes {lanes}")
                mag = np.zeros(rows)
                np.add.at(mag, np.repeat(np.arange(rows), np.diff(csr[0])), np.abs(csr[2].astype(np.float64) * x[csr[1]]))
                assert np.all(np.abs(got.astype(np.float64) - want) <= 64 * np.finfo(dtype).eps * (mag + np.abs(lhs))), (name, op, lanes)
                if lanes == 1:
                    np.testing.assert_array_equal(got, ref, err_msg=f"{name} op {op}")


def test_auto_uses_codes_for_a_large_matrix_with_many_diagonals(smm, oracle):
    """65 diagonals, 2^25 entries: AUTO's attempt fails the masks and takes the dictionary; bits of STREAM at the same lanes; BiCGStab +
    Jacobi (the divide epilogue) against the oracle"""
    import torch
    from oracle.oracle import PRECOND_JACOBI

    dev = torch.device("cuda:0")
    n, k, dtype = 600_000, 32, np.float32  # (the generator draws at most 32 offsets per side: 65 diagonals)
    nnz = smm.host.gen_banded_nnz(n, k, 0x5EED, 1 << 16)
    assert nnz >= 1 << 25
    d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
    d_val = torch.empty(nnz, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    smm.host.gen_banded_dev(n, k, 0x5EED, 1 << 16, d_start, d_pos, d_val, dtype, stream)
    A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, dtype)
    x = torch.rand(n, dtype=torch.float32, device=dev) - 0.5
    lhs = torch.rand(n, dtype=torch.float32, device=dev) - 0.5
    y_auto = torch.empty(n, dtype=torch.float32, device=dev)
    A.spmv_dev(OP_SUB, lhs, x, y_auto, stream)
    torch.cuda.synchronize()
    fam, lanes = A.get_kernel()
    assert fam == PATTERN and A.pattern_info() == (CODES, 2 * k + 1)
    A.set_kernel(STREAM, lanes)
    y_stream = torch.empty_like(y_auto)
    A.spmv_dev(OP_SUB, lhs, x, y_stream, stream)
    torch.cuda.synchronize()
    body = (d_start[1:] <= nnz - 8200).cpu().numpy()
    np.testing.assert_array_equal(y_auto.cpu().numpy()[body], y_stream.cpu().numpy()[body])
    A.set_kernel(smm.SPMV_AUTO, 0)
    assert A.get_kernel()[0] == PATTERN
    csr = (d_start.cpu().numpy(), d_pos.cpu().numpy(), d_val.cpu().numpy())
    M = A.getPreconditioner(smm.SolverPreconditioner.JACOBI)
    x_true = np.random.default_rng(5).uniform(0.5, 1.5, n).astype(dtype)
    b = oracle.spmv(csr, OP_ASSIGN, None, x_true)
    xs = np.zeros(n, dtype=dtype)
    info = {}
    st = smm.BiCGStab(A, b, xs, 5, dtype(1e-30), M, info=info)
    st_o, x_o, it_o, _ = oracle.bicgstab(csr, b, np.zeros(n, dtype=dtype), 5, dtype(1e-30), PRECOND_JACOBI, oracle.jacobi_setup(csr)[1])
    assert int(st) == st_o and info["iterations"] == it_o == 5
    assert np.abs(xs - x_o).max() <= 3e-4 * np.abs(x_o).max()


def test_pattern_rectangular_and_empty_rows(smm, oracle):
    """offsets relative to the row also describe a rectangular band; empty rows have an empty mask"""
    rows, cols = 5000, 5600
    rng = np.random.default_rng(9)
    offs = np.array([0, 3, 17, 250, 599])
    keep = rng.random((rows, len(offs))) < 0.7
    keep[::97] = False  # empty rows
    start = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(keep.sum(axis=1), out=start[1:])
    r, j = np.nonzero(keep)
    pos = (r + offs[j]).astype(np.int32)
    val = rng.uniform(-1, 1, len(pos))
    csr = (start, pos, val)
    A = smm.CSRMatrix(rows, cols, *csr)
    x = rng.uniform(-1, 1, cols)
    lhs = rng.uniform(-1, 1, rows)
    ref = oracle.spmv(csr, OP_SUB, lhs, x)
    for lanes in (1, 2):
        A.set_kernel(PATTERN, lanes)
        out = np.zeros(rows)
        A.rMultSub(lhs, x, out)
        if lanes == 1:
            np.testing.assert_array_equal(out, ref)
        else:
            np.testing.assert_allclose(out, ref, rtol=0, atol=1e-14)


def test_auto_selects_pattern_for_a_large_banded_matrix_and_keeps_the_bits(smm, oracle):
    """AUTO: the first SpMV of a matrix with >= 2^25 stored entries analyses it on the device (offsets from a sample
    of rows, every entry verified) and switches it to the index-free family; results are those of STREAM at the same lanes bit for bit,
    and the oracle's within the re-ordering bound"""
    import torch

    dev = torch.device("cuda:0")
    n, k, dtype = 700_000, 25, np.float32
    nnz = smm.host.gen_banded_nnz(n, k, 0x5EED, 1 << 16)
    assert nnz >= 1 << 25
    d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
    d_val = torch.empty(nnz, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    smm.host.gen_banded_dev(n, k, 0x5EED, 1 << 16, d_start, d_pos, d_val, dtype, stream)
    A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, dtype)
    assert A.get_kernel() == (STREAM, 2)  # before the first SpMV: the heuristic's choice from nnz / row
    x = torch.rand(n, dtype=torch.float32, device=dev) - 0.5
    lhs = torch.rand(n, dtype=torch.float32, device=dev) - 0.5
    y_auto = torch.empty(n, dtype=torch.float32, device=dev)
    A.spmv_dev(OP_SUB, lhs, x, y_auto, stream)
    torch.cuda.synchronize()
    assert A.get_kernel() == (PATTERN, 2)
    y_again = torch.empty_like(y_auto)
    A.spmv_dev(OP_SUB, lhs, x, y_again, stream)
    A.set_kernel(STREAM, 2)
    y_stream = torch.empty_like(y_auto)
    A.spmv_dev(OP_SUB, lhs, x, y_stream, stream)
    torch.cuda.synchronize()
    assert torch.equal(y_auto, y_again)
    body = (d_start[1:] <= nnz - 8200).cpu().numpy()  # (the directly streamed last tiles differ in how a row is cut into lanes)
    np.testing.assert_array_equal(y_auto.cpu().numpy()[body], y_stream.cpu().numpy()[body])
    A.set_kernel(smm.SPMV_AUTO, 0)
    assert A.get_kernel() == (PATTERN, 2)  # analysed and verified already: AUTO's choice stands
    csr = (d_start.cpu().numpy(), d_pos.cpu().numpy(), d_val.cpu().numpy())
    ref = oracle.spmv(csr, OP_SUB, lhs.cpu().numpy(), x.cpu().numpy())
    mag = np.zeros(n)
    np.add.at(mag, np.repeat(np.arange(n), np.diff(csr[0])), np.abs(csr[2].astype(np.float64) * x.cpu().numpy()[csr[1]]))
    assert np.all(np.abs(y_auto.cpu().numpy().astype(np.float64) - ref) <= 64 * np.finfo(dtype).eps * (mag + np.abs(lhs.cpu().numpy())))
    # the Jacobi fold (divide epilogue) exists in this family too: BiCGStab + Jacobi, 5 passes, against the oracle
    from oracle.oracle import PRECOND_JACOBI

    M = A.getPreconditioner(smm.SolverPreconditioner.JACOBI)
    x_true = np.random.default_rng(5).uniform(0.5, 1.5, n).astype(dtype)
    b = oracle.spmv(csr, OP_ASSIGN, None, x_true)
    xs = np.zeros(n, dtype=dtype)
    info = {}
    st = smm.BiCGStab(A, b, xs, 5, dtype(1e-30), M, info=info)
    st_o, x_o, it_o, _ = oracle.bicgstab(csr, b, np.zeros(n, dtype=dtype), 5, dtype(1e-30), PRECOND_JACOBI, oracle.jacobi_setup(csr)[1])
    assert int(st) == st_o and info["iterations"] == it_o == 5
    assert np.abs(xs - x_o).max() <= 3e-4 * np.abs(x_o).max()


CONST = 3


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_constant_diagonals(smm, oracle, dtype):
    """every diagonal holds one value (the Laplacians, constant-coefficient convection-diffusion): the CONST encoding reads no values[];
    bit for bit the oracle's numbers with one lane per row (ref:1484-1499), STREAM's at more lanes (served by the mask kernels); a single
    deviating value in a row the sampling does not look at sends the matrix back to the masks"""
    rng = np.random.default_rng(31)
    start, pos, val = gen.poisson2d(300, dtype=dtype)
    odd = val.copy()
    row = 44_441
    odd[start[row] + 1] = dtype(-1.0000001) if dtype == np.float32 else -1.0000000000001  # one entry of one diagonal, one ulp-ish off
    cases = {
        "poisson2d_200": (gen.poisson2d(200, dtype=dtype), CONST),
        "poisson2d_37x23": (gen.poisson2d(37, 23, dtype=dtype), CONST),
        "stencil3d_40x9x21": (gen.stencil3d(40, 9, 21, 6.0, -1.25, -0.75, dtype=dtype), CONST),
        "convdiff3d_17": (gen.convdiff3d(17, 0.3, dtype=dtype), CONST),
        "banded_random_values": (gen.banded_random_spd(20_000, 12, 0x5EED, 4096, dtype=dtype), MASKS),
        "poisson2d_300_one_value_off": ((start, pos, odd), MASKS),
    }
    for name, (csr, want_enc) in cases.items():
        rows = len(csr[0]) - 1
        A = smm.CSRMatrix(rows, rows, *csr)
        x = rng.uniform(-1, 1, rows).astype(dtype)
        lhs = rng.uniform(-1, 1, rows).astype(dtype)
        A.set_kernel(PATTERN, 1)
        assert A.pattern_info()[0] == want_enc, name
        for op in (OP_ASSIGN, OP_ADD, OP_SUB):
            ref = oracle.spmv(csr, op, lhs, x)
            np.testing.assert_array_equal(run(A, op, lhs, x, rows, dtype), ref, err_msg=f"{name} op {op}")
        if want_enc != CONST:
            continue
        # the same matrix on the mask kernel (values[] read): the same bits
        A.pattern_allow_const(False)
        assert A.pattern_info()[0] == MASKS
        np.testing.assert_array_equal(run(A, OP_SUB, lhs, x, rows, dtype), oracle.spmv(csr, OP_SUB, lhs, x))
        A.pattern_allow_const(True)
        # two lanes per row: STREAM's bits (before both families' directly streamed tails)
        A.set_kernel(STREAM, 2)
        want = run(A, OP_ADD, lhs, x, rows, dtype)
        A.set_kernel(PATTERN, 2)
        got = run(A, OP_ADD, lhs, x, rows, dtype)
        body = csr[0][1:] <= csr[0][-1] - 8200
        np.testing.assert_array_equal(got[body], want[body], err_msg=name)
    # inside the solvers (Jacobi folded into the rows: the divide epilogue; CG: the fused dots) against the oracle
    from oracle.oracle import PRECOND_JACOBI

    csr = gen.convdiff3d(17, 0.3, dtype=dtype)
    rows = len(csr[0]) - 1
    A = smm.CSRMatrix(rows, rows, *csr)
    A.set_kernel(PATTERN, 1)
    assert A.pattern_info()[0] == CONST
    b = oracle.spmv(csr, OP_ASSIGN, None, rng.uniform(0.5, 1.5, rows).astype(dtype))
    xs = np.zeros(rows, dtype=dtype)
    info = {}
    st = smm.BiCGStab(A, b, xs, 4, dtype(1e-30), A.getPreconditioner(smm.SolverPreconditioner.JACOBI), info=info)
    st_o, x_o, it_o, _ = oracle.bicgstab(csr, b, np.zeros(rows, dtype=dtype), 4, dtype(1e-30), PRECOND_JACOBI, oracle.jacobi_setup(csr)[1])
    assert int(st) == st_o and info["iterations"] == it_o == 4
    assert np.abs(xs - x_o).max() <= (3e-4 if dtype == np.float32 else 1e-10) * np.abs(x_o).max()


def test_constant_diagonals_at_scale(smm, oracle):
    """4.8 M rows of a 3-D stencil (512 x 256 x 37; AUTO: >= 2^25 entries): CONST chosen on the first SpMV, the reference's bits, the
    fused dot products finished inside the launch"""
    import torch

    dev = torch.device("cuda:0")
    nx, ny, nz, dtype = 512, 256, 37, np.float64
    n = nx * ny * nz
    nnz = smm.host.gen_stencil3d_nnz(nx, ny, nz)
    d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
    d_val = torch.empty(nnz, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    smm.host.gen_stencil3d_dev(nx, ny, nz, 6.0, -1.25, -0.75, d_start, d_pos, d_val, dtype, stream)
    A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, dtype)
    x = torch.rand(n, dtype=torch.float64, device=dev) - 0.5
    lhs = torch.rand(n, dtype=torch.float64, device=dev) - 0.5
    y = torch.empty(n, dtype=torch.float64, device=dev)
    A.spmv_dev(OP_SUB, lhs, x, y, stream)
    torch.cuda.synchronize()
    assert A.get_kernel() == (PATTERN, 1) and A.pattern_info() == (CONST, 7)
    csr = (d_start.cpu().numpy(), d_pos.cpu().numpy(), d_val.cpu().numpy())
    np.testing.assert_array_equal(y.cpu().numpy(), oracle.spmv(csr, OP_SUB, lhs.cpu().numpy(), x.cpu().numpy()))
    fin = torch.zeros(smm.host.finish_len(), dtype=torch.float64, device=dev)
    A.spmv_fused_dev(OP_ASSIGN, None, x, y, 2, x, fin, stream, finish=True)
    torch.cuda.synchronize()
    ref = oracle.spmv(csr, OP_ASSIGN, None, x.cpu().numpy())
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    off = smm.host.finish_totals_offset()
    totals = fin.cpu().numpy()
    xs = x.cpu().numpy()
    assert abs(totals[off] - float(ref @ ref)) <= 1e-10 * float(ref @ ref)
    assert abs(totals[off + 1] - float(ref @ xs)) <= 1e-10 * float(np.abs(ref * xs).sum())
    # the same matrix with values[] read (what a stencil with varying coefficients gets): the wave-private MASKS kernel, same bits
    A.pattern_allow_const(False)
    assert A.pattern_info() == (MASKS, 7)
    A.spmv_dev(OP_SUB, lhs, x, y, stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y.cpu().numpy(), oracle.spmv(csr, OP_SUB, lhs.cpu().numpy(), x.cpu().numpy()))
    fin.zero_()
    A.spmv_fused_dev(OP_ASSIGN, None, x, y, 2, x, fin, stream, finish=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    totals = fin.cpu().numpy()
    assert abs(totals[off] - float(ref @ ref)) <= 1e-10 * float(ref @ ref)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_wave_kernel_16_at_scale(smm, oracle, dtype):
    """spmvPatternWaveKernel<T, 16> (rows of 9-16 entries at one lane per row) on 2.2 M rows x 13 diagonals with varying values: many
    groups per wavefront, the XCD chunk walk, the ragged first / last rows -- rMult, rMultSub, rMultAdd in place and the fused dot
    products finished inside the launch, bit for bit against the oracle (one lane per row = the reference's order, ref:1484-1499)"""
    import torch

    dev = torch.device("cuda:0")
    td = torch.float32 if dtype == np.float32 else torch.float64
    n, k, seed, max_off = 2_200_000, 6, 0xBEEF, 1 << 17
    stream = torch.cuda.current_stream().cuda_stream
    nnz = smm.host.gen_banded_nnz(n, k, seed, max_off)
    d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
    d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
    d_val = torch.empty(nnz, dtype=td, device=dev)
    smm.host.gen_banded_dev(n, k, seed, max_off, d_start, d_pos, d_val, dtype, stream)
    A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, dtype)
    A.set_kernel(PATTERN, 1)
    assert A.pattern_info() == (MASKS, 2 * k + 1)
    assert A.kernel_desc()[0] == "spmvPatternWaveKernel"
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.rand(n, dtype=td, device=dev, generator=g) - 0.5
    lhs = torch.rand(n, dtype=td, device=dev, generator=g) - 0.5
    y = torch.empty(n, dtype=td, device=dev)
    csr = (d_start.cpu().numpy(), d_pos.cpu().numpy(), d_val.cpu().numpy())
    xh, lh = x.cpu().numpy(), lhs.cpu().numpy()
    for op in (OP_ASSIGN, OP_SUB):
        y.fill_(float("nan"))
        A.spmv_dev(op, lhs if op != OP_ASSIGN else None, x, y, stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(y.cpu().numpy(), oracle.spmv(csr, op, lh, xh))
    z = lhs.clone()
    A.spmv_dev(OP_ADD, z, x, z, stream)  # in place
    torch.cuda.synchronize()
    np.testing.assert_array_equal(z.cpu().numpy(), oracle.spmv(csr, OP_ADD, lh, xh))
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
    # ... and the same bits as the STREAM family at one lane per row
    A.set_kernel(STREAM, 1)
    A.spmv_dev(OP_ASSIGN, None, x, y, stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(y.cpu().numpy(), ref)
    A.close()


def test_auto_keeps_stream_for_a_large_matrix_without_a_pattern(smm):
    """i.i.d. columns (SURVEY.md section 8d's secondary matrix): far more than 64 offsets -- AUTO's attempt is refused quietly, the
    matrix stays with STREAM and the result is right"""
    import torch

    dev = torch.device("cuda:0")
    n, width = 700_000, 50
    g = torch.Generator(device=dev).manual_seed(11)
    cols = torch.sort(torch.randint(0, n, (n, width), device=dev, generator=g, dtype=torch.int64), dim=1).values
    d_pos = cols.reshape(-1).to(torch.int32).contiguous()
    d_start = (torch.arange(n + 1, device=dev, dtype=torch.int64) * width).to(torch.int32)
    d_val = torch.rand(n * width, dtype=torch.float32, device=dev, generator=g) - 0.5
    assert n * width >= 1 << 25
    A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np.float32)
    x = torch.rand(n, dtype=torch.float32, device=dev, generator=g) - 0.5
    y = torch.empty(n, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    A.spmv_dev(OP_ASSIGN, None, x, y, stream)
    torch.cuda.synchronize()
    assert A.get_kernel()[0] == STREAM
    want = (d_val.reshape(n, width).double() * x[cols].double()).sum(dim=1)
    assert float((y.double() - want).abs().max()) <= 64 * np.finfo(np.float32).eps * float((d_val.reshape(n, width).abs() * x[cols].abs()).sum(dim=1).max())
    with pytest.raises(smm.SmmHipError):
        A.set_kernel(PATTERN, 0)
