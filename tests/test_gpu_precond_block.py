"""GPU parity of the block preconditioners (csrc/smm_precond_block.hip: one wavefront per block, the block's vector in LDS):
apply and factor bit-identical to the oracle's sequential sweeps over the matrix M is built from -- the block-diagonal part of A
without the entries the level cut drops (smm_oracle_block_level_cut; level_cap 0: the block-diagonal part itself) -- BLOCK_SGS
without a cut bit-identical to what the REAL reference computes for that construction (tests/golden/reference_outputs_v3.npz),
BiCGStab with them within the solver tolerance of the oracle / the reference at fixed iterations."""
import os

import numpy as np
import pytest
from test_oracle_block import golden_v3  # noqa: F401  (fixture)

from oracle.gen_golden_v3 import bounds_sets, matrices, rhs_of
from oracle.oracle import PRECOND_BLOCK_ILU0, PRECOND_BLOCK_SGS
from sparse_matrix_math_amd import generators as gen

pytestmark = pytest.mark.gpu
DTYPES = [np.float32, np.float64]
AUTO, CONTIGUOUS, BRICKS = 0, 1, 2
TOL = {np.dtype(np.float32): 3e-4, np.dtype(np.float64): 1e-10}


def make(smm, csr):
    rows = len(csr[0]) - 1
    return smm.CSRMatrix(rows, rows, *csr)


def check_bounds(bounds, rows, block_rows, csr):
    assert bounds[0] == 0 and bounds[-1] == rows
    sizes = np.diff(bounds)
    assert (sizes >= 1).all() and sizes.max() <= block_rows
    nnz = csr[0][bounds[1:]] - csr[0][bounds[:-1]]
    assert (nnz[sizes > 1] <= 8192).all()


def permuted(csr, order):
    """(P A P^T as sorted CSR, for every stored entry of P A P^T the index of the same entry in A's arrays): row p of the result is row
    order[p] of A, columns renumbered by the inverse of `order`.  The block kinds with a brick partition are, by definition, the block
    kinds of this matrix with contiguous blocks."""
    import scipy.sparse as sp

    start, pos, val = csr
    n = len(start) - 1
    tag = sp.csr_matrix((np.arange(1, len(pos) + 1, dtype=np.float64), pos, start), shape=(n, n))  # entry k of A carries k + 1
    tp = tag[order][:, order].tocsr()
    tp.sort_indices()
    src = tp.data.astype(np.int64) - 1
    return (tp.indptr.astype(np.int32), tp.indices.astype(np.int32), val[src].copy()), src


def compare_with_oracle(smm, oracle, csr, block_rows, dtype, seed=5, level_cap=None, partition=None):
    """factor + apply of both kinds against the oracle on the library's own partition; returns (bounds, row order, the matrix M is built
    from -- in the order of the partition)"""
    P = smm.SolverPreconditioner
    rows = len(csr[0]) - 1
    A = make(smm, csr)
    rhs = np.random.default_rng(seed).uniform(-1, 1, rows).astype(dtype)
    I = A.getPreconditioner(P.BLOCK_ILU0, block_rows, level_cap, partition)
    cap = I.level_cap()
    assert cap == (16 if level_cap is None else level_cap)
    bounds = I.block_bounds()
    order, brick = I.block_rows()
    assert sorted(order.tolist()) == list(range(rows))
    contiguous = np.array_equal(order, np.arange(rows))
    assert contiguous == (brick == (0, 0, 0))
    for b0, b1 in zip(bounds[:-1], bounds[1:]):  # inside a block the rows keep their natural order
        assert np.all(np.diff(order[b0:b1]) > 0)
    pcsr, src = permuted(csr, order)
    check_bounds(bounds, rows, block_rows or 1024, pcsr)
    mcsr, keep, deepest = oracle.level_cut_matrix(pcsr, bounds, cap)
    assert max(I.levels()) == deepest and (cap == 0 or deepest <= cap)
    err, lu = oracle.block_ilu0_factorize(mcsr, bounds)
    assert err == 0
    # the factor lives on A's pattern: entries M does not hold (other blocks, dropped by the cut) keep A's value
    dev_lu = I.values()[src]  # ... in the order of P A P^T
    np.testing.assert_array_equal(dev_lu[keep], lu)
    np.testing.assert_array_equal(dev_lu[~keep], pcsr[2][~keep])
    x = np.zeros(rows, dtype=dtype)
    assert I.apply(rhs, x) == 0
    np.testing.assert_array_equal(x[order], oracle.block_ilu0_apply(mcsr, bounds, lu, rhs[order])[1])
    S = A.getPreconditioner(P.BLOCK_SGS, block_rows, level_cap, partition)
    np.testing.assert_array_equal(S.block_bounds(), bounds)
    np.testing.assert_array_equal(S.block_rows()[0], order)
    assert S.levels() == I.levels()
    x2 = np.zeros(rows, dtype=dtype)
    assert S.apply(rhs, x2) == 0
    np.testing.assert_array_equal(x2[order], oracle.block_sgs_apply(mcsr, bounds, rhs[order])[1])
    return bounds, order, mcsr


@pytest.mark.parametrize("dtype", DTYPES)
def test_block_apply_and_factor_bit_identical_to_oracle(smm, oracle, dtype):
    cases = [
        (gen.convdiff3d(24, 0.3, dtype=dtype), None),
        (gen.convdiff3d(24, 0.3, dtype=dtype), 64),
        (gen.convdiff3d(14, 0.3, dtype=dtype), 2048),
        (gen.poisson2d(70, dtype=dtype), 256),
        (gen.banded_random_spd(3000, k=3, seed=3, max_offset=40, dtype=dtype), 128),     # up to 3 + 3 in-block entries: 4 per record
        (gen.banded_random_spd(3000, k=7, seed=5, max_offset=60, dtype=dtype), 512),     # up to 7 + 7: 8 per record
        (gen.banded_random_spd(4000, k=12, seed=11, max_offset=90, dtype=dtype), None),  # 12 + 12: records + overflow lists
        (gen.random_rows(900, 900, 2, 25, seed=9, dtype=dtype, diag_dominant=True), 300),
        (gen.poisson2d(5, dtype=dtype), None),  # 25 rows: a single partly filled chunk
    ]
    for csr, block_rows in cases:
        compare_with_oracle(smm, oracle, csr, block_rows, dtype)  # the defaults: level cut 16, bricks where the matrix is a grid stencil
        compare_with_oracle(smm, oracle, csr, block_rows, dtype, level_cap=0)  # no cut: the block-diagonal part itself
        compare_with_oracle(smm, oracle, csr, block_rows, dtype, partition=CONTIGUOUS)


@pytest.mark.parametrize("dtype", DTYPES)
def test_spmv_inside_the_apply_bit_identical_to_oracle(smm, oracle, dtype):
    """x = M^-1 (A v) in ONE launch (smm_hip_precond_apply_spmv: what BiCGStab's loop asks for twice per pass, ref:2234-2235,
    2250-2251): every row of A v summed in the order of its stored entries (the oracle's rMult, ref:1484-1489), then the oracle's
    sweeps -- bit for bit, on bricks and contiguous blocks, rows shorter and longer than the 8 entries requested at once, an empty row
    at the end, and the kinds that have no fused form (the SpMV at one lane per row, then the apply)"""
    P = smm.SolverPreconditioner
    long_rows = gen.random_rows(900, 900, 2, 25, seed=9, dtype=dtype, diag_dominant=True)
    cases = [
        (gen.convdiff3d(24, 0.3, dtype=dtype), None, None),
        (gen.convdiff3d(24, 0.3, dtype=dtype), 64, CONTIGUOUS),
        (gen.convdiff3d(14, 0.3, dtype=dtype), 2048, None),  # one block of 2048 rows: eight rows per lane
        (gen.poisson2d(70, dtype=dtype), 256, None),
        (gen.banded_random_spd(4000, k=12, seed=11, max_offset=90, dtype=dtype), None, None),  # 25-entry rows: the tail loop
        (long_rows, 300, None),
        (gen.poisson2d(5, dtype=dtype), None, None),
    ]
    for csr, block_rows, partition in cases:
        rows = len(csr[0]) - 1
        A = make(smm, csr)
        A.set_kernel(smm.SPMV_VECTOR, 1)  # (only the unfused kinds below run an SpMV kernel)
        v = np.random.default_rng(17).uniform(-1, 1, rows).astype(dtype)
        av = oracle.spmv(csr, 0, None, v)
        for kind in (P.BLOCK_ILU0, P.BLOCK_SGS):
            M = A.getPreconditioner(kind, block_rows, None, partition)
            bounds, order = M.block_bounds(), M.block_rows()[0]
            mcsr = oracle.level_cut_matrix(permuted(csr, order)[0], bounds, M.level_cap())[0]
            x = np.zeros(rows, dtype=dtype)
            assert M.apply_spmv(v, x) == 0
            if kind == P.BLOCK_ILU0:
                want = oracle.block_ilu0_apply(mcsr, bounds, oracle.block_ilu0_factorize(mcsr, bounds)[1], av[order])[1]
            else:
                want = oracle.block_sgs_apply(mcsr, bounds, av[order])[1]
            np.testing.assert_array_equal(x[order], want)
            x2 = np.zeros(rows, dtype=dtype)
            assert M.apply(av, x2) == 0  # the two-launch form on the same A v
            np.testing.assert_array_equal(x, x2)
            with pytest.raises(Exception):
                M.apply_spmv(x, x)
        J = A.getPreconditioner(P.JACOBI)
        x = np.zeros(rows, dtype=dtype)
        assert J.apply_spmv(v, x) == 0
        x2 = np.zeros(rows, dtype=dtype)
        assert J.apply(av, x2) == 0
        np.testing.assert_array_equal(x, x2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_spmv_inside_the_apply_from_the_constant_diagonal_encoding(smm, oracle, dtype):
    """a matrix the PATTERN analysis found constant diagonals in (what the solvers adopt from 2^20 entries; asked for here): the one
    launch forms A v from the row masks and the diagonals' values -- the same products in the same order, so the bits of the form that
    reads positions[] / values[] and of the oracle; bricks and contiguous blocks, grids the bricks do not divide, and a matrix whose
    diagonals are NOT constant (the masks alone: the launch takes the columns from the masks and reads values[])"""
    P = smm.SolverPreconditioner
    varying = gen.poisson2d(40, dtype=dtype)
    varying[2][::7] *= dtype(1.5)
    for csr, block_rows, partition, want_const in (
        (gen.convdiff3d(24, 0.3, dtype=dtype), None, None, True),
        (gen.convdiff3d(21, 0.3, dtype=dtype), 512, None, True),
        (gen.poisson2d(70, dtype=dtype), 256, CONTIGUOUS, True),
        (varying, None, None, False),
    ):
        rows = len(csr[0]) - 1
        v = np.random.default_rng(23).uniform(-1, 1, rows).astype(dtype)
        av = oracle.spmv(csr, 0, None, v)
        plain = make(smm, csr)
        plain.set_kernel(smm.SPMV_VECTOR, 1)
        A = make(smm, csr)
        A.set_kernel(smm.SPMV_PATTERN, 1)
        y = np.zeros(rows, dtype=dtype)
        A.rMult(v, y)  # (the analysis runs with the first SpMV of the family)
        np.testing.assert_array_equal(y, av)
        assert (A.pattern_info()[0] == 3) == want_const, A.pattern_info()
        for kind in (P.BLOCK_ILU0, P.BLOCK_SGS):
            M = A.getPreconditioner(kind, block_rows, None, partition)
            Mp = plain.getPreconditioner(kind, block_rows, None, partition)
            x, xp, x2 = (np.zeros(rows, dtype=dtype) for _ in range(3))
            assert M.apply_spmv(v, x) == 0 and Mp.apply_spmv(v, xp) == 0 and M.apply(av, x2) == 0
            np.testing.assert_array_equal(x, xp)
            np.testing.assert_array_equal(x, x2)
            bounds, order = M.block_bounds(), M.block_rows()[0]
            mcsr = oracle.level_cut_matrix(permuted(csr, order)[0], bounds, M.level_cap())[0]
            if kind == P.BLOCK_SGS:
                np.testing.assert_array_equal(x[order], oracle.block_sgs_apply(mcsr, bounds, av[order])[1])


@pytest.mark.parametrize("dtype", DTYPES)
def test_bicgstab_loop_with_the_spmv_inside_the_apply_is_the_two_launch_loop(smm, dtype, monkeypatch):
    """SMM_HIP_BLOCK_FUSE_SPMV=0 (read per solve) runs SpMV and apply as two launches; with the SpMV at one lane per row both loops do
    the same arithmetic on the same operands -- the dot products ride in the apply's epilogue either way -- so x agrees bit for bit;
    also with the matrix in the PATTERN family (the stencils: the one launch then reads their constant diagonals)"""
    P = smm.SolverPreconditioner
    for csr, block_rows in ((gen.convdiff3d(24, 0.3, dtype=dtype), None), (gen.poisson2d(70, dtype=dtype), 256),
                            (gen.banded_random_spd(4000, k=12, seed=11, max_offset=90, dtype=dtype), None)):
        rows = len(csr[0]) - 1
        b = gen.row_sums(csr[0], csr[2])
        for family in (smm.SPMV_VECTOR, smm.SPMV_PATTERN):
            A = make(smm, csr)
            A.set_kernel(family, 1)
            for kind in (P.BLOCK_ILU0, P.BLOCK_SGS):
                M = A.getPreconditioner(kind, block_rows)
                got = []
                for fuse in ("1", "0"):
                    monkeypatch.setenv("SMM_HIP_BLOCK_FUSE_SPMV", fuse)
                    x = np.zeros(rows, dtype=dtype)
                    info = {}
                    st = smm.BiCGStab(A, b, x, 6, dtype(1e-30), M, info=info)
                    got.append((int(st), info["iterations"], x))
                monkeypatch.delenv("SMM_HIP_BLOCK_FUSE_SPMV")
                assert got[0][:2] == got[1][:2]
                np.testing.assert_array_equal(got[0][2], got[1][2])


@pytest.mark.parametrize("dtype", DTYPES)
def test_level_cut_bounds_every_sweep(smm, oracle, dtype):
    """caps from 2 (every row keeps only entries that point to rows without kept entries of their own) upwards, on matrices whose
    uncut blocks are 30 to 200 levels deep; a tridiagonal matrix (a block is ONE chain, 512 levels) cut to 8"""
    tri = gen.banded_random_spd(5000, k=1, seed=2, max_offset=2, dtype=dtype)
    for csr, block_rows, caps in (
        (gen.convdiff3d(24, 0.3, dtype=dtype), None, (2, 3, 5, 8, 40)),
        (gen.poisson2d(70, dtype=dtype), 256, (2, 7, 64)),
        (gen.banded_random_spd(4000, k=12, seed=11, max_offset=90, dtype=dtype), None, (2, 4, 33)),
        (tri, 512, (2, 8, 4095)),
    ):
        for cap in caps:
            compare_with_oracle(smm, oracle, csr, block_rows, dtype, level_cap=cap)
    with pytest.raises(Exception):
        make(smm, tri).getPreconditioner(smm.SolverPreconditioner.BLOCK_ILU0, 512, 1)  # 1 would drop every coupling: refused
    with pytest.raises(Exception):
        make(smm, tri).getPreconditioner(smm.SolverPreconditioner.BLOCK_ILU0, 512, 4096)


def test_brick_partition(smm, oracle):
    """grid stencils get bricks of the grid as blocks (16 x 8 x 8 points for 1024 rows, squares in 2-D), also on grids the bricks do not
    divide; every row in exactly one block, natural order inside (checked by compare_with_oracle); other matrices keep the contiguous
    cut, and asking for bricks there is refused; BiCGStab needs fewer iterations with bricks than with runs of consecutive rows"""
    P = smm.SolverPreconditioner
    dtype = np.float64
    for csr, dims, block_rows, want in (
        (gen.convdiff3d(24, 0.3, dtype=dtype), (24, 24, 24), None, (16, 8, 8)),
        (gen.stencil3d(40, 9, 21, 6.0, -1.25, -0.75, dtype=dtype), (40, 9, 21), None, (16, 8, 8)),
        (gen.stencil3d(13, 30, 11, 6.0, -1.0, -1.0, dtype=dtype), (13, 30, 11), 256, (4, 8, 8)),
        (gen.poisson2d(70, dtype=dtype), (70, 70, 1), None, (32, 32, 1)),
        (gen.poisson2d(37, 23, dtype=dtype), (37, 23, 1), 256, (16, 16, 1)),
    ):
        bounds, order, _ = compare_with_oracle(smm, oracle, csr, block_rows, dtype)
        rows = len(csr[0]) - 1
        I = make(smm, csr).getPreconditioner(P.BLOCK_ILU0, block_rows)
        brick = I.block_rows()[1]
        assert brick == tuple(min(w, d) for w, d in zip(want, dims)), (dims, brick)
        nx, ny, _ = dims
        ix, iy, iz = order % nx, (order // nx) % ny, order // (nx * ny)
        for b0, b1 in zip(bounds[:-1], bounds[1:]):  # a block is one brick: its points agree in (ix // bx, iy // by, iz // bz)
            assert len({(int(a), int(b), int(c)) for a, b, c in zip(ix[b0:b1] // brick[0], iy[b0:b1] // brick[1], iz[b0:b1] // brick[2])}) == 1
        compare_with_oracle(smm, oracle, csr, block_rows, dtype, partition=BRICKS)
    banded = gen.banded_random_spd(3000, k=7, seed=5, max_offset=60, dtype=dtype)
    assert make(smm, banded).getPreconditioner(P.BLOCK_ILU0).block_rows()[1] == (0, 0, 0)
    with pytest.raises(Exception):
        make(smm, banded).getPreconditioner(P.BLOCK_ILU0, None, None, BRICKS)
    csr = gen.convdiff3d(40, 0.3, dtype=dtype)
    rows = len(csr[0]) - 1
    A = make(smm, csr)
    b = gen.row_sums(csr[0], csr[2])
    its = {}
    for name, part in (("bricks", AUTO), ("contiguous", CONTIGUOUS)):
        x = np.zeros(rows)
        info = {}
        st = smm.BiCGStab(A, b, x, -1, 1e-9, A.getPreconditioner(P.BLOCK_ILU0, None, None, part), info=info)
        assert int(st) == 0
        np.testing.assert_allclose(x, np.ones(rows), atol=1e-6)
        its[name] = info["iterations"]
    assert its["bricks"] < its["contiguous"], its


@pytest.mark.parametrize("dtype", DTYPES)
def test_block_sgs_bit_identical_to_the_reference_on_the_block_diagonal(smm, golden_v3, dtype):  # noqa: F811
    """uniform cuts that the library's greedy cut reproduces (rows only: these blocks stay far below 8192 entries)"""
    P = smm.SolverPreconditioner
    dn = np.dtype(dtype).name
    for mname, csr in matrices(dtype).items():
        rows = len(csr[0]) - 1
        A = make(smm, csr)
        rhs = rhs_of(rows, dtype)
        b = gen.row_sums(csr[0], csr[2])
        for bname, block_rows in (("u64", 64), ("u256", 256)) + ((("one", 2048),) if rows <= 2048 and csr[0][-1] <= 8192 else ()):
            if mname == "banded_2000" and bname == "u256":
                continue  # 256 rows x 51 entries exceed the 8192-entry cap: the library cuts those blocks shorter
            M = A.getPreconditioner(P.BLOCK_SGS, block_rows, 0, CONTIGUOUS)  # no level cut, contiguous blocks: M = the block-diagonal part the reference was run on
            np.testing.assert_array_equal(M.block_bounds(), bounds_sets(rows)[bname])
            tag = f"block_sgs/{mname}/{bname}/{dn}"
            x = np.zeros(rows, dtype=dtype)
            assert M.apply(rhs, x) == 0
            np.testing.assert_array_equal(x, golden_v3[f"{tag}/apply/x"], err_msg=tag)
            for maxit in (1, 3, 10):
                xs = np.zeros(rows, dtype=dtype)
                info = {}
                st = smm.BiCGStab(A, b, xs, maxit, dtype(1e-30), M, info=info)
                ref = golden_v3[f"{tag}/bicgstab/{maxit}/x"]
                assert int(st) == int(golden_v3[f"{tag}/bicgstab/{maxit}/status"]) and info["iterations"] == maxit
                assert np.abs(xs - ref).max() <= TOL[np.dtype(dtype)] * max(1.0, np.abs(ref).max()), (tag, maxit)


@pytest.mark.parametrize("dtype", DTYPES)
def test_one_block_is_the_global_preconditioner(smm, oracle, dtype):
    P = smm.SolverPreconditioner
    csr = gen.convdiff3d(12, 0.3, dtype=dtype)  # 1728 rows, 11 664 entries: over the entry cap -> two blocks; 10^3 fits one
    csr1 = gen.convdiff3d(10, 0.3, dtype=dtype)
    rows = len(csr1[0]) - 1
    A = make(smm, csr1)
    rhs = rhs_of(rows, dtype)
    for kind_b, kind_g in ((P.BLOCK_ILU0, P.ILU0), (P.BLOCK_SGS, P.SYMMETRIC_GAUS_SEIDEL)):
        B = A.getPreconditioner(kind_b, 2048, 0, CONTIGUOUS)
        np.testing.assert_array_equal(B.block_bounds(), [0, rows])
        G = A.getPreconditioner(kind_g)
        xb, xg = np.zeros(rows, dtype=dtype), np.zeros(rows, dtype=dtype)
        B.apply(rhs, xb)
        G.apply(rhs, xg)
        np.testing.assert_array_equal(xb, xg)
        assert B.levels() == G.levels()
    assert len(make(smm, csr).getPreconditioner(P.BLOCK_ILU0, 2048, None, CONTIGUOUS).block_bounds()) == 3


@pytest.mark.parametrize("dtype", DTYPES)
def test_bicgstab_with_block_preconditioners_matches_oracle(smm, oracle, dtype):
    P = smm.SolverPreconditioner
    tol = TOL[np.dtype(dtype)]
    for csr, block_rows in ((gen.convdiff3d(20, 0.3, dtype=dtype), None), (gen.poisson2d(60, dtype=dtype), 128)):
        rows = len(csr[0]) - 1
        A = make(smm, csr)
        b = gen.row_sums(csr[0], csr[2])
        for kind, okind in ((P.BLOCK_ILU0, PRECOND_BLOCK_ILU0), (P.BLOCK_SGS, PRECOND_BLOCK_SGS)):
            M = A.getPreconditioner(kind, block_rows)
            bounds = M.block_bounds()
            order = M.block_rows()[0]
            pcsr = permuted(csr, order)[0]  # the oracle solves the same system in the order of the partition: P A P^T (P x) = P b
            mcsr = oracle.level_cut_matrix(pcsr, bounds, M.level_cap())[0]
            pv = oracle.block_ilu0_factorize(mcsr, bounds)[1] if kind == P.BLOCK_ILU0 else None
            # (few iterations: with these strong preconditioners the fp32 solve reaches round-off within ~10 passes, after which the
            # iterates follow the summation order of the dot products, not the algorithm)
            for maxit in (1, 2, 4):
                x = np.zeros(rows, dtype=dtype)
                info = {}
                st = smm.BiCGStab(A, b, x, maxit, dtype(1e-30), M, info=info)
                st_o, x_o, it_o, _ = oracle.bicgstab_block_of(pcsr, mcsr, b[order], np.zeros(rows, dtype=dtype), maxit, dtype(1e-30), okind, bounds, pv)
                assert int(st) == st_o and info["iterations"] == it_o == maxit
                # (tolerance per pass: every pass multiplies the last-bit differences of the dot products, summed in another order)
                assert np.abs(x[order] - x_o).max() <= tol * maxit * max(1.0, np.abs(x_o).max()), (kind, maxit)
            # converged: the iteration count within 4 % (the dot products are summed in another order -- the oracle's in the order of the
            # partition), x = 1
            eps = dtype(1e-4 if dtype == np.float32 else 1e-9)
            x = np.zeros(rows, dtype=dtype)
            info = {}
            st = smm.BiCGStab(A, b, x, -1, eps, M, info=info)
            st_o, x_o, it_o, _ = oracle.bicgstab_block_of(pcsr, mcsr, b[order], np.zeros(rows, dtype=dtype), -1, eps, okind, bounds, pv)
            assert int(st) == st_o == 0 and abs(info["iterations"] - it_o) <= max(1, it_o // 25), (info, it_o)
            np.testing.assert_allclose(x, np.ones(rows), atol=50 * float(eps))


@pytest.mark.parametrize("n,dtype", [(108, np.float64), (1000, np.float64)])
def test_block_preconditioners_at_config_sizes(smm, oracle, n, dtype):
    """BASELINE config 5's stand-in (convection-diffusion 108^3) and configs 1 / 2's matrix (Poisson 1000^2) at full size: apply and
    factor bit-identical to the oracle, the preconditioned solve against the oracle's"""
    P = smm.SolverPreconditioner
    csr = gen.convdiff3d(108, 0.3, dtype=dtype) if n == 108 else gen.poisson2d(1000, dtype=dtype)
    rows = len(csr[0]) - 1
    bounds, order, mcsr = compare_with_oracle(smm, oracle, csr, None, dtype)
    pcsr = permuted(csr, order)[0]
    A = make(smm, csr)
    b = gen.row_sums(csr[0], csr[2])
    M = A.getPreconditioner(P.BLOCK_ILU0)
    lu = oracle.block_ilu0_factorize(mcsr, bounds)[1]
    x = np.zeros(rows, dtype=dtype)
    info = {}
    # (5 passes: BiCGStab's early iterates on these matrices swing far from the solution -- max |x| 13 after 20 passes at 108^3 -- and
    # every swing amplifies the last-bit differences of the dot products, which are summed in another order than the oracle's)
    st = smm.BiCGStab(A, b, x, 5, dtype(1e-30), M, info=info)
    st_o, x_o, it_o, _ = oracle.bicgstab_block_of(pcsr, mcsr, b[order], np.zeros(rows, dtype=dtype), 5, dtype(1e-30), PRECOND_BLOCK_ILU0, bounds, lu)
    assert int(st) == st_o and info["iterations"] == it_o == 5
    assert np.abs(x[order] - x_o).max() <= 1e-9 * max(1.0, np.abs(x_o).max())
    # and the converged solve: x = 1 and the oracle's iteration count within 2 % (108^3); the 2-D Poisson matrix needs ~1500 passes
    # with or without these (line-shaped) blocks, and over that many passes the count itself depends on the summation order of the
    # dot products: within 25 %
    x = np.zeros(rows, dtype=dtype)
    st = smm.BiCGStab(A, b, x, -1, dtype(1e-8), M, info=info)
    st_o, x_o, it_o, _ = oracle.bicgstab_block_of(pcsr, mcsr, b[order], np.zeros(rows, dtype=dtype), -1, dtype(1e-8), PRECOND_BLOCK_ILU0, bounds, lu)
    assert int(st) == st_o == 0 and abs(info["iterations"] - it_o) <= max(2, it_o // (50 if n == 108 else 4)), (info, it_o)
    np.testing.assert_allclose(x, np.ones(rows), atol=1e-6)


def test_creating_the_preconditioner_first_still_gives_the_solver_its_spmv_family(smm):
    """the brick partition analyses the matrix's offsets at create time; a solver that comes later finds the analysis done and must
    still move the matrix to the PATTERN family (>= 2^20 entries)"""
    P = smm.SolverPreconditioner
    csr = gen.convdiff3d(60, 0.3, dtype=np.float64)  # 1.5 M entries
    rows = len(csr[0]) - 1
    A = make(smm, csr)
    assert A.get_kernel()[0] == 2 and A.pattern_info()[0] == 0  # STREAM, not analysed
    M = A.getPreconditioner(P.BLOCK_ILU0)
    assert M.block_rows()[1] == (16, 8, 8)
    assert A.get_kernel() == (3, 1) and A.pattern_info() == (3, 7)  # PATTERN, one lane per row, constant diagonals, 7 offsets
    b = gen.row_sums(csr[0], csr[2])
    x = np.zeros(rows)
    assert int(smm.BiCGStab(A, b, x, -1, 1e-9, M)) == 0
    np.testing.assert_allclose(x, np.ones(rows), atol=1e-6)


def test_block_apply_is_repeatable_and_asynchronous(smm):
    """device-pointer apply on the caller's stream, twice: same bits; and the fused-dot epilogue leaves x unchanged"""
    torch = pytest.importorskip("torch")
    P = smm.SolverPreconditioner
    csr = gen.convdiff3d(30, 0.3, dtype=np.float64)
    rows = len(csr[0]) - 1
    A = make(smm, csr)
    M = A.getPreconditioner(P.BLOCK_ILU0)
    rhs = torch.rand(rows, dtype=torch.float64, device="cuda")
    x1 = torch.zeros_like(rhs)
    x2 = torch.zeros_like(rhs)
    s = torch.cuda.current_stream().cuda_stream
    M.apply_dev(rhs, x1, s)
    M.apply_dev(rhs, x2, s)
    torch.cuda.synchronize()
    assert torch.equal(x1, x2)
    host = np.zeros(rows)
    M.apply(rhs.cpu().numpy(), host)
    np.testing.assert_array_equal(host, x1.cpu().numpy())


def test_block_preconditioner_structural_failures(smm):
    P = smm.SolverPreconditioner
    start, pos, val = gen.poisson2d(12, dtype=np.float64)
    rows = len(start) - 1
    rowof = np.repeat(np.arange(rows), np.diff(start))
    keep = ~((pos == rowof) & (rowof == 77))  # row 77 loses its diagonal
    s2 = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(np.bincount(rowof[keep], minlength=rows), out=s2[1:])
    A = smm.CSRMatrix(rows, rows, s2, pos[keep].copy(), val[keep].copy())
    for kind in (P.BLOCK_ILU0, P.BLOCK_SGS):
        with pytest.raises(Exception):
            A.getPreconditioner(kind, 64)
    tiny = val.copy()
    tiny[(pos == rowof) & (rowof == 5)] = 1e-7  # |d| < 1e-5: SGS refuses (ref:1691-1693), ILU0's pivot is tiny as well
    A2 = smm.CSRMatrix(rows, rows, start, pos, tiny)
    with pytest.raises(Exception):
        A2.getPreconditioner(P.BLOCK_SGS, 64)
    with pytest.raises(Exception):
        A.getPreconditioner(P.BLOCK_ILU0, 32)  # block_rows out of range
    with pytest.raises(Exception):
        A.getPreconditioner(P.JACOBI, 64)  # not a block kind
    start, pos, val = gen.poisson2d(12, dtype=np.float64)
    good = smm.CSRMatrix(rows, rows, start, pos, val)
    with pytest.raises(Exception):
        good.getPreconditioner(P.BLOCK_ILU0, 64, None, 3)  # unknown partition
    with pytest.raises(Exception):
        good.getPreconditioner(P.BLOCK_ILU0, 64, -2)  # level cut out of range
    M = good.getPreconditioner(P.BLOCK_SGS, 64, 0, 2)  # bricks asked for, on a 12 x 12 grid: 8 x 8 squares
    assert M.block_rows()[1] == (8, 8, 1) and M.level_cap() == 0 and len(M.block_bounds()) == 5
