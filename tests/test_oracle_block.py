"""CPU tests of the oracle's block preconditioners (oracle/smm_oracle_impl.inc, smm_oracle_block_*): pinned against the committed outputs
of the real reference for the construction that defines them (tests/golden/reference_outputs_v3.npz, oracle/gen_golden_v3.py: the
reference's SGSPreconditioner of the block-diagonal part of A, plugged into the reference's BiCGStab template) and, in the build
container, against the real reference live.  BLOCK_ILU0 has no reference behaviour (like ILU0): it is checked against the global
textbook ILU0 on the block-diagonal matrix and by (L U)_ij = A_ij inside every block."""
import numpy as np
import pytest

from oracle.gen_golden_v3 import bounds_sets, matrices, rhs_of
from oracle.oracle import PRECOND_BLOCK_ILU0, PRECOND_BLOCK_SGS, PRECOND_ILU0, PRECOND_SGS, block_diagonal_part
from sparse_matrix_math_amd import generators as gen

DTYPES = [np.float32, np.float64]


@pytest.fixture(scope="module")
def golden_v3():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs_v3.npz"))


@pytest.mark.parametrize("dtype", DTYPES)
def test_block_sgs_matches_committed_reference_outputs(oracle, golden_v3, dtype):
    dn = np.dtype(dtype).name
    for mname, csr in matrices(dtype).items():
        rows = len(csr[0]) - 1
        rhs = rhs_of(rows, dtype)
        b = gen.row_sums(csr[0], csr[2])
        for bname, bounds in bounds_sets(rows).items():
            tag = f"block_sgs/{mname}/{bname}/{dn}"
            err, x = oracle.block_sgs_apply(csr, bounds, rhs)
            assert err == 0
            np.testing.assert_array_equal(x, golden_v3[f"{tag}/apply/x"], err_msg=tag)
            for maxit in (1, 3, 10):
                st, xs, it, _ = oracle.bicgstab_block(csr, b, np.zeros(rows, dtype=dtype), maxit, dtype(1e-30), PRECOND_BLOCK_SGS, bounds)
                assert st == int(golden_v3[f"{tag}/bicgstab/{maxit}/status"]) and it == maxit
                np.testing.assert_array_equal(xs, golden_v3[f"{tag}/bicgstab/{maxit}/x"], err_msg=f"{tag} maxit {maxit}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_block_sgs_is_reference_sgs_of_block_diagonal(oracle, reference, dtype):
    """live, in the build container: the real reference on the block-diagonal part"""
    csr = gen.convdiff3d(9, 0.3, dtype=dtype)
    rows = len(csr[0]) - 1
    rhs = rhs_of(rows, dtype)
    b = gen.row_sums(csr[0], csr[2])
    for bounds in (np.array([0, rows]), np.append(np.arange(0, rows, 50), rows), np.array([0, 3, 4, 400, rows])):
        bounds = bounds.astype(np.int32)
        bd, _ = block_diagonal_part(csr, bounds)
        with reference.csr(csr) as a, reference.csr(bd) as m:
            err, x_ref = reference.sgs_apply(m, rhs)
            assert err == 0
            np.testing.assert_array_equal(oracle.block_sgs_apply(csr, bounds, rhs)[1], x_ref)
            st_ref, xs_ref = reference.bicgstab_sgs_of(a, m, b, np.zeros(rows, dtype=dtype), 7, dtype(1e-30))
            st, xs, it, _ = oracle.bicgstab_block(csr, b, np.zeros(rows, dtype=dtype), 7, dtype(1e-30), PRECOND_BLOCK_SGS, bounds)
            assert st == st_ref
            np.testing.assert_array_equal(xs, xs_ref)


def slow_level_cut(csr, bounds, cap):
    """the rule of smm_oracle_block_level_cut written once more, with sets: which (row, column) pairs M keeps, the level of every row"""
    start, pos, _ = csr
    kept, lo, up = set(), {}, {}
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        for i in range(b0, b1):  # forward sweep, rows ascending
            deps = [j for j in pos[start[i]:start[i + 1]] if b0 <= j < i and (cap <= 0 or lo[j] < cap - 1)]
            kept.update((i, j) for j in deps)
            lo[i] = 1 + max(lo[j] for j in deps) if deps else 0
        for i in range(b1 - 1, b0 - 1, -1):  # backward sweep, rows descending
            deps = [j for j in pos[start[i]:start[i + 1]] if i < j < b1 and (cap <= 0 or up[j] < cap - 1)]
            kept.update((i, j) for j in deps)
            up[i] = 1 + max(up[j] for j in deps) if deps else 0
            kept.add((i, i))
    return kept, lo, up


def test_level_cut_rule(oracle):
    """smm_oracle_block_level_cut against the restatement above; no cap = the block-diagonal part; every kept entry points to a row
    above the deepest level, every dropped in-block entry to a row ON it; and the levels of the cut matrix really stop at the cap"""
    for csr, bounds in (
        (gen.convdiff3d(9, 0.3, dtype=np.float64), np.array([0, 3, 4, 400, 729], dtype=np.int32)),
        (gen.poisson2d(20, dtype=np.float64), np.array([0, 400], dtype=np.int32)),
        (gen.banded_random_spd(600, k=6, seed=4, max_offset=30, dtype=np.float64), np.append(np.arange(0, 600, 128), 600).astype(np.int32)),
        (gen.random_rows(300, 300, 2, 12, seed=9, dtype=np.float64, diag_dominant=True), np.array([0, 150, 300], dtype=np.int32)),
    ):
        start, pos, _ = csr
        rows = len(start) - 1
        rowof = np.repeat(np.arange(rows), np.diff(start))
        keep0, deepest0 = oracle.block_level_cut(csr, bounds, 0)
        np.testing.assert_array_equal(keep0, block_diagonal_part(csr, bounds)[1])
        for cap in (2, 3, 5, 9, deepest0, deepest0 + 7):
            keep, deepest = oracle.block_level_cut(csr, bounds, cap)
            want, lo, up = slow_level_cut(csr, bounds, cap)
            assert set(zip(rowof[keep].tolist(), pos[keep].tolist())) == want
            assert deepest == 1 + max(max(lo.values()), max(up.values())) <= cap
            assert not (keep & ~keep0).any()
            dropped = keep0 & ~keep
            assert all((lo if j < i else up)[j] == cap - 1 for i, j in zip(rowof[dropped].tolist(), pos[dropped].tolist()))
            if cap >= deepest0:
                np.testing.assert_array_equal(keep, keep0)  # a cap the blocks never reach drops nothing
            # the cut matrix has the levels the rule promised: cutting it again with no cap changes nothing and reports the same depth
            mcsr = oracle.level_cut_matrix(csr, bounds, cap)[0]
            again, deepest2 = oracle.block_level_cut(mcsr, bounds, 0)
            assert again.all() and deepest2 == deepest


@pytest.mark.parametrize("dtype", DTYPES)
def test_level_cut_block_sgs_is_reference_sgs_of_the_cut_matrix(oracle, reference, dtype):
    """live, in the build container: block SGS with a level cut = the real reference's SGS of the matrix the cut leaves, apply and
    inside the reference's BiCGStab template"""
    csr = gen.convdiff3d(9, 0.3, dtype=dtype)
    rows = len(csr[0]) - 1
    rhs = rhs_of(rows, dtype)
    b = gen.row_sums(csr[0], csr[2])
    bounds = np.append(np.arange(0, rows, 200), rows).astype(np.int32)
    for cap in (2, 4, 16):
        mcsr = oracle.level_cut_matrix(csr, bounds, cap)[0]
        with reference.csr(csr) as a, reference.csr(mcsr) as m:
            err, x_ref = reference.sgs_apply(m, rhs)
            assert err == 0
            np.testing.assert_array_equal(oracle.block_sgs_apply(mcsr, bounds, rhs)[1], x_ref)
            st_ref, xs_ref = reference.bicgstab_sgs_of(a, m, b, np.zeros(rows, dtype=dtype), 7, dtype(1e-30))
            st, xs, it, _ = oracle.bicgstab_block_of(csr, mcsr, b, np.zeros(rows, dtype=dtype), 7, dtype(1e-30), PRECOND_BLOCK_SGS, bounds)
            assert st == st_ref
            np.testing.assert_array_equal(xs, xs_ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_block_forms_are_the_global_forms_on_the_block_diagonal(oracle, dtype):
    for mname, csr in matrices(dtype).items():
        rows = len(csr[0]) - 1
        rhs = rhs_of(rows, dtype)
        b = gen.row_sums(csr[0], csr[2])
        for bname, bounds in bounds_sets(rows).items():
            bd, keep = block_diagonal_part(csr, bounds)
            e1, lu = oracle.block_ilu0_factorize(csr, bounds)
            e2, lu_bd = oracle.ilu0_factorize(bd)
            assert e1 == 0 and e2 == 0
            np.testing.assert_array_equal(lu[keep], lu_bd)
            np.testing.assert_array_equal(lu[~keep], csr[2][~keep])  # couplings between blocks keep A's value and are never read
            np.testing.assert_array_equal(oracle.block_ilu0_apply(csr, bounds, lu, rhs)[1], oracle.ilu0_apply(bd, lu_bd, rhs)[1])
            np.testing.assert_array_equal(oracle.block_sgs_apply(csr, bounds, rhs)[1], oracle.sgs_apply(bd, rhs)[1])
            if bname == "one":  # one block = the global preconditioner, also inside the solver
                for kind_b, kind_g, pv in ((PRECOND_BLOCK_ILU0, PRECOND_ILU0, lu), (PRECOND_BLOCK_SGS, PRECOND_SGS, None)):
                    got = oracle.bicgstab_block(csr, b, np.zeros(rows, dtype=dtype), 5, dtype(1e-30), kind_b, bounds, pv)
                    want = oracle.bicgstab(csr, b, np.zeros(rows, dtype=dtype), 5, dtype(1e-30), kind_g, pv)
                    assert got[0] == want[0] and got[2] == want[2]
                    np.testing.assert_array_equal(got[1], want[1])


def test_block_ilu0_reproduces_a_on_the_pattern_of_every_block(oracle):
    """(L U)_ij = A_ij for every stored entry inside a block: the defining property of ILU(0)"""
    csr = gen.convdiff3d(8, 0.3, dtype=np.float64)
    start, pos, val = csr
    rows = len(start) - 1
    bounds = np.array([0, 100, 101, 300, rows], dtype=np.int32)
    err, lu = oracle.block_ilu0_factorize(csr, bounds)
    assert err == 0
    L = np.eye(rows)
    U = np.zeros((rows, rows))
    inblock = np.zeros((rows, rows), dtype=bool)
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        inblock[b0:b1, b0:b1] = True
    A = np.zeros((rows, rows))
    for i in range(rows):
        for k in range(start[i], start[i + 1]):
            j = pos[k]
            if not inblock[i, j]:
                continue
            A[i, j] = val[k]
            if j < i:
                L[i, j] = lu[k]
            else:
                U[i, j] = lu[k]
    LU = L @ U
    mask = A != 0
    np.testing.assert_allclose(LU[mask], A[mask], rtol=1e-12, atol=1e-12)


def test_block_preconditioners_reject_bad_input(oracle):
    csr = gen.poisson2d(6, dtype=np.float64)
    rows = len(csr[0]) - 1
    rhs = np.ones(rows)
    for bad in (np.array([1, rows]), np.array([0, 10, 10, rows]), np.array([0, rows - 1])):
        assert oracle.block_sgs_apply(csr, bad.astype(np.int32), rhs)[0] != 0
        assert oracle.block_ilu0_factorize(csr, bad.astype(np.int32))[0] != 0
    start, pos, val = csr
    keep = pos != np.repeat(np.arange(rows), np.diff(start))  # strip the diagonal
    keep[: start[5]] = True
    s2 = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(np.bincount(np.repeat(np.arange(rows), np.diff(start))[keep], minlength=rows), out=s2[1:])
    nodiag = (s2, pos[keep].copy(), val[keep].copy())
    assert oracle.block_sgs_apply(nodiag, np.array([0, 12, rows], dtype=np.int32), rhs)[0] != 0
    assert oracle.block_ilu0_factorize(nodiag, np.array([0, 12, rows], dtype=np.int32))[0] == 2
