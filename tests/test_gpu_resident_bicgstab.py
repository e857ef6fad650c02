"""The single-launch BiCGStab (csrc/smm_resident_bicg.hip: vectors in registers, five grid barriers per iteration) against the loop of
seven launches (csrc/smm_solvers.hip) and the oracle (ref:2191-2283): constant diagonals and values read, with and without Jacobi, every
compiled rows-per-lane shape, the reference's quirks (maxIterations == 0, an exact start vector), and the fall-back."""
import numpy as np
import pytest

from oracle.oracle import PRECOND_JACOBI
from sparse_matrix_math_amd import generators as gen
from sparse_matrix_math_amd import host

pytestmark = pytest.mark.gpu
PATTERN = 3
# fixed iterations: the two GPU paths share every per-row operation and differ in the partition of the global sums only
PATHS_TOL = {np.float32: 2e-4, np.float64: 1e-11}
ORACLE_TOL = {np.float32: 3e-3, np.float64: 1e-9}


@pytest.fixture()
def modes(smm):
    before = host.bicgstab_resident(-1)
    yield
    host.bicgstab_resident(before)


def _solve(smm, A, b, maxit, eps, M, mode, x0=None):
    host.bicgstab_resident(mode)
    x = np.zeros(len(b), dtype=b.dtype) if x0 is None else x0.copy()
    info = {}
    st = smm.BiCGStab(A, b.copy(), x, maxit, eps, M, info=info)
    return int(st), info["iterations"], x.astype(np.float64), info.get("resnorm")


def _cases(dtype):
    return {
        # forced into the PATTERN family (small matrices do not adopt it by themselves): 2 rows per lane, 16 workgroups
        "convdiff_24 const": (gen.convdiff3d(24, 0.3, dtype=dtype), True),
        "convdiff_varying_24 values": (gen.convdiff3d_varying(24, 0.3, dtype=dtype), True),
        "poisson2d_200 const": (gen.poisson2d(200, dtype=dtype), True),
        "banded_13_diagonals values": (gen.banded_random_spd(30000, k=6, seed=4, max_offset=2000, dtype=dtype), True),
        # adopted by the solver itself (>= 2^20 stored entries): 64^3 -> 2 rows per lane, 80^3 -> 4
        "convdiff_64 const": (gen.convdiff3d(64, 0.3, dtype=dtype), False),
        "convdiff_varying_80 values": (gen.convdiff3d_varying(80, 0.3, dtype=dtype), False),
    }


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_single_launch_matches_loop_and_oracle(smm, oracle, modes, dtype):
    P = smm.SolverPreconditioner
    for name, (csr, force) in _cases(dtype).items():
        start, pos, val = csr
        n = len(start) - 1
        A = smm.CSRMatrix(n, n, *csr)
        if force:
            A.set_kernel(PATTERN, 1)
        x_true = np.random.default_rng(3).uniform(0.5, 1.5, n).astype(dtype)
        b = oracle.spmv(csr, 0, None, x_true)
        _, diag = oracle.jacobi_setup(csr)
        for pname, M, pk, pv in (("none", None, 0, None), ("jacobi", A.getPreconditioner(P.JACOBI), PRECOND_JACOBI, diag)):
            # to convergence first (a solve with many iterations ahead is what lets the unforced matrices adopt the PATTERN family): where the
            # stopping test fires depends on the rounding of the global sums -- BiCGStab's convergence is erratic, on the symmetric Poisson
            # matrix the two paths end 20 % apart --, the solution does not
            eps = 1e-4 if dtype == np.float32 else 1e-9
            res = _solve(smm, A, b, -1, eps, M, host.CG_RESIDENT_REQUIRE)
            loop = _solve(smm, A, b, -1, eps, M, host.CG_RESIDENT_OFF)
            assert res[0] == loop[0] == 0, (name, pname)
            assert res[1] <= 1.5 * loop[1] + 5 and loop[1] <= 1.5 * res[1] + 5, (name, pname, res[1], loop[1])
            assert res[3] <= eps and loop[3] <= eps
            # (how close x comes to x_true at this residual is the matrix's condition: the single launch must do as well as the loop)
            err_res, err_loop = float(np.max(np.abs(res[2] - x_true))), float(np.max(np.abs(loop[2] - x_true)))
            assert err_res <= 3 * err_loop + (1e-3 if dtype == np.float32 else 1e-7), (name, pname, err_res, err_loop)
            assert A.get_kernel()[0] == PATTERN
            for maxit in (1, 3, 6):  # (further on, fp32 BiCGStab amplifies the last-bit differences of the sums: the converged runs cover that)
                res = _solve(smm, A, b, maxit, 1e-30, M, host.CG_RESIDENT_REQUIRE)
                loop = _solve(smm, A, b, maxit, 1e-30, M, host.CG_RESIDENT_OFF)
                st_o, x_o, it_o, _ = oracle.bicgstab(csr, b, np.zeros(n, dtype=dtype), maxit, 1e-30, pk, pv)
                assert res[0] == loop[0] == st_o == 0 and res[1] == loop[1] == it_o == maxit, (name, pname, maxit)
                scale = max(1.0, float(np.max(np.abs(x_o))))
                assert float(np.max(np.abs(res[2] - loop[2]))) <= PATHS_TOL[dtype] * scale, (name, pname, maxit)
                assert float(np.max(np.abs(res[2] - x_o))) <= ORACLE_TOL[dtype] * scale, (name, pname, maxit)


@pytest.mark.parametrize("grid,dtype", [(100, np.float64), (108, np.float64), (115, np.float64), (128, np.float32), (144, np.float32)])
def test_every_rows_per_lane_shape(smm, oracle, modes, grid, dtype):
    """8 / 10 / 12 rows per lane in fp64 (10^6, 1.26 M -- BASELINE config 5's size -- and 1.52 M rows), 16 / 24 in fp32: a few iterations of
    the varying-coefficient operator against the loop and the oracle, then the constant one to convergence"""
    csr = gen.convdiff3d_varying(grid, 0.3, dtype=dtype)
    n = len(csr[0]) - 1
    A = smm.CSRMatrix(n, n, *csr)
    A.set_kernel(PATTERN, 1)  # (five iterations ahead are too few for the solver to adopt the family by itself)
    b = gen.row_sums(csr[0], csr[2]).astype(dtype)
    res = _solve(smm, A, b, 5, 1e-30, None, host.CG_RESIDENT_REQUIRE)
    loop = _solve(smm, A, b, 5, 1e-30, None, host.CG_RESIDENT_OFF)
    st_o, x_o, it_o, _ = oracle.bicgstab(csr, b, np.zeros(n, dtype=dtype), 5, 1e-30)
    assert res[:2] == loop[:2] == (st_o, it_o) == (0, 5)
    assert float(np.max(np.abs(res[2] - loop[2]))) <= PATHS_TOL[dtype]
    # (the fp32 oracle adds millions of products one after the other in fp32: on grids of this size BOTH GPU paths sit some 1e-3 away from
    # it, by the same amount -- tests/test_gpu_resident.py has the same note for CG)
    assert float(np.max(np.abs(res[2] - x_o))) <= (2e-2 if dtype == np.float32 else ORACLE_TOL[dtype])
    # ... then a constant-diagonal operator with Jacobi to convergence: the convection-diffusion one in fp64; in fp32 BiCGStab's residual on
    # that non-normal matrix climbs to 1e4-1e6 before it turns NaN in BOTH paths (tools/lab/resident_fp32_diag.py: a property of the method
    # in that precision, not of a kernel), so fp32 takes the symmetric Laplacian of the same grid
    csr = gen.convdiff3d(grid, 0.3, dtype=dtype) if dtype == np.float64 else gen.poisson3d(grid, dtype=dtype)
    A = smm.CSRMatrix(n, n, *csr)
    b = gen.row_sums(csr[0], csr[2]).astype(dtype)
    eps = 2e-2 if dtype == np.float32 else 1e-8  # (fp32 with 2-3 M rows: the attainable residual is ~1e-7 x ||A|| ||x|| sqrt(n) ~ 2e-3)
    res = _solve(smm, A, b, -1, eps, A.getPreconditioner(smm.SolverPreconditioner.JACOBI), host.CG_RESIDENT_REQUIRE)
    loop = _solve(smm, A, b, -1, eps, A.getPreconditioner(smm.SolverPreconditioner.JACOBI), host.CG_RESIDENT_OFF)
    assert res[0] == loop[0] == 0 and res[3] <= eps and loop[3] <= eps and A.pattern_info()[0] == 3  # constant diagonals
    err_res, err_loop = float(np.max(np.abs(res[2] - 1.0))), float(np.max(np.abs(loop[2] - 1.0)))
    assert err_res <= 3 * err_loop + (1e-2 if dtype == np.float32 else 1e-7), (err_res, err_loop)


def test_quirks_and_fall_back(smm, oracle, modes):
    dtype = np.float64
    csr = gen.convdiff3d(24, 0.3, dtype=dtype)
    n = len(csr[0]) - 1
    A = smm.CSRMatrix(n, n, *csr)
    b = gen.row_sums(csr[0], csr[2]).astype(dtype)
    # not in the PATTERN family (too small to adopt it): REQUIRE says so, AUTO runs the loop
    host.bicgstab_resident(host.CG_RESIDENT_REQUIRE)
    with pytest.raises(Exception, match="single-launch"):
        smm.BiCGStab(A, b.copy(), np.zeros(n), 5, 1e-30)
    auto = _solve(smm, A, b, 5, 1e-30, None, host.CG_RESIDENT_AUTO)
    assert auto[:2] == (0, 5)
    A.set_kernel(PATTERN, 1)
    # maxIterations == 0: the body runs once, MAX_ITERATIONS_REACHED (ref:2232, 2277-2282)
    res = _solve(smm, A, b, 0, 1e-30, None, host.CG_RESIDENT_REQUIRE)
    loop = _solve(smm, A, b, 0, 1e-30, None, host.CG_RESIDENT_OFF)
    assert res[:2] == loop[:2] == (2, 1)
    np.testing.assert_allclose(res[2], loop[2], rtol=1e-12)
    # an exact start vector: r = 0, alpha = 0 / 0, x turns NaN and the loop leaves with SUCCESS (the reference's behaviour)
    ones = np.ones(n)
    res = _solve(smm, A, b, 10, 1e-12, None, host.CG_RESIDENT_REQUIRE, x0=ones)
    loop = _solve(smm, A, b, 10, 1e-12, None, host.CG_RESIDENT_OFF, x0=ones)
    assert res[:2] == loop[:2] == (0, 1) and np.isnan(res[2]).all() and np.isnan(loop[2]).all()
    # x is an in / out argument holding the start vector (ref:2215): a start away from zero
    x0 = np.random.default_rng(1).uniform(-1, 1, n)
    res = _solve(smm, A, b, 8, 1e-30, None, host.CG_RESIDENT_REQUIRE, x0=x0)
    st_o, x_o, it_o, _ = oracle.bicgstab(csr, b, x0, 8, 1e-30)
    assert res[:2] == (st_o, it_o) and float(np.max(np.abs(res[2] - x_o))) <= 1e-9
    # a preconditioner the single launch does not take (ILU0) runs the loop also under REQUIRE: only none / Jacobi are tried
    M = A.getPreconditioner(smm.SolverPreconditioner.ILU0)
    res = _solve(smm, A, b, 6, 1e-30, M, host.CG_RESIDENT_REQUIRE)
    assert res[:2] == (0, 6)


def test_fuzz_random_banded_matrices():
    """tools/resident_bicg_fuzz.py: 24 random banded matrices -- 2-16 offsets, holes in every diagonal, row counts that leave workgroups
    partly or wholly idle, constant and varying diagonals, with and without Jacobi, fp32 / fp64 -- single launch against loop and oracle"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "resident_bicg_fuzz.py"), "24"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "single-launch BiCGStab fuzz: ALL OK" in r.stdout
