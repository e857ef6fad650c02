"""GPU parity of the Krylov loops through the C ABI: fixed-iteration runs against the committed outputs of the real
reference (x within a stated tolerance, iteration count and status identical), converged runs, the reference's own
asset tests, edge semantics, and BASELINE config 2 (1000x1000 Poisson, CG, tol 1e-6) against the oracle."""
import numpy as np
import pytest
from test_oracle import gen_matrices

from oracle.oracle import PRECOND_ILU0, PRECOND_JACOBI, PRECOND_NONE, PRECOND_SGS
from sparse_matrix_math_amd import generators as gen

pytestmark = pytest.mark.gpu
DTYPES = [np.float32, np.float64]
# Tolerance on x for a fixed number of iterations: only the summation order of the dot products (and of rows split over
# several lanes) differs from the reference, so x agrees to a small multiple of the unit roundoff amplified by the
# conditioning of the recurrences; stated relative to max|x|.
RTOL = {np.float32: 3e-4, np.float64: 1e-10}


def close(x, ref, dtype, scale=1.0):
    return float(np.max(np.abs(x.astype(np.float64) - ref))) <= RTOL[dtype] * scale * max(1.0, float(np.max(np.abs(ref))))


def make(smm, csr):
    rows = len(csr[0]) - 1
    return smm.CSRMatrix(rows, rows, *csr)


def bicgstab_sensitivity(oracle, csr, b, it, precond, precond_values):
    """How far the REFERENCE ALGORITHM's own x moves when b changes by one unit in the last place (three random sign
    patterns), in the oracle.  A different summation order of the dot products is a perturbation of that kind, and BiCGStab
    in fp32 can amplify it by 10^3 within ten iterations on the non-symmetric test matrix (measured: 1e-4 .. 2e-3 for
    max|x| = 2.7); a fixed tolerance that ignores this passes or fails by luck."""
    rows = len(b)
    base = oracle.bicgstab(csr, b, np.zeros(rows, dtype=b.dtype), it, 1e-30, precond, precond_values)[1].astype(np.float64)
    worst = 0.0
    for seed in range(3):
        sign = np.random.default_rng(seed).choice([-1.0, 1.0], size=rows).astype(b.dtype)
        pert = np.nextafter(b, b + sign).astype(b.dtype)
        x = oracle.bicgstab(csr, pert, np.zeros(rows, dtype=b.dtype), it, 1e-30, precond, precond_values)[1]
        worst = max(worst, float(np.max(np.abs(x.astype(np.float64) - base))))
    return worst


@pytest.mark.parametrize("dtype", DTYPES)
def test_fixed_iterations_match_reference(smm, golden, oracle, dtype):
    P = smm.SolverPreconditioner
    dn = np.dtype(dtype).name
    for mname, csr in gen_matrices(dtype).items():
        if mname == "ragged_300":
            continue
        symmetric = mname != "convdiff3d_12"
        start, pos, val = csr
        rows = len(start) - 1
        A = make(smm, csr)
        b = gen.row_sums(start, val)
        tag = f"gen/{mname}/{dn}"
        preconds = {"none": None, "sgs": A.getPreconditioner(P.SYMMETRIC_GAUS_SEIDEL), "jacobi": A.getPreconditioner(P.JACOBI)}
        ocodes = {"none": (PRECOND_NONE, None), "sgs": (PRECOND_SGS, None), "jacobi": (PRECOND_JACOBI, oracle.jacobi_setup(csr)[1])}
        for it in (1, 3, 10):
            if symmetric:
                x0 = np.zeros(rows, dtype=dtype)
                x = np.full(rows, 123, dtype=dtype)
                info = {}
                st = smm.ConjugateGradient(A, b, x0, x, it, 0.0, info=info)
                assert int(st) == int(golden[f"{tag}/cg/it{it}/status"]) == 2 and info["iterations"] == it
                assert close(x, golden[f"{tag}/cg/it{it}/x"], dtype), (mname, "cg", it)
            for pname, M in preconds.items():
                x = np.zeros(rows, dtype=dtype)
                info = {}
                st = smm.BiCGStab(A, b.copy(), x, it, 1e-30, M, info=info)
                assert int(st) == int(golden[f"{tag}/bicgstab_{pname}/it{it}/status"]) == 0 and info["iterations"] == it
                ref = golden[f"{tag}/bicgstab_{pname}/it{it}/x"]
                allowed = max(RTOL[dtype] * max(1.0, float(np.max(np.abs(ref)))), 4 * bicgstab_sensitivity(oracle, csr, b, it, *ocodes[pname]))
                assert float(np.max(np.abs(x.astype(np.float64) - ref))) <= allowed, (mname, pname, it)
        if symmetric:
            x = np.zeros(rows, dtype=dtype)
            st = smm.ConjugateGradient(A, b, x, x, -1, 1e-6)  # x aliases x0, as the reference's tests call it
            assert int(st) == int(golden[f"{tag}/cg/conv/status"]) == 0
            assert close(x, golden[f"{tag}/cg/conv/x"], dtype, 10), mname
            M = A.getPreconditioner(P.IC0)
            x = np.zeros(rows, dtype=dtype)
            info = {}
            st = smm.ConjugateGradient(A, b, np.zeros(rows, dtype=dtype), x, 5, 0.0, M, info=info)
            assert int(st) == int(golden[f"{tag}/pcg_ic0/it5/status"]) and info["iterations"] == 5
            assert close(x, golden[f"{tag}/pcg_ic0/it5/x"], dtype), mname
        x = np.zeros(rows, dtype=dtype)
        st = smm.BiCGStab(A, b.copy(), x, -1, 1e-6)
        assert int(st) == int(golden[f"{tag}/bicgstab_none/conv/status"]) == 0
        assert close(x, golden[f"{tag}/bicgstab_none/conv/x"], dtype, 10), mname


@pytest.mark.parametrize("dtype", DTYPES)
def test_reference_asset_cases(smm, golden, dtype):
    """test/cpp/cg.cpp:7-26, 62-84, test/cpp/bicgstab.cpp:124-167, test/cpp/bicgsymmetric.cpp:7-26 on the HIP path:
    status SUCCESS and every x_i == 1 within infEps<T> (1e-4 / 1e-8 relative)"""
    P = smm.SolverPreconditioner
    dn = np.dtype(dtype).name
    eps = {np.float32: 1e-4, np.float64: 1e-8}[dtype]
    for key in ("mesh1e1", "mesh1em1", "mesh1em6"):
        start, pos = golden[f"asset/{key}/start"], golden[f"asset/{key}/positions"]
        val = golden[f"asset/{key}/values"].astype(dtype)
        rows = len(start) - 1
        A = smm.CSRMatrix(rows, rows, start, pos, val)
        b = gen.row_sums(start, val)
        runs = {}
        x = np.zeros(rows, dtype=dtype)
        runs["cg"] = (smm.ConjugateGradient(A, b, x, x, -1, eps), x)
        x = np.zeros(rows, dtype=dtype)
        runs["bicgstab"] = (smm.BiCGStab(A, b.copy(), x, -1, eps), x)
        x = np.zeros(rows, dtype=dtype)
        runs["bicgstab_sgs"] = (smm.BiCGStab(A, b.copy(), x, -1, eps, A.getPreconditioner(P.SYMMETRIC_GAUS_SEIDEL)), x)
        x = np.zeros(rows, dtype=dtype)
        runs["pcg_ic0"] = (smm.ConjugateGradient(A, b, x, x, -1, eps, A.getPreconditioner(P.IC0)), x)
        x = np.zeros(rows, dtype=dtype)
        runs["bicgsymmetric"] = (smm.BiCGSymmetric(A, b.copy(), x, -1, eps), x)
        for name, (st, x) in runs.items():
            assert int(st) == int(golden[f"asset/{key}/{dn}/{name}/status"]) == 0, (key, name)
            np.testing.assert_allclose(x, 1.0, rtol=eps, err_msg=f"{key} {name}")
            np.testing.assert_allclose(x, golden[f"asset/{key}/{dn}/{name}/x"], rtol=10 * eps, err_msg=f"{key} {name}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_edge_semantics(smm, golden, dtype):
    dn = np.dtype(dtype).name
    csr = gen.poisson2d(32, dtype=dtype)
    rows = len(csr[0]) - 1
    A = make(smm, csr)
    b = gen.row_sums(csr[0], csr[2])
    ones = np.ones(rows, dtype=dtype)
    # exact x0: CG returns SUCCESS before the loop and leaves x untouched (ref:2342-2344)
    x = np.full(rows, 7, dtype=dtype)
    info = {}
    st = smm.ConjugateGradient(A, b, ones, x, -1, 1e-3, info=info)
    assert int(st) == int(golden[f"edge/{dn}/cg_exact_x0/status"]) == 0 and info["iterations"] == 0
    np.testing.assert_array_equal(x, 7)
    # maxIterations == 0: CG -> MAX_ITERATIONS_REACHED without touching x
    x = np.full(rows, 7, dtype=dtype)
    st = smm.ConjugateGradient(A, b, np.zeros(rows, dtype=dtype), x, 0, 1e-6, info=info)
    assert int(st) == int(golden[f"edge/{dn}/cg_maxit0/status"]) == 2 and info["iterations"] == 0
    np.testing.assert_array_equal(x, 7)
    # BiCGStab maxIterations == 0: one pass, MAX_ITERATIONS_REACHED (ref:2277-2281)
    x = np.zeros(rows, dtype=dtype)
    st = smm.BiCGStab(A, b.copy(), x, 0, 1e-6, info=info)
    assert int(st) == int(golden[f"edge/{dn}/bicgstab_maxit0/status"]) == 2 and info["iterations"] == 1
    assert float(np.max(np.abs(x - golden[f"edge/{dn}/bicgstab_maxit0/x"]))) <= RTOL[dtype]
    # BiCGStab from the exact solution: rr0 == 0 -> NaN after one pass, status SUCCESS (no breakdown guard, ref:2260, 2270)
    x = ones.copy()
    st = smm.BiCGStab(A, b.copy(), x, -1, 1e-6, info=info)
    assert int(st) == int(golden[f"edge/{dn}/bicgstab_exact_x0/status"]) == 0 and info["iterations"] == 1
    assert np.isnan(x).all()
    # a 0x0 system
    E = smm.CSRMatrix(0, 0, np.zeros(1, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=dtype))
    z = np.zeros(0, dtype=dtype)
    assert int(smm.ConjugateGradient(E, z, z, z, -1, 1e-6)) == 0  # eps^2 > ||r||^2 == 0: SUCCESS before the loop (ref:2342)
    # wrong preconditioner kinds are rejected
    with pytest.raises(smm.SmmHipError):
        smm.ConjugateGradient(A, b, ones, x, 1, 0.0, A.getPreconditioner(smm.SolverPreconditioner.JACOBI))
    with pytest.raises(smm.SmmHipError):
        smm.BiCGStab(A, b.copy(), x, 1, 0.0, A.getPreconditioner(smm.SolverPreconditioner.IC0))


def test_config2_poisson_1000_cg(smm, oracle):
    """BASELINE.json config 2: 1000x1000 5-point Laplacian fp64, CG, b = A*1, x0 = 0, tol 1e-6 -- compare x / residual /
    iteration count with the CPU run of config 1 (oracle; the reference itself takes 1693 iterations)"""
    csr = gen.poisson2d(1000, dtype=np.float64)
    start, pos, val = csr
    n = 1000 * 1000
    b = gen.row_sums(start, val)
    st_ref, x_ref, it_ref, _ = oracle.cg(csr, b, np.zeros(n), -1, 1e-6)
    assert st_ref == 0 and it_ref == 1693
    A = smm.CSRMatrix(n, n, *csr)
    x = np.zeros(n)
    info = {}
    st = smm.ConjugateGradient(A, b, x, x, -1, 1e-6, info=info)
    assert int(st) == 0
    assert abs(info["iterations"] - it_ref) <= 17, info  # within 1 %: the stopping test sits on a plateau of the residual
    assert float(np.max(np.abs(x - x_ref))) <= 1e-6
    r = b - oracle.spmv(csr, 0, None, x)
    assert float(np.linalg.norm(r)) <= 5e-6
    assert float(np.max(np.abs(x - 1))) <= 1e-6
    # fixed iteration count: identical count, x to rounding
    x = np.zeros(n)
    st = smm.ConjugateGradient(A, b, x, x, 200, 0.0, info=info)
    st_ref, x_ref, it_ref, _ = oracle.cg(csr, b, np.zeros(n), 200, 0.0)
    assert int(st) == st_ref == 2 and info["iterations"] == it_ref == 200
    assert float(np.max(np.abs(x - x_ref))) <= 1e-9


@pytest.mark.parametrize("dtype", DTYPES)
def test_config5_nonsymmetric_preconditioned(smm, oracle, dtype):
    """BASELINE.json config 5 stand-in: non-symmetric convection-diffusion, BiCGStab with none / Jacobi / ILU0 (and SGS):
    fixed iterations against the oracle, converged runs reach the all-ones solution, ILU0 needs fewer iterations"""
    P = smm.SolverPreconditioner
    csr = gen.convdiff3d(24, 0.3, dtype=dtype)
    start, pos, val = csr
    n = len(start) - 1
    A = smm.CSRMatrix(n, n, *csr)
    b = gen.row_sums(start, val)
    _, diag = oracle.jacobi_setup(csr)
    _, lu = oracle.ilu0_factorize(csr)
    kinds = {"none": (None, 0, None), "jacobi": (A.getPreconditioner(P.JACOBI), 1, diag), "ilu0": (A.getPreconditioner(P.ILU0), PRECOND_ILU0, lu),
             "sgs": (A.getPreconditioner(P.SYMMETRIC_GAUS_SEIDEL), 3, None)}
    iters = {}
    for name, (M, pk, pv) in kinds.items():
        x = np.zeros(n, dtype=dtype)
        info = {}
        st = smm.BiCGStab(A, b.copy(), x, 6, 1e-30, M, info=info)
        st_ref, x_ref, it_ref, _ = oracle.bicgstab(csr, b, np.zeros(n, dtype=dtype), 6, 1e-30, pk, pv)
        assert int(st) == st_ref == 0 and info["iterations"] == it_ref == 6
        assert close(x, x_ref, dtype), name
        eps = 1e-4 if dtype == np.float32 else 1e-9
        x = np.zeros(n, dtype=dtype)
        st = smm.BiCGStab(A, b.copy(), x, -1, eps, M, info=info)
        assert int(st) == 0
        np.testing.assert_allclose(x, 1.0, rtol=100 * eps, err_msg=name)
        iters[name] = info["iterations"]
        st_ref, _, it_ref, _ = oracle.bicgstab(csr, b, np.zeros(n, dtype=dtype), -1, eps, pk, pv)
        assert abs(info["iterations"] - it_ref) <= max(2, it_ref // 5), (name, info, it_ref)
    assert iters["ilu0"] < iters["none"] and iters["sgs"] < iters["none"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_bicgsymmetric_diverged_heuristics_match_reference(smm, golden_v2, dtype):
    """BiCGSymmetric's DIVERGED branches (ref:2056-2058, 2079-2081) on the GPU against the real reference's decisions: same status; x
    within the solver tolerance (identical where the loop leaves before any update)"""
    from conftest import bicgsymmetric_cases

    tol = 3e-4 if dtype == np.float32 else 1e-10
    statuses = set()
    for name, csr, b, maxit, eps, st_ref, x_ref in bicgsymmetric_cases(golden_v2, dtype):
        n = len(b)
        A = smm.CSRMatrix(n, n, *csr)
        x = np.zeros(n, dtype=dtype)
        st = smm.BiCGSymmetric(A, b.copy(), x, maxit, dtype(eps))
        assert int(st) == st_ref, name
        scale = max(float(np.max(np.abs(x_ref))), 1.0)
        # the 2 x 2 cases leave the loop within a pass or two: x agrees to rounding.  The indefinite shifted Laplacians run tens of
        # ill-conditioned passes before the reference decides: the DECISION must agree; x agrees as far as eps / the blow-up allow
        bound = 50 * tol if len(b) == 2 else max(50 * tol, 20 * eps, 0.05 if st_ref == 1 else 0.0)
        assert float(np.max(np.abs(x - x_ref))) <= bound * scale, name
        statuses.add(int(st))
    assert statuses == {0, 1}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_cg_deferred_x_update_is_bit_identical(smm, oracle, dtype):
    """cgLazyXP (csrc/smm_solvers.hip): for vectors beyond the caches CG keeps its last eight directions and brings x up to date every eighth
    iteration, in the last planned one and in whichever launch finds the iteration converged -- the reference's roundings in the
    reference's order (ref:2362-2366).  Forced on at a small size: every iteration count from 0 to 21, convergence inside a window of eight,
    x0 in place and apart, must give the bits of the eager loop."""
    from sparse_matrix_math_amd import host

    csr = gen.poisson2d(150, dtype=dtype)
    start, pos, val = csr
    n = len(start) - 1
    A = smm.CSRMatrix(n, n, *csr)
    b = gen.row_sums(start, val)
    rng = np.random.default_rng(9)
    before = host.cg_resident(-1)
    host.cg_resident(host.CG_RESIDENT_OFF)  # (the register-resident solve would take a matrix of this size first)
    try:
        for maxit, eps in [(k, 0.0) for k in range(0, 22)] + [(-1, 1e-3 if dtype == np.float32 else 1e-8), (-1, 2.0), (500, 1e-1)]:
            for in_place in (True, False):
                got = {}
                for lazy in (True, False):
                    host.set_cg_lazy_x_min_bytes(0 if lazy else 1 << 60)
                    x0 = rng.uniform(-1, 1, n).astype(dtype) if maxit != 3 else np.zeros(n, dtype=dtype)
                    rng = np.random.default_rng(9 + abs(maxit))  # (the same x0 for both)
                    x0 = rng.uniform(-1, 1, n).astype(dtype)
                    x = x0.copy() if in_place else np.full(n, 7, dtype=dtype)
                    info = {}
                    st = smm.ConjugateGradient(A, b, x if in_place else x0, x, maxit, dtype(eps), info=info)
                    got[lazy] = (int(st), info["iterations"], x.copy())
                assert got[True][:2] == got[False][:2], (maxit, eps, in_place)
                np.testing.assert_array_equal(got[True][2], got[False][2], err_msg=f"maxit {maxit} eps {eps} in_place {in_place}")
        st_o, x_o, it_o, _ = oracle.cg(csr, b, np.zeros(n, dtype=dtype), 9, 0.0)
        host.set_cg_lazy_x_min_bytes(0)
        x = np.zeros(n, dtype=dtype)
        st = smm.ConjugateGradient(A, b, x, x, 9, dtype(0.0))
        assert int(st) == st_o == 2
        np.testing.assert_allclose(x, x_o, rtol=3e-4 if dtype == np.float32 else 1e-10, atol=1e-6 if dtype == np.float32 else 1e-12)
    finally:
        host.set_cg_lazy_x_min_bytes(-1)
        host.cg_resident(before)


def test_cg_direction_formed_inside_the_spmv_is_bit_identical():
    """tools/cg_fuse_check.py in a process of its own (the non-temporal policy and the march threshold are read from the environment once):
    ConjugateGradient with p = beta p_old + r formed in the load phase of the 2.5-D SpMV kernel (MarchFuse) and x deferred, against the
    deferred-x loop and the eager three-launch loop -- bit for bit, 3-D and 2-D stencils with partial tiles and planes, every iteration count
    0-19, convergence inside the loop, x0 in place and apart, fp32 / fp64 -- and against the oracle"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "cg_fuse_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "cg fuse check: ALL OK" in r.stdout and r.stdout.count("fused == deferred == eager") == 6
