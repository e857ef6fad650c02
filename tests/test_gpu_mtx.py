"""BASELINE config 5 from a FILE: a non-symmetric Matrix Market matrix written by the committed generator (tools/write_mtx.py) is read
direct to CSR by the drop-in header's SMM::loadMatrix and solved with BiCGStab + Jacobi / ILU0 / SGS on the GPU
(tests/cpp/mtx_bicgstab.cpp); the CSR arrays and x are compared with the generator and the CPU oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sparse_matrix_math_amd import generators as gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
TOOL = os.path.join(ROOT, "tests", "cpp", "mtx_bicgstab")


def _run(path, kind, max_it, eps, dump):
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "mtx_bicgstab"], check=True)
    r = subprocess.run([TOOL, path, kind, str(max_it), repr(eps), dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_mtx_general_file_bicgstab_preconditioned(tmp_path, oracle):
    from write_mtx import write_mtx
    from oracle.oracle import PRECOND_ILU0, PRECOND_JACOBI, PRECOND_SGS

    csr = gen.convdiff3d(14, 0.3, dtype=np.float64)
    n = len(csr[0]) - 1
    path = str(tmp_path / "convdiff14.mtx")
    write_mtx(path, csr, shuffle=True, seed=5)  # a coordinate file promises no order: the reader must sort
    b = gen.row_sums(csr[0], csr[2])
    _, diag = oracle.jacobi_setup(csr)
    _, lu = oracle.ilu0_factorize(csr)
    for kind, code, pv in (("none", 0, None), ("jacobi", PRECOND_JACOBI, diag), ("ilu0", PRECOND_ILU0, lu), ("sgs", PRECOND_SGS, None)):
        for max_it, eps in ((4, 1e-30), (-1, 1e-9)):
            out = _run(path, kind, max_it, eps, str(tmp_path))
            assert (out["rows"], out["nnz"]) == (n, len(csr[1]))
            # the reader rebuilt exactly the generator's CSR arrays (values survive the text round trip: repr() of a double)
            np.testing.assert_array_equal(np.fromfile(tmp_path / "start.i32", dtype=np.int32), csr[0])
            np.testing.assert_array_equal(np.fromfile(tmp_path / "positions.i32", dtype=np.int32), csr[1])
            np.testing.assert_array_equal(np.fromfile(tmp_path / "values.f64", dtype=np.float64), csr[2])
            x = np.fromfile(tmp_path / "x.f64", dtype=np.float64)
            st_ref, x_ref, it_ref, res_ref = oracle.bicgstab(csr, b.copy(), np.zeros(n), max_it, eps, code, pv)
            assert out["status"] == st_ref == 0
            if max_it > 0:  # fixed iterations: same count, x within the solver tolerance of DESIGN.md section 5
                assert out["iterations"] == it_ref == max_it
                assert float(np.max(np.abs(x - x_ref))) <= 1e-10 * float(np.max(np.abs(x_ref)))
            else:
                assert abs(out["iterations"] - it_ref) <= max(2, it_ref // 10) and out["resnorm"] <= eps
                np.testing.assert_allclose(x, np.ones(n), rtol=1e-6)


@pytest.mark.gpu
def test_mtx_varying_coefficients_all_kinds_from_one_load(tmp_path, oracle):
    """what bench.py's extras.mtx_bicgstab runs on every line, at 24^3: the convection-diffusion matrix with SPATIALLY VARYING coefficients
    (every diagonal varies, so the PATTERN family must keep reading values[]: MASKS, never CONST) written as a shuffled `general` file,
    loaded ONCE by SMM::loadMatrix and solved with none / Jacobi / ILU0 / BLOCK_ILU0; x of the last kind against the oracle"""
    from write_mtx import write_mtx

    csr = gen.convdiff3d_varying(24, 0.3, dtype=np.float64)
    n = len(csr[0]) - 1
    path = str(tmp_path / "convdiff_varying24.mtx")
    write_mtx(path, csr, shuffle=True, seed=11)
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "mtx_bicgstab"], check=True)
    r = subprocess.run([TOOL, path, "none,ilu0,block_ilu0,jacobi", "-1", "1e-10", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(ln) for ln in r.stdout.strip().splitlines()]
    assert [ln["precond"] for ln in lines] == ["none", "ilu0", "block_ilu0", "jacobi"]
    np.testing.assert_array_equal(np.fromfile(tmp_path / "positions.i32", dtype=np.int32), csr[1])
    np.testing.assert_array_equal(np.fromfile(tmp_path / "values.f64", dtype=np.float64), csr[2])
    b = gen.row_sums(csr[0], csr[2])
    for ln in lines:
        assert ln["status"] == 0 and ln["resnorm"] <= 1e-10 and ln["max_abs_err_vs_ones"] < 1e-8
        assert ln["pattern_encoding"] in (0, 1)  # row masks + values[] (or no adoption below the solvers' threshold): never constant diagonals
    _, diag = oracle.jacobi_setup(csr)
    from oracle.oracle import PRECOND_JACOBI

    st_ref, x_ref, it_ref, _ = oracle.bicgstab(csr, b.copy(), np.zeros(n), -1, 1e-10, PRECOND_JACOBI, diag)
    x = np.fromfile(tmp_path / "x.f64", dtype=np.float64)  # the last kind's solution
    assert st_ref == 0 and abs(lines[-1]["iterations"] - it_ref) <= max(2, it_ref // 10)
    np.testing.assert_allclose(x, x_ref, rtol=0, atol=1e-8)
    # the strong preconditioners need fewer iterations than none (ILU0 fewest)
    its = {ln["precond"]: ln["iterations"] for ln in lines}
    assert its["ilu0"] < its["block_ilu0"] < its["none"]


@pytest.mark.gpu
def test_mtx_symmetric_and_pattern_files(tmp_path, oracle):
    from write_mtx import write_mtx

    csr = gen.poisson2d(20, dtype=np.float64)
    n = len(csr[0]) - 1
    path = str(tmp_path / "poisson20.mtx")
    write_mtx(path, csr, symmetric=True, shuffle=True)  # lower triangle only: the reader mirrors (ref:2598-2601)
    out = _run(path, "sgs", -1, 1e-10, str(tmp_path))
    np.testing.assert_array_equal(np.fromfile(tmp_path / "positions.i32", dtype=np.int32), csr[1])
    np.testing.assert_array_equal(np.fromfile(tmp_path / "values.f64", dtype=np.float64), csr[2])
    assert out["status"] == 0 and out["max_abs_err_vs_ones"] < 1e-8
    ppath = str(tmp_path / "pattern.mtx")
    write_mtx(ppath, csr, pattern=True)
    out = _run(ppath, "none", 1, 1e-30, str(tmp_path))
    assert out["nnz"] == len(csr[1])
    np.testing.assert_array_equal(np.fromfile(tmp_path / "positions.i32", dtype=np.int32), csr[1])
    np.testing.assert_array_equal(np.fromfile(tmp_path / "values.f64", dtype=np.float64), np.ones(len(csr[1])))
