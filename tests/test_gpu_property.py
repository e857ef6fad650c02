"""Property tests (hypothesis) of the HIP SpMV and dot against the oracle: arbitrary ragged CSR shapes -- empty rows anywhere,
single-entry rows, rows longer than an LDS tile, rectangular matrices, every kernel family and lanes-per-row setting, the
three ops, in-place output -- instead of the handful of shapes the fixed tests pick."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle.oracle import OP_ADD, OP_ASSIGN, OP_SUB

pytestmark = pytest.mark.gpu
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])


@st.composite
def ragged_csr(draw):
    dtype = draw(st.sampled_from([np.float32, np.float64]))
    rows = draw(st.integers(1, 180))
    cols = draw(st.integers(1, 400))
    seed = draw(st.integers(0, 2**31 - 1))
    shape = draw(st.sampled_from(["short", "mixed", "long", "empty-heavy"]))
    rng = np.random.default_rng(seed)
    if shape == "short":
        lens = rng.integers(0, min(cols, 9) + 1, size=rows)
    elif shape == "mixed":
        lens = rng.integers(0, min(cols, 70) + 1, size=rows)
    elif shape == "empty-heavy":
        lens = np.where(rng.random(rows) < 0.7, 0, rng.integers(1, min(cols, 12) + 1, size=rows))
    else:  # a few rows far longer than one LDS tile (1021 entries at the smallest capacity)
        cols = draw(st.integers(1500, 6000))
        lens = rng.integers(0, 20, size=rows)
        for r in rng.choice(rows, size=min(rows, 3), replace=False):
            lens[r] = rng.integers(1022, cols + 1)
    start = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(lens, out=start[1:])
    pos = np.concatenate([np.sort(rng.choice(cols, size=int(n), replace=False)) for n in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    val = rng.uniform(-2, 2, int(start[-1])).astype(dtype)
    x = rng.uniform(-2, 2, cols).astype(dtype)
    lhs = rng.uniform(-2, 2, rows).astype(dtype)
    return dtype, rows, cols, (start, pos, val), x, lhs


def bound(csr, x, dtype, lhs):
    start, pos, val = csr
    rows = len(start) - 1
    mag = np.zeros(rows)
    np.add.at(mag, np.repeat(np.arange(rows), np.diff(start)), np.abs(val.astype(np.float64) * x[pos].astype(np.float64)))
    lens = np.maximum(np.diff(start), 1)
    return (lens + 2) * np.finfo(dtype).eps * (mag + np.abs(lhs)) + np.finfo(dtype).tiny


@settings(max_examples=60, **COMMON)
@given(case=ragged_csr(), family=st.sampled_from([0, 1, 2]), lanes=st.sampled_from([1, 2, 4, 16, 64]), op=st.sampled_from([OP_ASSIGN, OP_ADD, OP_SUB]),
       inplace=st.booleans())
def test_spmv_any_shape(smm, oracle, case, family, lanes, op, inplace):
    dtype, rows, cols, csr, x, lhs = case
    A = smm.CSRMatrix(rows, cols, *csr)
    A.set_kernel(family, 0 if family == 0 else lanes)
    ref = oracle.spmv(csr, op, lhs, x)
    out = lhs.copy() if (inplace and op != OP_ASSIGN) else np.full(rows, 99, dtype=dtype)
    src = out if (inplace and op != OP_ASSIGN) else lhs
    {OP_ASSIGN: lambda: A.rMult(x, out), OP_ADD: lambda: A.rMultAdd(src, x, out), OP_SUB: lambda: A.rMultSub(src, x, out)}[op]()
    fam, ln = A.get_kernel()
    if ln == 1 and int(np.diff(csr[0]).max(initial=0)) <= 1021:
        # one lane per row and no over-long row: the reference's order of additions, bit for bit (ref:1484-1489)
        np.testing.assert_array_equal(out, ref)
    else:
        assert np.all(np.abs(out.astype(np.float64) - ref) <= bound(csr, x, dtype, lhs if op != OP_ASSIGN else 0 * lhs)), (fam, ln, op)
    # empty rows give op(lhs, 0) exactly (ref:1476-1483)
    empty = np.diff(csr[0]) == 0
    want = {OP_ASSIGN: np.zeros(rows, dtype=dtype), OP_ADD: lhs, OP_SUB: lhs}[op]
    np.testing.assert_array_equal(out[empty], want[empty])


@settings(max_examples=40, **COMMON)
@given(n=st.integers(0, 70_000), seed=st.integers(0, 2**31 - 1), dtype=st.sampled_from([np.float32, np.float64]), same=st.booleans())
def test_dot_any_length(smm, n, seed, dtype, same):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-1, 1, n).astype(dtype)
    b = a if same else rng.uniform(-1, 1, n).astype(dtype)
    got = float(smm.dot(a, b))
    exact = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    tol = 8 * np.finfo(dtype).eps * float(np.abs(a.astype(np.float64) * b).sum()) * max(1.0, np.log2(max(n, 2)))
    assert abs(got - exact) <= tol + 1e-300
    assert float(smm.dot(a, b)) == got  # bitwise reproducible
    if same:
        assert got >= 0.0
