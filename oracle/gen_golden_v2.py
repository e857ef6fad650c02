#!/usr/bin/env python3
"""tests/golden/reference_outputs_v2.npz: BiCGSymmetric's DIVERGED heuristics (ref:2056-2058, 2079-2081) as the REAL reference
(oracle/_ref/libsmm_ref.so) evaluates them.  Runs only in the build container:   python oracle/gen_golden_v2.py
Inputs are tiny symmetric indefinite matrices written down here (stored in the fixture next to the outputs); outputs are the
reference's status and x.  Data only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.oracle import Reference, build  # noqa: E402
from sparse_matrix_math_amd import generators as gen  # noqa: E402


def dense_to_csr(d, dtype):
    start, pos, val = [0], [], []
    for r in range(d.shape[0]):
        for c in range(d.shape[1]):
            if d[r, c] != 0:
                pos.append(c)
                val.append(d[r, c])
        start.append(len(pos))
    return np.array(start, dtype=np.int32), np.array(pos, dtype=np.int32), np.array(val, dtype=dtype)


def cases(dtype):
    """name -> (csr, b, maxIterations, eps)"""
    out = {}
    swap = np.array([[0.0, 1.0], [1.0, 0.0]])
    # p.Ap == 0 with ||r||^2 = 9 > 1: the first heuristic fires in the first pass (ref:2056-2058)
    out["swap_denominator_zero"] = (dense_to_csr(swap, dtype), np.array([3.0, 0.0], dtype=dtype), -1, 1e-6)
    # the same matrix with a right-hand side it solves in one step
    out["swap_success"] = (dense_to_csr(swap, dtype), np.array([1.0, 1.0], dtype=dtype), -1, 1e-6)
    # diag(1, -1): alpha = 0.3 / 0.02 = 15 blows the residual up from 0.3 (< eps = 0.5) to ~67 (> 1): second heuristic (ref:2079-2081)
    pm = np.diag([1.0, -1.0])
    out["residual_growth"] = (dense_to_csr(pm, dtype), np.array([0.4, np.sqrt(0.14)], dtype=dtype), -1, 0.5)
    # a small |p.Ap| < eps with ||r||^2 <= 1: NOT diverged by the first test (both conditions are needed)
    out["small_denominator_small_residual"] = (dense_to_csr(pm, dtype), np.array([0.5, 0.5 - 1e-3], dtype=dtype), 3, 1e-2)
    # indefinite shifted Laplacians: the reference decides after several passes
    for n, shift, seed in ((6, 3.0, 1), (8, 3.7, 2), (10, 4.2, 3)):
        st, pos, val = gen.poisson2d(n, dtype=np.float64)
        rows = len(st) - 1
        val = val.copy()
        for r in range(rows):
            k = st[r] + int(np.searchsorted(pos[st[r]:st[r + 1]], r))
            val[k] -= shift
        b = np.random.default_rng(seed).uniform(-2, 2, rows)
        for eps in (1e-3, 0.9):
            out[f"shifted_poisson_{n}_eps{eps}"] = ((st, pos, val.astype(dtype)), b.astype(dtype), -1, eps)
    return out


def main():
    build(ref=True)
    if not Reference.available():
        raise SystemExit("oracle/_ref/libsmm_ref.so missing: /root/reference is not mounted here")
    ref = Reference()
    out = {}
    for dtype in (np.float32, np.float64):
        dn = np.dtype(dtype).name
        for name, (csr, b, maxit, eps) in cases(dtype).items():
            tag = f"bicgsymmetric/{name}/{dn}"
            rows = len(csr[0]) - 1
            with ref.csr(csr) as m:
                st, x = ref.bicgsymmetric(m, b.copy(), np.zeros(rows, dtype=dtype), maxit, dtype(eps))
            out[f"{tag}/start"], out[f"{tag}/positions"], out[f"{tag}/values"] = csr
            out[f"{tag}/b"], out[f"{tag}/maxit"], out[f"{tag}/eps"] = b, np.int32(maxit), np.float64(eps)
            out[f"{tag}/status"], out[f"{tag}/x"] = np.int32(st), x
            print(f"{tag}: status {st}")
    path = os.path.join(ROOT, "tests", "golden", "reference_outputs_v2.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
