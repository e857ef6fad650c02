#!/usr/bin/env python3
"""tests/golden/reference_outputs_v3.npz: what the REAL reference (oracle/_ref/libsmm_ref.so) computes for the block preconditioners'
defining construction -- its own SGSPreconditioner::apply (ref:1658-1713) of the BLOCK-DIAGONAL part of a matrix, and its own
BiCGStab<SGSPreconditioner> template (ref:2191-2283) on the whole matrix with that preconditioner plugged in.  SMM_PRECOND_BLOCK_SGS
and the oracle's smm_oracle_block_sgs_* must reproduce these bit for bit / within the solver tolerance.
Runs only in the build container:   python oracle/gen_golden_v3.py        Data only (inputs come from the repo's own generators)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.oracle import Reference, block_diagonal_part, build  # noqa: E402
from sparse_matrix_math_amd import generators as gen  # noqa: E402


def matrices(dtype):
    return {
        "poisson2d_32": gen.poisson2d(32, dtype=dtype),
        "banded_2000": gen.banded_random_spd(2000, k=25, seed=0x5EED, max_offset=1 << 20, dtype=dtype),
        "banded_near_3000": gen.banded_random_spd(3000, k=12, seed=11, max_offset=90, dtype=dtype),
        "convdiff3d_12": gen.convdiff3d(12, 0.3, dtype=dtype),
    }


def bounds_sets(rows):
    """name -> nblocks + 1 row numbers"""
    uni = lambda b: np.append(np.arange(0, rows, b), rows).astype(np.int32)  # noqa: E731
    return {"one": np.array([0, rows], dtype=np.int32), "u64": uni(64), "u256": uni(256), "ragged": np.array([0, 1, 65, 66, 700, rows], dtype=np.int32)}


def rhs_of(rows, dtype):
    return np.random.default_rng(4321).uniform(-1, 1, rows).astype(dtype)


def main():
    build(ref=True)
    if not Reference.available():
        raise SystemExit("oracle/_ref/libsmm_ref.so missing: /root/reference is not mounted here")
    ref = Reference()
    out = {}
    for dtype in (np.float32, np.float64):
        dn = np.dtype(dtype).name
        for mname, csr in matrices(dtype).items():
            rows = len(csr[0]) - 1
            rhs = rhs_of(rows, dtype)
            b = gen.row_sums(csr[0], csr[2])
            for bname, bounds in bounds_sets(rows).items():
                tag = f"block_sgs/{mname}/{bname}/{dn}"
                bd, _ = block_diagonal_part(csr, bounds)
                with ref.csr(csr) as a, ref.csr(bd) as m:
                    err, x = ref.sgs_apply(m, rhs)
                    assert err == 0
                    out[f"{tag}/apply/x"] = x
                    for maxit in (1, 3, 10):
                        st, xs = ref.bicgstab_sgs_of(a, m, b, np.zeros(rows, dtype=dtype), maxit, dtype(1e-30))
                        out[f"{tag}/bicgstab/{maxit}/status"], out[f"{tag}/bicgstab/{maxit}/x"] = np.int32(st), xs
                print(tag, "ok")
    path = os.path.join(ROOT, "tests", "golden", "reference_outputs_v3.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
