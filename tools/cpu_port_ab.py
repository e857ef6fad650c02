#!/usr/bin/env python3
"""One core of the GPU box's host: the OpenMP port's SpMV in its three loop forms (1 / 2 / 4 rows in lock step) against the REAL reference (oracle/_ref/libsmm_ref.so) on the same banded matrix -- which form bench.py's cpu_baseline should
time so that the port is not slower than the reference on one core.   python tools/cpu_port_ab.py [rows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import Oracle, Reference  # noqa: E402
from sparse_matrix_math_amd import generators as gen  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
csr = gen.banded_random_spd(rows, k=25, seed=0x5EED, max_offset=1 << 20, dtype=np.float32)
n = len(csr[0]) - 1
x = np.random.default_rng(1).uniform(0.5, 1.5, n).astype(np.float32)
print(f"matrix: {n} rows, {len(csr[1])} entries, fp32; one core; best of 3 SpMVs, then 6 BiCGStab iterations (13 SpMVs)")


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


x0 = np.zeros(n, dtype=np.float32)
for name, o in (("gcc -O2", Oracle()),):
    o.set_threads(1)
    for form in (1, 2, 4):
        o.set_spmv_form(form)
        t_spmv = best(lambda: o.spmv(csr, 0, None, x, omp=True))
        t0 = time.perf_counter()
        o.bicgstab(csr, x, x0, 6, 0.0, omp=True)
        t_solve = time.perf_counter() - t0
        print(f"port {name:10s} rows in lock step {form}: SpMV {t_spmv * 1e3:8.1f} ms   6 iterations {t_solve:6.2f} s = {6 / t_solve:5.2f} it/s", flush=True)
    o.set_spmv_form(1)
    t0 = time.perf_counter()
    o.bicgstab(csr, x, x0, 6, 0.0)  # the serial restatement (no OpenMP regions at all)
    t_solve = time.perf_counter() - t0
    print(f"serial restatement {name}: 6 iterations {t_solve:6.2f} s = {6 / t_solve:5.2f} it/s", flush=True)
if Reference.available():
    ref = Reference()
    with ref.csr(csr) as m:
        t_spmv = best(lambda: ref.spmv(m, 0, None, x))
        t0 = time.perf_counter()
        ref.bicgstab(m, x, x0, 6, 0.0)
        t_solve = time.perf_counter() - t0
    print(f"REAL reference (clang -O2): SpMV {t_spmv * 1e3:8.1f} ms   6 iterations {t_solve:6.2f} s = {6 / t_solve:5.2f} it/s")
