#!/usr/bin/env python3
"""Latency of the FIRST SpMV on a freshly wrapped matrix (handle creation + tile table + launch) vs a later one, on BASELINE config 4's
matrix (3-D Laplacian 512^3, fp64, 134 M rows).  Works with any build of the library through plain ctypes (old builds lack newer
symbols):   python tools/first_spmv_latency.py [path/to/libsmm_hip.so]"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sparse_matrix_math_amd", "lib", "libsmm_hip.so")
ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
lib = ctypes.CDLL(path)
P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
lib.smm_hip_gen_stencil3d_nnz.restype = ctypes.c_longlong
lib.smm_hip_gen_stencil3d_nnz.argtypes = [I, I, I]
lib.smm_hip_gen_stencil3d_dev_f64.argtypes = [I, I, I, D, D, D, P, P, P, P]
lib.smm_hip_csr_create_dev_f64.argtypes = [I, I, P, P, P, ctypes.POINTER(P)]
lib.smm_hip_spmv_dev_f64.argtypes = [P, I, P, P, P, P]
lib.smm_hip_csr_destroy.argtypes = [P]
assert lib.smm_hip_init(0) == 0
dev = torch.device("cuda:0")
N = 512
n, nnz = N ** 3, lib.smm_hip_gen_stencil3d_nnz(N, N, N)
ds = torch.empty(n + 1, dtype=torch.int32, device=dev)
dp = torch.empty(nnz, dtype=torch.int32, device=dev)
dv = torch.empty(nnz, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
assert lib.smm_hip_gen_stencil3d_dev_f64(N, N, N, 6.0, -1.0, -1.0, ds.data_ptr(), dp.data_ptr(), dv.data_ptr(), stream) == 0
x = torch.ones(n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
torch.cuda.synchronize()
for rep in range(3):
    h = P()
    t0 = time.perf_counter()
    assert lib.smm_hip_csr_create_dev_f64(n, n, ds.data_ptr(), dp.data_ptr(), dv.data_ptr(), ctypes.byref(h)) == 0
    t1 = time.perf_counter()
    assert lib.smm_hip_spmv_dev_f64(h, 0, None, x.data_ptr(), y.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert lib.smm_hip_spmv_dev_f64(h, 0, None, x.data_ptr(), y.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"{os.path.basename(path)}: create {1e3 * (t1 - t0):8.2f} ms   first SpMV (tile table + launch) {1e3 * (t2 - t1):8.2f} ms   next SpMV {1e3 * (t3 - t2):6.2f} ms", flush=True)
    lib.smm_hip_csr_destroy(h)
