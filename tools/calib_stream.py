#!/usr/bin/env python3
"""Streaming-rate calibration on the GPU box: our dot kernel, torch sum / copy, on arrays far larger than the Infinity Cache."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sparse_matrix_math_amd as smm
from sparse_matrix_math_amd import host

smm.init(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
n = 500_000_000
a = torch.rand(n, dtype=torch.float32, device=dev)
b = torch.rand(n, dtype=torch.float32, device=dev)
r = torch.zeros(1, dtype=torch.float32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timeit(lambda: host.dot_dev(n, a, b, r, np.float32, stream))
print(f"smm dot (2 x {n * 4 / 1e9:.1f} GB read): {ms:.3f} ms  {2 * n * 4 / ms / 1e6:.0f} GB/s")
ms = timeit(lambda: torch.dot(a, b))
print(f"torch.dot: {ms:.3f} ms  {2 * n * 4 / ms / 1e6:.0f} GB/s")
ms = timeit(lambda: a.sum())
print(f"torch.sum (1 x read): {ms:.3f} ms  {n * 4 / ms / 1e6:.0f} GB/s")
ms = timeit(lambda: b.copy_(a))
print(f"torch copy (read+write): {ms:.3f} ms  {2 * n * 4 / ms / 1e6:.0f} GB/s")
