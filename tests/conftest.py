import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_PATH = os.path.join(ROOT, "tests", "golden", "reference_outputs_v1.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _have_gpu_node():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # On a machine without an AMD GPU device node the gpu tests cannot run at all: skip them.  On a GPU box they
    # are never skipped -- a missing library or a failing hipInit is a loud failure there.
    if _have_gpu_node():
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: not a GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return np.load(GOLDEN_PATH)


@pytest.fixture(scope="session")
def golden_v2():
    """BiCGSymmetric DIVERGED cases as the real reference decides them (oracle/gen_golden_v2.py)"""
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_outputs_v2.npz"))


def bicgsymmetric_cases(golden_v2, dtype):
    dn = np.dtype(dtype).name
    names = sorted({k.split("/")[1] for k in golden_v2.files if k.startswith("bicgsymmetric/") and k.split("/")[2] == dn})
    for name in names:
        tag = f"bicgsymmetric/{name}/{dn}"
        g = golden_v2
        yield name, (g[f"{tag}/start"], g[f"{tag}/positions"], g[f"{tag}/values"]), g[f"{tag}/b"], int(g[f"{tag}/maxit"]), float(g[f"{tag}/eps"]), \
            int(g[f"{tag}/status"]), g[f"{tag}/x"]


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def oracle_fma():
    from oracle.oracle import Oracle

    return Oracle(fma=True)


@pytest.fixture(scope="session")
def reference():
    from oracle.oracle import Reference

    if not Reference.available():
        pytest.skip("oracle/_ref/libsmm_ref.so not built (the reference is only mounted in the build container)")
    return Reference()


@pytest.fixture(scope="session")
def smm():
    """the product package, initialised on device 0 -- gpu tests only"""
    import sparse_matrix_math_amd as smm

    smm.init(0)
    return smm


def kat_matrix(dtype):
    """the 5x4 matrix of test/cpp/csr.cpp:263-275 (row 4 is empty)"""
    start = np.array([0, 2, 5, 7, 10, 10], dtype=np.int32)
    positions = np.array([0, 2, 0, 1, 3, 1, 2, 0, 1, 3], dtype=np.int32)
    values = np.array([4.5, 3.2, 3.1, 2.9, 0.9, 1.7, 3.0, 3.5, 0.4, 1.0], dtype=dtype)
    return start, positions, values


# (mult, lhs, expected rMultAdd, expected rMultSub) -- test/cpp/csr.cpp:314, 354, 448, 500 and the zero cases
KAT_CASES = [
    ([1, 2, 3, 4], [0, 0, 0, 0, 0], [14.1, 12.5, 12.4, 8.3, 0], [-14.1, -12.5, -12.4, -8.3, 0]),
    ([1, 0, 3, 4], [5, 6, 7, 8, 10], [19.1, 12.7, 16.0, 15.5, 10], [-9.1, -0.7, -2.0, 0.5, 10]),
    ([0, 0, 0, 0], [5, 6, 7, 8, 9], [5, 6, 7, 8, 9], [5, 6, 7, 8, 9]),
    ([0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]),
]
