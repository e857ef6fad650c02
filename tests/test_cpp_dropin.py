"""The C++ drop-in header (include/smm_hip/sparse_matrix_math.h): the reference's own hot-path tests re-run against it
(tests/cpp/test_dropin.cpp, built by __graft_entry__.build())."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "tests", "cpp", "test_dropin")


def _build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)


def test_compiles_against_the_reference_api():
    """host-only g++ build of code written like the reference's tests; without a GPU it must refuse to compute"""
    _build()
    assert os.path.exists(BINARY)
    if not os.path.exists("/dev/kfd"):
        r = subprocess.run([BINARY], capture_output=True, text=True, timeout=60)
        assert r.returncode == 77 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_reference_tests_pass_on_gpu():
    if not os.path.exists(BINARY):
        _build()
    r = subprocess.run([BINARY], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failed" in r.stdout
