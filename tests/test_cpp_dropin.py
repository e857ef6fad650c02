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


def test_loaders_and_error_reporting_host_only():
    """Matrix Market (general / symmetric / pattern, duplicates, the reference's status codes by name) and dense-text loaders,
    direct-to-CSR == triplet route, and -- without a GPU -- NaN / DIVERGED / lastHipStatus() instead of silently doing nothing"""
    _build()
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_loader")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 failed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_host_side_under_address_and_ub_sanitizers():
    """the same translation unit built with -fsanitize=address,undefined (CPU sanitizer build)"""
    _build()
    # sanitizers never run on the GPU (not supported on this pool): hide every device, so that on a GPU box too the binary takes its
    # no-device path (the HIP runtime then reports zero devices and every ABI call returns SMM_HIP_ERR_NO_DEVICE)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", HIP_VISIBLE_DEVICES="-1",
               ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="-1")
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_loader_asan")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "0 failed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


@pytest.mark.gpu
def test_reference_tests_pass_on_gpu():
    if not os.path.exists(BINARY):
        _build()
    r = subprocess.run([BINARY], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 failed" in r.stdout
