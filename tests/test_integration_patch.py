"""INTEGRATION.md section B compiled for real: tests/integration/smm_with_hip.patch -- the `#if defined(SMM_WITH_HIP)`
hooks a maintainer of the reference would add, insertions only, no line of the reference in it -- is applied to a temporary copy of the
mounted reference header, and tests/integration/caller.cpp (a program written against the reference's API) is built against it with
clang++ -fdelayed-template-parsing, with and without -DSMM_WITH_HIP, and run.  Build container only: the reference is not on the GPU box
(the test skips there); here there is no GPU, so the hooked build must report SMM_HIP_ERR_NO_DEVICE through every hook -- NaN / DIVERGED,
never a silent CPU answer -- while the unhooked build of the SAME patched header gives the reference's own results."""
import hashlib
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/include/sparse_matrix_math.h"
PATCH = os.path.join(ROOT, "tests", "integration", "smm_with_hip.patch")
CALLER = os.path.join(ROOT, "tests", "integration", "caller.cpp")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
LIB = os.path.join(ROOT, "sparse_matrix_math_amd", "lib")


def test_patch_holds_only_lines_of_this_repository():
    """zero-context diff: comments, file labels, hunk headers and inserted lines -- nothing removed, no context"""
    body = [ln for ln in open(PATCH, encoding="utf-8").read().splitlines() if not ln.startswith("#")]
    assert body[0].startswith("--- ") and body[1].startswith("+++ ")
    hunks = 0
    for ln in body[2:]:
        if ln.startswith("@@"):
            assert re.fullmatch(r"@@ -\d+,0 \+\d+(,\d+)? @@", ln), ln  # an insertion: zero lines of the original
            hunks += 1
        else:
            assert ln.startswith("+"), ln
    assert hunks >= 10


@pytest.fixture(scope="module")
def patched_dir(tmp_path_factory):
    if not os.path.exists(REFERENCE):
        pytest.skip("the reference is only mounted in the build container")
    if not (os.path.exists(CLANG) and shutil.which("patch")):
        pytest.skip("needs clang++ and patch")
    sha = re.search(r"sha256 ([0-9a-f]{64})", open(PATCH, encoding="utf-8").read()).group(1)
    if hashlib.sha256(open(REFERENCE, "rb").read()).hexdigest() != sha:
        pytest.skip("the mounted reference is not the version the patch's line numbers belong to")
    d = tmp_path_factory.mktemp("smm_with_hip")
    os.makedirs(d / "include")
    shutil.copy(REFERENCE, d / "include" / "sparse_matrix_math.h")  # a temporary copy outside the repository, deleted with the session
    r = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return d


def build_and_run(patched_dir, with_hip):
    exe = patched_dir / ("caller_hip" if with_hip else "caller_cpu")
    cmd = [CLANG, "-std=c++17", "-O1", "-fdelayed-template-parsing", "-ffp-contract=off", "-w", f"-I{patched_dir / 'include'}", f"-I{os.path.join(ROOT, 'include')}",
           CALLER, "-o", str(exe)]
    if with_hip:
        cmd += ["-DSMM_WITH_HIP", f"-L{LIB}", "-lsmm_hip", f"-Wl,-rpath,{LIB}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {}
    for ln in r.stdout.splitlines():
        key = ln.split(" status ")[0].split(" b0 ")[0].split(" rc ")[0] if not ln.startswith("dot") else "dot"
        out[key] = ln
    return out


def test_patched_header_without_the_switch_is_the_reference(patched_dir):
    out = build_and_run(patched_dir, with_hip=False)
    assert out["rMult"].startswith("rMult b0 1 b1 0 ")
    assert out["dot"].startswith("dot 2 ")
    for key in ("cg", "bicgstab", "bicgstab+sgs"):
        assert " status 0 x0 1.000000000" in out[key], out[key]
    assert out["sgs apply"].startswith("sgs apply rc 0 ")
    assert out["rMult after edit"].startswith("rMult after edit b0 3 ")


def test_hooks_compile_link_and_report_the_missing_gpu(patched_dir):
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: this test is about the no-device reporting of the hooks")
    if not os.path.exists(os.path.join(LIB, "libsmm_hip.so")):
        pytest.skip("libsmm_hip.so not built")
    out = build_and_run(patched_dir, with_hip=True)
    no_device = " hip -3"  # SMM_HIP_ERR_NO_DEVICE
    assert "nan" in out["rMult"].lower() and out["rMult"].endswith(no_device), out["rMult"]
    assert "nan" in out["dot"].lower() and out["dot"].endswith(no_device)
    for key in ("cg", "bicgstab", "bicgstab+sgs"):
        assert " status 1 " in out[key] and out[key].endswith(no_device), out[key]  # DIVERGED + the reason beside it
    assert out["sgs apply"].startswith("sgs apply rc 1 ") and out["sgs apply"].endswith(no_device)
    assert "nan" in out["rMult after edit"].lower()
