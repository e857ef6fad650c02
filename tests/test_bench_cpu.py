"""bench.py host-side logic that needs no GPU: the self-launch command line, the traffic stamp."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_committed_traffic_measurement_belongs_to_the_committed_kernel():
    """profiles/spmv_traffic.json is stamped with a hash of the SpMV kernel sources (comments stripped); bench.py drops
    `roofline.traffic` when the sources have moved on.  The committed pair must agree, or the next bench line silently loses it."""
    with open(os.path.join(ROOT, "profiles", "spmv_traffic.json")) as f:
        stamp = json.load(f)
    assert stamp["kernel_source_sha16"] == bench.spmv_kernel_source_sha()
    args = argparse.Namespace(rows=10_000_000, dtype="f32", band_k=25)
    t = bench.load_traffic(args, "spmvTileKernel")  # roofline_csr.traffic
    assert t is not None and 3.9e9 < t < 6.0e9  # algorithmic 3.996 GB <= traffic
    # ... and the kernel the timed region runs (AUTO: the PATTERN tile kernel) has an entry of its own, stamped with ITS sources
    pat = stamp["other_kernels"]["pattern_family_same_matrix"]
    assert pat["kernel_source_sha16"] == bench.pattern_kernel_source_sha()
    tp = bench.load_traffic(args, "spmvPatternTileKernel")  # roofline.traffic
    assert tp is not None and 2.1e9 < tp < t  # 2.138 GB of its own bytes <= traffic < the CSR kernel's
    assert bench.load_traffic(args, "spmvStreamKernel") is None  # never another kernel's number


def test_kernel_hash_ignores_comments_and_white_space(tmp_path, monkeypatch):
    src = os.path.join(ROOT, "sparse_matrix_math_amd", "csrc")
    before = bench.spmv_kernel_source_sha()
    fake = tmp_path / "sparse_matrix_math_amd" / "csrc"
    fake.mkdir(parents=True)
    for name in ("smm_spmv.hip", "smm_device.h"):
        text = open(os.path.join(src, name), encoding="utf-8").read()
        (fake / name).write_text("// a new comment\n" + text.replace("\n", "\n  ") + "/* trailing */\n", encoding="utf-8")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.spmv_kernel_source_sha() == before
    (fake / "smm_device.h").write_text((fake / "smm_device.h").read_text(encoding="utf-8") + "int changed;\n", encoding="utf-8")
    assert bench.spmv_kernel_source_sha() != before
