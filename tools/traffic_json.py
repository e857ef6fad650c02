#!/usr/bin/env python3
"""profiles/spmv_traffic.json from the counter passes of tools/pmc_traffic.sh:   python tools/traffic_json.py gpurun_out/traffic_<tag>

Reads summary_spmv.txt (mean of every counter over the timed launches of the SpMV kernel), applies the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE counts a 128-byte request as 64 bytes; WRITE_SIZE is exact; cross-check with the size-resolved request
counters) and stamps the result with the sha of the kernel source it was measured on (bench.py quotes it only for that source)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    src = sys.argv[1]
    vals, kernel = {}, None
    for line in open(os.path.join(src, "summary_spmv.txt")):
        m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
        elif line.strip() and kernel is None:
            kernel = line.strip()
    import bench

    read_fetch = 2 * vals["FETCH_SIZE"] * 1024
    read_req = 128 * vals.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * vals.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * vals.get("TCC_EA0_RDREQ_32B_sum", 0)
    write = vals["WRITE_SIZE"] * 1024
    def family_bytes(name):
        """the same corrections for another kernel's summary, stamped with the PATTERN sources' sha (bench.py quotes the PATTERN tile kernel's
        entry as roofline.traffic when AUTO runs the bench matrix on it)"""
        path = os.path.join(src, name)
        if not os.path.exists(path):
            return None
        v, k = {}, None
        for line in open(path):
            m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)", line)
            if m:
                v[m.group(1)] = float(m.group(2))
            elif line.strip() and k is None:
                k = line.strip()
        if "FETCH_SIZE" not in v:
            return None
        rq = 128 * v.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0) + 32 * v.get("TCC_EA0_RDREQ_32B_sum", 0)
        return {"kernel": k.replace("void smm::", ""), "kernel_source_sha16": bench.pattern_kernel_source_sha(),
                "read_bytes_per_launch": 2 * v["FETCH_SIZE"] * 1024, "read_bytes_per_launch_from_request_sizes": rq,
                "write_bytes_per_launch": v.get("WRITE_SIZE", 0) * 1024, "TCC_HIT_sum": v.get("TCC_HIT_sum"), "TCC_MISS_sum": v.get("TCC_MISS_sum")}

    out = {
        "rows": 10_000_000, "dtype": "f32", "band_k": 25, "kernel": kernel.replace("void smm::", ""),
        "kernel_source_sha16": bench.spmv_kernel_source_sha(),
        "source": "rocprofv3 --kernel-trace --pmc, one pass per counter group (tools/pmc_traffic.sh), MI355X, mean over the timed launches",
        "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
        "TCC_EA0_RDREQ_128B_sum": vals.get("TCC_EA0_RDREQ_128B_sum"), "TCC_EA0_RDREQ_64B_sum": vals.get("TCC_EA0_RDREQ_64B_sum"),
        "TCC_EA0_RDREQ_32B_sum": vals.get("TCC_EA0_RDREQ_32B_sum"), "TCC_HIT_sum": vals.get("TCC_HIT_sum"), "TCC_MISS_sum": vals.get("TCC_MISS_sum"),
        "correction": "gfx950: FETCH_SIZE counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section), so read bytes = 2 x FETCH_SIZE x 1024; "
                      "cross-checked with the size-resolved request counters. WRITE_SIZE is exact.",
        "read_bytes_per_launch": read_fetch, "read_bytes_per_launch_from_request_sizes": read_req, "write_bytes_per_launch": write,
        "hbm_bytes_per_launch": (read_req if read_req else read_fetch) + write,
        "other_kernels": {"pattern_family_same_matrix": family_bytes("summary_pattern.txt"), "laplacian512_f64_stream": family_bytes("summary_lap_stream.txt"),
                          "laplacian512_f64_pattern": family_bytes("summary_lap_pattern.txt")},
    }
    path = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
