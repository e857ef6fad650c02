#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline: CSR SpMV GB/s (% of HBM peak) + BiCGStab iterations/s on the 10M-row, ~51 nnz/row
banded-random SPD matrix (config 3), fp32, on N GPUs of one node.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one BiCGStab iteration (ref:2232-2277: 2 SpMV + 5 reductions + 3 vector updates) with every operand already
resident in HBM.  The timed region is K iterations run as consecutive smm_hip_bicgstab_dev calls of --iters-per-solve
iterations each (eps = 0, x0 = 0, b = A x_true), bracketed by a barrier and a device synchronise on both sides; it includes
every solve's set-up (r = b - A x, r0 = p = r, one dot) and status read-back.  Rank 0 prints ONE JSON line.  N > 1: rows are range-partitioned across the ranks (strong scaling, the matrix is
the same 10M-row matrix), see sparse_matrix_math_amd/distributed.py.

roofline: the dominant kernel is the SpMV; `achieved` = algorithmic bytes of one SpMV launch
(B_spmv = nnz*(s+4) + (rows+1)*4 + cols*s + rows*s; the fused-dot launches also read one or two more
vectors, which are NOT counted) divided by the average launch duration measured with HIP events inside the timed region.
cpu_baseline: the OpenMP port of the same loop (oracle/, kind "port") on this host's cores, a bounded number of
iterations of the very same matrix copied back from the GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy achieves
METRIC = "CSR SpMV GB/s (% HBM peak) + BiCGStab iters/sec, 10M rows, 1/2/4/8 GPU"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--band-k", type=int, default=25, help="offsets per side: 2k+1 nonzeros per interior row")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--max-offset", type=int, default=1 << 20)
    ap.add_argument("--diag-shift", type=float, default=1.0, help="A[i][i] = shift + sum|offdiag| (SURVEY.md section 8d: 1.0)")
    ap.add_argument("--iters-per-solve", type=int, default=20, help="BiCGStab iterations per solver call inside the timed region")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline budget; 0 disables it")
    ap.add_argument("--spmv-family", type=int, default=0)
    ap.add_argument("--spmv-lanes", type=int, default=0)
    ap.add_argument("--autotune", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra (non-headline) PATTERN-family measurement")
    ap.add_argument("--mtx", default=None, help="extras leg (BASELINE config 5): a Matrix Market file read by the drop-in header's direct-to-CSR loader and "
                    "solved with BiCGStab + none / Jacobi / BLOCK_ILU0 / ILU0 (tests/cpp/mtx_bicgstab); default: ./atmosmodd.mtx when it exists, else a "
                    "generated 1.26 M-row general file with varying coefficients")
    ap.add_argument("--dist", action="store_true", help="take the row-partitioned multi-GPU code path even with one rank")
    ap.add_argument("--dist-driver", choices=["native", "python"], default="native",
                    help="N > 1: the loop behind the C ABI (csrc/smm_dist.hip, RCCL) or the Python driver over torch.distributed")
    return ap.parse_args()


def spmv_bytes(rows, cols, nnz, s):
    return nnz * (s + 4) + (rows + 1) * 4 + cols * s + rows * s


def _source_sha(names):
    """sha256 over the CODE the named kernel sources are compiled from (comments and white space stripped, so that a reworded comment
    does not orphan a measurement): a traffic measurement is only quoted for the kernel it was taken on"""
    import hashlib
    import re

    h = hashlib.sha256()
    for name in names:  # (not smm_internal.h: it changes with every unrelated entry point)
        with open(os.path.join(ROOT, "sparse_matrix_math_amd", "csrc", name), encoding="utf-8") as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # block comments
        text = re.sub(r"//[^\n]*", "", text)                # line comments (no string literal of these files holds "//")
        h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def spmv_kernel_source_sha():
    """the CSR kernels (spmvTileKernel / spmvStreamKernel)"""
    return _source_sha(("smm_spmv.hip", "smm_device.h"))


def pattern_kernel_source_sha():
    """the PATTERN family's kernels"""
    return _source_sha(("smm_spmv_pattern.hip", "smm_pattern_dev.h", "smm_device.h"))


def load_traffic(args, kernel):
    """Fabric-side bytes per launch of `kernel` on the bench matrix from the committed rocprofv3 PMC passes (profiles/spmv_traffic.json;
    tools/pmc_traffic.sh + tools/traffic_json.py write it, one rocprofv3 pass per counter group) -- quoted ONLY when it was measured on
    this workload, on a kernel of this name AND on the kernel source this tree builds (sha stamped at measurement time); otherwise
    null rather than a stale number."""
    path = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if not (t.get("rows") == args.rows and t.get("dtype") == args.dtype and t.get("band_k") == args.band_k):
            return None
        if t.get("kernel", "").startswith(kernel + "<") and t.get("kernel_source_sha16") == spmv_kernel_source_sha():
            return t.get("hbm_bytes_per_launch")
        o = (t.get("other_kernels") or {}).get("pattern_family_same_matrix") or {}
        if o.get("kernel", "").startswith(kernel + "<") and o.get("kernel_source_sha16") == pattern_kernel_source_sha():
            return (o.get("read_bytes_per_launch_from_request_sizes") or o.get("read_bytes_per_launch")) + o.get("write_bytes_per_launch", 0)
    except (OSError, ValueError, TypeError):
        pass
    return None


def host_cores():
    """(all, share): every PHYSICAL core this process may run on (affinity mask clipped by the cgroup CPU quota, SMT siblings counted
    once -- SURVEY.md section 8d: "all physical cores of the GPU host"), and the GPU box's CPU share of 16 cores per GPU.
    SMM_BENCH_CPU_THREADS overrides the first."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    physical = set()
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                physical.add(f.read().strip())
        except OSError:
            physical.add(str(c))
    n = len(physical)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    if os.environ.get("SMM_BENCH_CPU_THREADS"):
        n = max(1, int(os.environ["SMM_BENCH_CPU_THREADS"]))
    return max(1, n), max(1, min(n, 16))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, np_dtype, start, positions, values, b, budget_s):
    """The bounded CPU sample: the GPU leg's own unit of work -- solves of --iters-per-solve BiCGStab iterations from x0 = 0, set-up
    included -- on the same matrix.  Every figure is taken AFTER a warm-up solve of the same thread count (first touch of the
    temporaries, page cache, OpenMP team start-up) and the all-core figure is the median of three samples with their spread printed:
    the GPU boxes' hosts are shared, and single samples of this leg moved by 30 % between rounds."""
    import numpy as np

    from oracle.oracle import Oracle

    oracle = Oracle()
    cores, share = host_cores()
    csr = (start, positions, values)
    x0 = np.zeros(len(b), dtype=np_dtype)
    per_solve = max(1, args.iters_per_solve)

    def solve_rate(fn, iters):
        t0 = time.perf_counter()
        it = fn(iters)
        return it / (time.perf_counter() - t0)

    def port(iters):
        return oracle.bicgstab(csr, b, x0, iters, 0.0, omp=True)[2]

    def samples(threads, n, iters, floor=2):
        oracle.set_threads(threads)
        t0 = time.perf_counter()
        port(1)  # warm-up, untimed (one iteration = 3 SpMVs with the set-up)
        per_it = (time.perf_counter() - t0) * 2 / 3
        iters = int(max(floor, min(iters, budget_s * 0.2 / max(per_it, 1e-3))))  # a sample stays within ~20 % of the budget
        return sorted(solve_rate(port, iters) for _ in range(n)), iters

    rates, iters = samples(cores, 3, per_solve)
    out = {
        "value": rates[1],
        "unit": "iterations/s",
        "cores": cores,
        "kind": "port",
        "sample": f"median of 3 solves of {iters} BiCGStab iterations (set-up SpMV included, like the GPU leg's solves of {per_solve}) of the same {len(b)}-row "
                  f"matrix after a warm-up solve: OpenMP port of the reference loop on all {cores} physical cores",
        "spread": [rates[0], rates[-1]],
        "cpu_model": cpu_model(),
    }
    if share != cores:  # the 16-core share of one GPU of the box
        r16, _ = samples(share, 3, per_solve)
        out["value_16_cores"] = r16[1]
    # one core: the port, and the real reference (oracle/_ref/libsmm_ref.so: built from /root/reference in the build container and
    # DELIBERATELY carried to the GPU box as a binary) -- its own SMM::BiCGStab, single-threaded as its default
    # build is; both after a warm-up call, the same number of iterations (>= 5) each
    one_core_iters = int(max(5, min(10, per_solve)))
    r1, it1 = samples(1, 1, one_core_iters, floor=5)
    oracle.set_threads(cores)
    out["value_1_core"] = r1[0]
    out["one_core_sample"] = f"{it1} iterations after a warm-up solve"
    try:
        from oracle.oracle import Reference

        if Reference.available():
            ref = Reference()
            with ref.csr(csr) as m:
                ref.bicgstab(m, b, x0, 1, 0.0)  # warm-up
                t0 = time.perf_counter()
                ref.bicgstab(m, b, x0, it1, 0.0)
                out["reference_1_core"] = it1 / (time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001 -- the reference leg is optional; the port above is the baseline
        out["reference_error"] = str(e)[:200]
    return out


def extra_spmv_legs(args, smm, host, torch, np, dev, stream):
    """NOT the headline: two more SpMV-only measurements the roofline discussion refers to:
    (a) SURVEY.md section 8d's secondary matrix -- the benchmark matrix's shape with i.i.d. uniform columns, the worst case for the x gather;
    (b) BASELINE config 4's matrix on one GPU -- the 3-D 7-point Laplacian 512^3 in fp64 (13.9 GB per SpMV)."""
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def time_spmv(A, n, td, reps):
        x = torch.rand(n, dtype=td, device=dev)
        y = torch.empty(n, dtype=td, device=dev)
        for _ in range(2):
            A.spmv_dev(0, None, x, y, stream)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            A.spmv_dev(0, None, x, y, stream)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    try:
        n = args.rows
        np_dtype = np.float32 if args.dtype == "f32" else np.float64
        td = torch.float32 if args.dtype == "f32" else torch.float64
        s = np.dtype(np_dtype).itemsize
        nnz = host.gen_banded_nnz(n, args.band_k, args.seed, args.max_offset)
        d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
        d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
        d_val = torch.empty(nnz, dtype=td, device=dev)
        host.gen_banded_dev(n, args.band_k, args.seed, args.max_offset, d_start, d_pos, d_val, np_dtype, stream, diag_shift=args.diag_shift)
        torch.cuda.synchronize()
        # stratified i.i.d. columns: entry j of a row of length L falls uniformly into the j-th of L equal slices of [0, n): ascending
        lens = (d_start[1:] - d_start[:-1]).to(torch.int64)
        j = torch.arange(nnz, device=dev, dtype=torch.int64) - torch.repeat_interleave(d_start[:-1].to(torch.int64), lens)
        width = (n // torch.repeat_interleave(lens, lens)).clamp_(min=1)
        g = torch.Generator(device=dev).manual_seed(7)
        r = (torch.rand(nnz, device=dev, generator=g, dtype=torch.float64) * width.to(torch.float64)).to(torch.int64)
        d_pos.copy_((j * width + torch.minimum(r, width - 1)).clamp_(0, n - 1).to(torch.int32))
        del lens, j, width, r
        A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np_dtype)
        # one lane per row: every family / lane count lands on the same time for this matrix, and the launches
        # then carry a kernel name of their own (spmvStreamKernel<float, 1>) -- a rocprofv3 --stats of this command keeps the headline
        # kernel's average (spmvStreamKernel<float, 2>) clean
        A.set_kernel(2, 1)
        ms = time_spmv(A, n, td, 5)
        bts = spmv_bytes(n, n, nnz, s)
        out["spmv_iid_columns"] = {"rows": n, "nnz": nnz, "dtype": args.dtype, "lanes_per_row": 1, "avg_launch_ms": ms, "gbps": bts / ms / 1e6,
                                   "frac": bts / ms / 1e6 / HBM_PEAK_GBPS,
                                   "note": "same shape and values as the bench matrix, columns i.i.d. uniform: every gather pulls its own 128-byte line"}
        A.close()
        del A, d_start, d_pos, d_val
    except Exception as e:  # noqa: BLE001 -- extras never fail the bench line
        out["spmv_iid_columns"] = {"skipped": str(e)[:200]}
    try:
        N = 512
        n, nnz = N ** 3, host.gen_stencil3d_nnz(N, N, N)
        d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
        d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
        d_val = torch.empty(nnz, dtype=torch.float64, device=dev)
        host.gen_stencil3d_dev(N, N, N, 6.0, -1.0, -1.0, d_start, d_pos, d_val, np.float64, stream)
        A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np.float64)
        # the CSR number (the roofline's layout): STREAM forced -- AUTO would move this 937 M-entry stencil to the
        # index-free PATTERN family on its first SpMV
        A.set_kernel(2, 0)
        ones = torch.ones(n, dtype=torch.float64, device=dev)
        y = torch.empty_like(ones)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        A.spmv_dev(0, None, ones, y, stream)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3  # handle creation is free; this is the tile table + one launch
        del ones, y
        ms = time_spmv(A, n, torch.float64, 5)
        bts = spmv_bytes(n, n, nnz, 8)
        out["spmv_laplacian512_f64"] = {"rows": n, "nnz": nnz, "dtype": "f64", "family": "STREAM", "avg_launch_ms": ms, "gbps": bts / ms / 1e6,
                                        "frac": bts / ms / 1e6 / HBM_PEAK_GBPS, "first_spmv_ms": first_ms}
        # what AUTO does with it: the PATTERN family, with the bytes it really moves (values + 8 bytes per row + vectors + start[])
        try:
            A.set_kernel(3, 0)
            A.pattern_allow_const(False)  # first the MASKS kernel: values[] still read
            p_kernel, p_bytes = A.kernel_desc()  # the library's own account of what that kernel moves (values + 8 bytes of mask per row + start + x + out)
            ms_p = time_spmv(A, n, torch.float64, 5)
            out["spmv_laplacian512_f64"]["pattern_family"] = {"kernel": p_kernel, "avg_launch_ms": ms_p, "true_bytes_per_launch": p_bytes, "gbps": p_bytes / ms_p / 1e6,
                                                              "frac": p_bytes / ms_p / 1e6 / HBM_PEAK_GBPS}
            # ... and what AUTO really runs for a Laplacian: every diagonal holds one value (verified against every entry), so values[]
            # is not read either: the row's mask (32 bits in the 2.5-D kernel), x, y
            A.pattern_allow_const(True)
            if A.pattern_info()[0] == 3:
                c_kernel, c_bytes = A.kernel_desc()
                ms_c = time_spmv(A, n, torch.float64, 10)
                out["spmv_laplacian512_f64"]["const_diagonals"] = {"kernel": c_kernel, "avg_launch_ms": ms_c, "true_bytes_per_launch": c_bytes, "gbps": c_bytes / ms_c / 1e6,
                                                                   "frac": c_bytes / ms_c / 1e6 / HBM_PEAK_GBPS,
                                                                   "note": "PATTERN family, constant-diagonal encoding: what AUTO runs for this matrix"}
        except smm.SmmHipError as e:
            out["spmv_laplacian512_f64"]["pattern_family"] = {"skipped": str(e)[:200]}
        A.close()
        del A, d_start, d_pos, d_val
    except Exception as e:  # noqa: BLE001
        out["spmv_laplacian512_f64"] = {"skipped": str(e)[:200]}
    # (b') a 27-point stencil with constant coefficients -- HPCG's matrix shape -- at 192^3 in fp64: far offsets in clusters, i.e. the
    # three-window march kernel of r05 (AUTO's choice for it) beside the CSR stream; built on the device with torch (no host loop)
    try:
        N = 192
        n = N ** 3
        i = torch.arange(n, device=dev, dtype=torch.int64)
        ix, iy, iz = i % N, (i // N) % N, i // (N * N)
        cols, valid, vals = [], [], []
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    ok = (ix + dx >= 0) & (ix + dx < N) & (iy + dy >= 0) & (iy + dy < N) & (iz + dz >= 0) & (iz + dz < N)
                    cols.append(i + dz * N * N + dy * N + dx)
                    valid.append(ok)
                    vals.append(26.0 if (dx, dy, dz) == (0, 0, 0) else -1.0)
        cols, valid = torch.stack(cols, dim=1), torch.stack(valid, dim=1)
        d_start = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        d_start[1:] = torch.cumsum(valid.sum(dim=1), dim=0)
        d_pos = cols[valid].to(torch.int32)
        d_val = torch.tensor(vals, dtype=torch.float64, device=dev).expand(n, 27)[valid].contiguous()
        d_start = d_start.to(torch.int32)
        nnz = int(d_pos.numel())
        del i, ix, iy, iz, cols, valid
        A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np.float64)
        A.set_kernel(2, 1)
        x = torch.rand(n, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(11)) - 0.5
        y_csr = torch.empty_like(x)
        A.spmv_dev(0, None, x, y_csr, stream)
        ms = time_spmv(A, n, torch.float64, 10)
        bts = spmv_bytes(n, n, nnz, 8)
        leg = {"rows": n, "nnz": nnz, "dtype": "f64", "csr_stream": {"avg_launch_ms": ms, "gbps": bts / ms / 1e6, "frac": bts / ms / 1e6 / HBM_PEAK_GBPS}}
        A.set_kernel(3, 1)  # (one lane per row: what AUTO gives a matrix with constant diagonals -- the reference's summation order)
        k_name, k_bytes = A.kernel_desc()
        y = torch.empty_like(x)
        A.spmv_dev(0, None, x, y, stream)
        torch.cuda.synchronize()
        ms_m = time_spmv(A, n, torch.float64, 20)
        leg["pattern_family"] = {"kernel": k_name, "avg_launch_ms": ms_m, "true_bytes_per_launch": k_bytes, "gbps": k_bytes / ms_m / 1e6,
                                 "frac": k_bytes / ms_m / 1e6 / HBM_PEAK_GBPS, "bit_equal_to_csr_stream": bool(torch.equal(y, y_csr))}
        out["spmv_stencil27_192_f64"] = leg
        A.close()
        del A, d_start, d_pos, d_val, x, y, y_csr
    except Exception as e:  # noqa: BLE001
        out["spmv_stencil27_192_f64"] = {"skipped": str(e)[:200]}
    torch.cuda.empty_cache()
    # (c) BASELINE config 2: CG on the 1000 x 1000 Poisson matrix, fp64, tol 1e-6 -- the register-resident solve (one launch,
    # csrc/smm_resident.hip) and the three-launch loop, wall time of the whole call
    before = host.cg_resident(-1)
    try:
        N = 1000
        n, nnz = N * N, host.gen_poisson2d_nnz(N, N)
        d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
        d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
        d_val = torch.empty(nnz, dtype=torch.float64, device=dev)
        host.gen_poisson2d_dev(N, N, d_start, d_pos, d_val, np.float64, stream)
        A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np.float64)
        ones = torch.ones(n, dtype=torch.float64, device=dev)
        b = torch.empty_like(ones)
        A.spmv_dev(0, None, ones, b, stream)
        leg = {"rows": n, "nnz": nnz, "dtype": "f64", "tol": 1e-6}
        for mode, name in ((host.CG_RESIDENT_AUTO, "auto"), (host.CG_RESIDENT_OFF, "three_launch_loop")):
            host.cg_resident(mode)
            for _ in range(2):
                x = torch.zeros(n, dtype=torch.float64, device=dev)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st, it, _res = host.cg_dev(A, b, x, x, -1, 1e-6, None, stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            leg[name] = {"status": int(st), "iterations": it, "ms": dt * 1e3, "us_per_iteration": dt / max(it, 1) * 1e6,
                         "max_abs_err_vs_ones": float((x - 1).abs().max())}
        out["cg_poisson1000_f64"] = leg
        A.close()
        del A, d_start, d_pos, d_val
    except Exception as e:  # noqa: BLE001
        out["cg_poisson1000_f64"] = {"skipped": str(e)[:200]}
    host.cg_resident(before)
    torch.cuda.empty_cache()
    # (d) BASELINE config 5's stand-in: BiCGStab on the non-symmetric convection-diffusion matrix 108^3 (1.26 M rows, fp64) to 1e-8, without a
    # preconditioner and with the library's Jacobi / ILU0 / BLOCK_ILU0 / BLOCK_SGS (create time, solve time, time of one apply;
    # DESIGN.md section 3.5)
    try:
        N = 108
        n, nnz = N ** 3, host.gen_stencil3d_nnz(N, N, N)
        d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
        d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
        d_val = torch.empty(nnz, dtype=torch.float64, device=dev)
        host.gen_stencil3d_dev(N, N, N, 6.0, -1.3, -0.7, d_start, d_pos, d_val, np.float64, stream)
        A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np.float64)
        ones = torch.ones(n, dtype=torch.float64, device=dev)
        b = torch.empty_like(ones)
        A.spmv_dev(0, None, ones, b, stream)
        leg = {"rows": n, "nnz": nnz, "dtype": "f64", "tol": 1e-8}
        P = smm.SolverPreconditioner
        bytes_apply = 2 * (nnz * 12 + (n + 1) * 4) + 5 * n * 8  # two triangular sweeps over A's pattern ~ 2 x SpMV bytes
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # block_*: the defaults -- bricks of the grid as blocks, level cut 16; *_uncut: the same blocks without the cut (M = the
        # block-diagonal part of A exactly); *_contiguous: runs of consecutive rows as blocks (what a matrix that is no grid stencil gets)
        # *_values_read: the same legs with the constant-diagonal SpMV encoding turned off -- this stand-in has constant coefficients, so
        # the solvers' SpMV reads no values[] (PATTERN / CONST); a general matrix such as atmosmodd would run the kernel that does
        for name, kind, cap in (("none", None, None), ("jacobi", P.JACOBI, None), ("ilu0", P.ILU0, None), ("block_ilu0", P.BLOCK_ILU0, None),
                                ("block_sgs", P.BLOCK_SGS, None), ("block_ilu0_uncut", P.BLOCK_ILU0, 0), ("block_ilu0_contiguous", P.BLOCK_ILU0, None),
                                ("none_values_read", None, None), ("block_ilu0_values_read", P.BLOCK_ILU0, None)):
            A.pattern_allow_const(not name.endswith("_values_read"))
            M, tc = None, 0.0
            for _ in range(2 if kind is not None else 0):  # the second create is the steady state (device allocations cached)
                if M is not None:
                    M.close()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                M = A.getPreconditioner(kind, None, cap, 1 if name.endswith("_contiguous") else None)
                torch.cuda.synchronize()
                tc = time.perf_counter() - t0
            for _ in range(2):
                x = torch.zeros(n, dtype=torch.float64, device=dev)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                st, it, _res = host.bicgstab_dev(A, b, x, -1, 1e-8, M, stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            leg[name] = {"status": int(st), "iterations": it, "solve_ms": dt * 1e3, "create_ms": tc * 1e3, "create_plus_solve_ms": (tc + dt) * 1e3,
                         "max_abs_err_vs_ones": float((x - 1).abs().max())}
            if M is not None:
                if kind != P.JACOBI:
                    y = torch.empty_like(ones)
                    M.apply_dev(b, y, stream)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(20):
                        M.apply_dev(b, y, stream)
                    e1.record()
                    torch.cuda.synchronize()
                    ms_apply = e0.elapsed_time(e1) / 20
                    leg[name].update(apply_us=ms_apply * 1e3, apply_gbps=bytes_apply / ms_apply / 1e6, levels=list(M.levels()))
                    if kind in (P.BLOCK_ILU0, P.BLOCK_SGS):
                        leg[name]["blocks"] = len(M.block_bounds()) - 1
                        leg[name]["level_cap"] = M.level_cap()
                        rec = M.block_record_bytes()  # what the apply really moves: both sweeps' records, rhs, x (the fused dot's w1 is not read here)
                        true_b = n * (sum(rec) + 2 * 8)
                        leg[name].update(apply_true_bytes=true_b, apply_true_gbps=true_b / ms_apply / 1e6, apply_true_frac=true_b / ms_apply / 1e6 / HBM_PEAK_GBPS)
                        leg[name]["brick"] = list(M.block_rows()[1])  # blocks = bricks of the grid of this many points per axis ([0, 0, 0]: runs of consecutive rows)
                M.close()
        A.pattern_allow_const(True)
        leg["spmv"] = {"family_lanes": list(A.get_kernel()), "pattern_encoding": A.pattern_info()[0],
                       "note": "encoding 3 = row masks + constant diagonals (no values[] read), 1 = row masks + values[]"}
        out["bicgstab_convdiff108_f64"] = leg
        A.close()
        del A, d_start, d_pos, d_val
    except Exception as e:  # noqa: BLE001
        out["bicgstab_convdiff108_f64"] = {"skipped": str(e)[:200]}
    torch.cuda.empty_cache()
    # (d) what ONE RANK of BASELINE config 4 at 8 GPUs computes per ConjugateGradient iteration: a 512 x 512 x 64 slab of the Laplacian through the
    # row-partitioned loop (csrc/smm_dist.hip distCg) on a single-rank communicator -- with nothing remote, and with the last plane's columns counted
    # as another rank's (SMM_HIP_LAB_SELF_SPLIT: A_rem holds what a neighbouring slab would own; nothing travels).  DESIGN.md section 4's config-4 table.
    try:
        from sparse_matrix_math_amd.distributed import NativeComm, NativeDistMatrix

        nx, ny, nz = 512, 512, 64
        n, nnz = nx * ny * nz, host.gen_stencil3d_nnz(nx, ny, nz)
        d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
        d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
        d_val = torch.empty(nnz, dtype=torch.float64, device=dev)
        host.gen_stencil3d_dev(nx, ny, nz, 6.0, -1.0, -1.0, d_start, d_pos, d_val, np.float64, stream)
        torch.cuda.synchronize()
        comm = NativeComm.single()
        leg = {"rows": n, "dtype": "f64", "note": "microseconds per CG iteration of one rank's slab (100 iterations, best of 3); not the headline"}
        ones = torch.ones(n, dtype=torch.float64, device=dev)
        b = None
        for name, window in (("nothing_remote", 0), ("a_remote_plane", -nx * ny)):
            os.environ["SMM_HIP_LAB_SELF_SPLIT"] = str(window)
            try:
                A = NativeDistMatrix(comm, n, [0, n], d_start, d_pos, d_val, np.float64)
            finally:
                os.environ.pop("SMM_HIP_LAB_SELF_SPLIT")
            if b is None:
                b = torch.empty_like(ones)
                A.spmv(0, None, ones, b)
            x = torch.zeros_like(ones)
            A.cg(b, x, x, 10, 0.0)
            best = None
            for _ in range(3):
                x.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, iters, _ = A.cg(b, x, x, 100, 0.0)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / max(1, iters) * 1e6
                best = dt if best is None else min(best, dt)
            leg[name] = {"us_per_iteration": best, "nnz_remote": A.nnz_rem, "thin_remote_rows": A.thin_remote()[0], "spmvs_that_formed_p": A.cg_fused()}
            A.close()
        comm.close()
        out["dist_cg_slab512x512x64_f64"] = leg
        del d_start, d_pos, d_val, ones, b, x
    except Exception as e:  # noqa: BLE001
        out["dist_cg_slab512x512x64_f64"] = {"skipped": str(e)[:200]}
    torch.cuda.empty_cache()
    return out


def mtx_leg(path=None):
    """BASELINE config 5 -- "Matrix Market nonsymmetric (e.g. SuiteSparse atmosmodd), BiCGStab + Jacobi/ILU0" -- FROM A FILE, through the C++
    drop-in header (tests/cpp/mtx_bicgstab.cpp: SMM::loadMatrix direct to CSR, then the C ABI's BiCGStab): not the headline.  Without
    --mtx (atmosmodd cannot be fetched offline) the file is written here, once per run, into a temporary directory: the 108^3
    convection-diffusion operator with SPATIALLY VARYING coefficients (generators.convdiff3d_varying; 1.26 M rows, 8.75 M entries, entries
    shuffled, `general` banner) -- a matrix whose diagonals are not constant, so the SpMV reads values[] as it would for atmosmodd."""
    import shutil
    import subprocess
    import tempfile

    tool = os.path.join(ROOT, "tests", "cpp", "mtx_bicgstab")
    if not os.path.exists(tool):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "mtx_bicgstab"], check=True)
    tmp = None
    out = {}
    try:
        if path is None:
            import numpy as np

            from sparse_matrix_math_amd import generators as gen

            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from write_mtx import write_mtx

            tmp = tempfile.mkdtemp(prefix="smm_bench_mtx_")
            path = os.path.join(tmp, "convdiff_varying108.mtx")
            t0 = time.perf_counter()
            entries = write_mtx(path, gen.convdiff3d_varying(108, 0.3, dtype=np.float64), shuffle=True, seed=1)
            out.update(generated="generators.convdiff3d_varying(108, 0.3): general, shuffled", write_s=time.perf_counter() - t0, entries=entries,
                       file_bytes=os.path.getsize(path))
        out["file"] = os.path.basename(path)
        r = subprocess.run([tool, path, "none,jacobi,block_ilu0,ilu0", "2000", "1e-8"], capture_output=True, text=True, timeout=1800)
        if r.returncode != 0:
            out["error"] = (r.stderr or r.stdout)[-300:]
            return out
        for ln in r.stdout.strip().splitlines():
            j = json.loads(ln)
            out.update(rows=j["rows"], nnz=j["nnz"], load_s=j["load_s"], spmv_kernel=j["spmv_kernel"], pattern_encoding=j["pattern_encoding"])
            out[j["precond"]] = {"status": j["status"], "iterations": j["iterations"], "resnorm": j["resnorm"], "create_ms": j["precond_setup_s"] * 1e3,
                                 "solve_ms": j["solve_s"] * 1e3, "create_plus_solve_ms": (j["precond_setup_s"] + j["solve_s"]) * 1e3,
                                 "max_abs_err_vs_ones": j["max_abs_err_vs_ones"]}
        out["note"] = ("host-pointer API of the drop-in header: every solve_ms includes the PCIe copies of b and x; create / solve are the second "
                       "(steady-state) run of each kind; pattern_encoding 1 = row masks + values[]")
    except Exception as e:  # noqa: BLE001 -- extras never fail the bench line
        out["error"] = str(e)[:300]
    finally:
        if tmp:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as FRESH child processes -- before
    this process has imported torch or made any GPU call -- relay their output (rank 0 prints the JSON line) and return
    their exit status.  Nothing is ever exec'ed from a process that has touched the GPU."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")  # torch.distributed.run would set (and warn about) it anyway
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # SMM_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices, halos
    # are staged through host memory) -- a functional check of this file's N > 1 leg, never a measurement
    backend = os.environ.get("SMM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import sparse_matrix_math_amd as smm
    from sparse_matrix_math_amd import host

    smm.init(local_rank)
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    t_dtype = torch.float32 if args.dtype == "f32" else torch.float64
    s_bytes = 4 if args.dtype == "f32" else 8
    n = args.rows
    stream = torch.cuda.current_stream().cuda_stream

    if use_dist:
        from sparse_matrix_math_amd import distributed as dsm

        result = dsm.bench_bicgstab(args, rank, world, dev, np_dtype, t_dtype)
        if rank == 0 and args.cpu_seconds > 0:
            # the CPU baseline belongs to the WORKLOAD, not to the partition: rank 0 builds the whole matrix once more on its GPU, hands it
            # to the host and times the bounded sample exactly as the one-GPU run does (the other ranks wait in the all-reduce below)
            nnz = host.gen_banded_nnz(n, args.band_k, args.seed, args.max_offset)
            f_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
            f_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
            f_val = torch.empty(nnz, dtype=t_dtype, device=dev)
            host.gen_banded_dev(n, args.band_k, args.seed, args.max_offset, f_start, f_pos, f_val, np_dtype, stream, diag_shift=args.diag_shift)
            F = smm.CSRMatrix.from_device(n, n, f_start, f_pos, f_val, np_dtype)
            f_x = torch.rand(n, dtype=t_dtype, device=dev, generator=torch.Generator(device=dev).manual_seed(args.seed)) + 0.5
            f_b = torch.empty(n, dtype=t_dtype, device=dev)
            F.spmv_dev(0, None, f_x, f_b, stream)
            torch.cuda.synchronize()
            result["cpu_baseline"] = cpu_baseline(args, np_dtype, f_start.cpu().numpy(), f_pos.cpu().numpy(), f_val.cpu().numpy(), f_b.cpu().numpy(),
                                                  args.cpu_seconds)
            F.close()
            del F, f_start, f_pos, f_val, f_x, f_b
    else:
        nnz = host.gen_banded_nnz(n, args.band_k, args.seed, args.max_offset)
        d_start = torch.empty(n + 1, dtype=torch.int32, device=dev)
        d_pos = torch.empty(nnz, dtype=torch.int32, device=dev)
        d_val = torch.empty(nnz, dtype=t_dtype, device=dev)
        host.gen_banded_dev(n, args.band_k, args.seed, args.max_offset, d_start, d_pos, d_val, np_dtype, stream, diag_shift=args.diag_shift)
        A = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np_dtype)
        if args.autotune:
            A.autotune()
        elif args.spmv_family or args.spmv_lanes:
            A.set_kernel(args.spmv_family or 2, args.spmv_lanes)
        # b = A * x_true with x_true uniform in [0.5, 1.5).  (The reference's test convention b = A*1 is degenerate on this
        # matrix: every row sums to diag_shift, so 1 is an eigenvector and any Krylov method converges in one step.)
        x_true = torch.rand(n, dtype=t_dtype, device=dev, generator=torch.Generator(device=dev).manual_seed(args.seed)) + 0.5
        b = torch.empty(n, dtype=t_dtype, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        A.spmv_dev(0, None, x_true, b, stream)  # the matrix's FIRST SpMV: tile table and -- AUTO, large matrix -- the PATTERN analysis
        torch.cuda.synchronize()
        first_spmv_ms = (time.perf_counter() - t0) * 1e3
        family, lanes = A.get_kernel()  # what AUTO settled on (SMM_SPMV_PATTERN when the matrix passed the verification)
        # the same on a SECOND handle over the same arrays: the per-matrix set-up alone (analysis, verification of every entry, tile table),
        # with whatever the process pays once (code objects, allocator warm-up) already paid by the first
        A2 = smm.CSRMatrix.from_device(n, n, d_start, d_pos, d_val, np_dtype)
        y2 = torch.empty(n, dtype=t_dtype, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        A2.spmv_dev(0, None, x_true, y2, stream)
        torch.cuda.synchronize()
        second_handle_first_spmv_ms = (time.perf_counter() - t0) * 1e3
        A2.close()
        del A2, y2
        x = torch.zeros(n, dtype=t_dtype, device=dev)
        torch.cuda.synchronize()

        # BiCGStab on this expander-like matrix reaches fp32 round-off in ~25 iterations, after which the recursive residual
        # underflows to exactly 0 and the reference's loop (while resL2Norm > eps) stops.  K timed iterations are therefore
        # run as consecutive solves of `iters_per_solve` iterations each, every one from x0 = 0 (set-up included in the time).
        def run(total):
            done = 0
            last = None
            while done < total:
                it = min(args.iters_per_solve, total - done)
                x.zero_()
                status, iters, resnorm = host.bicgstab_dev(A, b, x, it, 0.0, None, stream)
                if iters != it or not np.isfinite(resnorm) or resnorm <= 0:
                    raise SystemExit(f"BiCGStab ran {iters} of {it} iterations (resnorm {resnorm}): the timed region is invalid")
                done += iters
                last = (status, resnorm)
            return done, last

        def timed_leg():
            """W untimed + K timed iterations with the matrix's current kernel: (iterations, seconds, summed SpMV ms, SpMV launches, resnorm)"""
            if args.warmup > 0:
                run(args.warmup)
            host.profile_enable(True)
            host.profile_read(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters, (_status, resnorm) = run(args.steps)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            spmv_ms, spmv_launches = host.profile_read(reset=True)
            host.profile_enable(False)
            return iters, elapsed, spmv_ms, spmv_launches, resnorm

        def roofline_of(kernel, nbytes, spmv_ms, spmv_launches, where):
            """the roofline object of one timed leg: the bytes ONE launch of `kernel` moves by its own layout (smm_hip_csr_kernel_desc) over
            the average launch time measured live with HIP events on the launch stream inside that leg"""
            avg_s = spmv_ms * 1e-3 / max(spmv_launches, 1)
            achieved = nbytes / avg_s / 1e9
            r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                 "traffic": load_traffic(args, kernel), "kernel": kernel, "algorithmic_bytes_per_launch": nbytes,
                 "avg_launch_ms": avg_s * 1e3, "launches": spmv_launches, "measured_in": where}
            if r["traffic"]:
                # measured fabric-side bytes (PMC, profiles/spmv_traffic.json) over the live launch time: how hard the memory system is
                # actually driven -- the gap to `achieved` is x[] cache lines re-fetched from beyond L2
                r["traffic_gbps"] = r["traffic"] / avg_s / 1e9
            return r

        # THE timed region: the library as a user gets it (AUTO).  value, ms_per_step and roofline all describe THIS leg: roofline is its
        # dominant kernel priced with the bytes that kernel's own data layout moves -- never with another layout's
        kernel, kernel_bytes = A.kernel_desc()
        iters, elapsed, spmv_ms, spmv_launches, resnorm = timed_leg()
        err = float(((x - x_true).abs() / x_true).max())
        b_spmv = spmv_bytes(n, n, nnz, s_bytes)
        chosen_by = "autotune" if args.autotune else "forced" if (args.spmv_family or args.spmv_lanes) else "auto"
        result = {"elapsed": elapsed, "iters": iters, "nnz": nnz, "resnorm": float(resnorm), "max_rel_err_vs_x_true": err,
                  "spmv_kernel": {"family": family, "lanes_per_row": lanes, "kernel": kernel, "chosen_by": chosen_by, "first_spmv_ms": first_spmv_ms,
                                  "second_handle_first_spmv_ms": second_handle_first_spmv_ms, "pattern_encoding": A.pattern_info()[0]}}
        result["roofline"] = roofline_of(kernel, kernel_bytes, spmv_ms, spmv_launches, "the timed region")
        if family == 3:
            result["roofline"]["note"] = ("PATTERN family chosen by AUTO on the first SpMV: positions[] replaced by one verified 64-bit mask per row "
                                          "(bit-identical to STREAM at equal lanes); bytes = values + masks + start + x + out, NOT the CSR formula")
            # The metric names "CSR SpMV GB/s (% HBM peak)": that contract -- the reference's CSR layout on the kernel that streams it
            #  -- is measured in a leg of its own with the STREAM family forced: same matrix, same K iterations
            A.set_kernel(2, 0)
            s_kernel, s_bytes_launch = A.kernel_desc()
            s_iters, s_elapsed, s_ms, s_launches, _s_res = timed_leg()
            csr = roofline_of(s_kernel, s_bytes_launch, s_ms, s_launches, "a leg of its own with the STREAM family forced (same matrix, same K iterations)")
            csr.update(value=s_iters / s_elapsed, value_unit="iterations/s", ms_per_step=s_elapsed / s_iters * 1e3, lanes_per_row=A.get_kernel()[1],
                       max_rel_err_vs_x_true=float(((x - x_true).abs() / x_true).max()))
            A.set_kernel(0, 0)  # back to AUTO's choice
            result["roofline_csr"] = csr
        else:
            result["roofline_csr"] = dict(result["roofline"], value=iters / elapsed, value_unit="iterations/s", ms_per_step=elapsed / iters * 1e3)
        assert result["roofline_csr"]["algorithmic_bytes_per_launch"] == b_spmv or result["roofline_csr"]["kernel"] == "spmvVectorKernel"
        if not args.no_extras:
            result.setdefault("extras", {}).update(extra_spmv_legs(args, smm, host, torch, np, dev, stream))
        mtx = args.mtx or (os.path.join(ROOT, "atmosmodd.mtx") if os.path.exists(os.path.join(ROOT, "atmosmodd.mtx")) else None)
        if mtx or not args.no_extras:
            torch.cuda.empty_cache()
            result.setdefault("extras", {})["mtx_bicgstab"] = mtx_leg(mtx)  # config 5 on a file, every run
        if args.cpu_seconds > 0:
            result["cpu_baseline"] = cpu_baseline(args, np_dtype, d_start.cpu().numpy(), d_pos.cpu().numpy(), d_val.cpu().numpy(),
                                                  b.cpu().numpy(), args.cpu_seconds)

    if use_dist:
        t = torch.tensor([result["elapsed"]], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        result["elapsed"] = float(t.item())
    if rank == 0:
        elapsed = result.pop("elapsed")
        iters = result.pop("iters")
        line = {
            "metric": METRIC,
            "value": iters / elapsed,
            "unit": "iterations/s",
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / iters * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"configs[2]: banded-random symmetric diagonally dominant CSR, {n / 1e6:g}M rows ~49 nnz/row, BiCGStab, b = A*x_true, x0 = 0, eps = 0 (fixed iterations)",
                "rows": n,
                "nnz": result.pop("nnz"),
                "band_offsets_per_side": args.band_k,
                "diag_shift": args.diag_shift,
                "iterations_per_solve": args.iters_per_solve,
                "seed": hex(args.seed),
                "partition": f"rows/{args.gpus}",
                "spmv_family": {2: "STREAM", 3: "PATTERN (chosen by AUTO: positions[] replaced by a verified per-row mask; set-up in spmv_kernel.first_spmv_ms)",
                                1: "VECTOR"}.get(result.get("spmv_kernel", {}).get("family"), "per rank: see distributed"),
            },
        }
        line.update(result)
        if backend != "nccl":
            line["rehearsal_backend"] = backend  # ranks share GPUs and halos go through the host: not a measurement
        if "roofline_csr" in line:  # the metric's first half: CSR SpMV GB/s and % of HBM peak -- always the CSR layout's figure
            line["spmv_gbps"] = line["roofline_csr"]["achieved"]
            line["spmv_pct_hbm_peak"] = 100.0 * line["roofline_csr"]["frac"]
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
