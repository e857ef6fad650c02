"""ctypes binding of libsmm_hip.so (the C ABI declared in include/smm_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at import of the symbols, and
if there is no HIP device every call returns SMM_HIP_ERR_NO_DEVICE, which `check` turns into SmmHipError.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))

SMM_HIP_OK = 0
SMM_HIP_ERR_INVALID = -1
SMM_HIP_ERR_HIP = -2
SMM_HIP_ERR_NO_DEVICE = -3
SMM_HIP_ERR_PRECOND = -4
SMM_HIP_ERR_NOMEM = -5
SMM_HIP_ERR_COMM = -6


class SmmHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libsmm_hip error {code}: {message}")
        self.code = code


def library_path(fma=False):
    if os.environ.get("SMM_HIP_LIBRARY"):  # A/B measurements of two builds of the library (tools/): never set in production
        return os.environ["SMM_HIP_LIBRARY"]
    name = "libsmm_hip_fma.so" if fma else "libsmm_hip.so"
    return os.path.join(_HERE, "lib", name)


_P = c_void_p  # opaque handles and device pointers travel as void*

# callbacks of the host-staged communicator (smm_hip_comm_create_host)
HOST_ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int, c_int)
HOST_SENDRECV_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, POINTER(c_int), POINTER(c_void_p), POINTER(c_size_t), c_int, POINTER(c_int),
                                    POINTER(c_void_p), POINTER(c_size_t))

# name -> (restype, argtypes); {T} expands to float/double for the _f32/_f64 twins
_TYPED = {
    "smm_hip_csr_create": (c_int, [c_int, c_int, _P, _P, _P, POINTER(_P)]),
    "smm_hip_csr_create_dev": (c_int, [c_int, c_int, _P, _P, _P, POINTER(_P)]),
    "smm_hip_spmv": (c_int, [_P, c_int, _P, _P, _P]),
    "smm_hip_spmv_dev": (c_int, [_P, c_int, _P, _P, _P, _P]),
    "smm_hip_dot": (c_int, [c_int, _P, _P, _P]),
    "smm_hip_dot_dev": (c_int, [c_int, _P, _P, _P, _P]),
    "smm_hip_axpy_dev": (c_int, [c_int, "T", _P, _P, _P, _P]),
    "smm_hip_cg": (c_int, [_P, _P, _P, _P, c_int, "T", _P, POINTER(c_int), POINTER(c_int), "PT"]),
    "smm_hip_cg_dev": (c_int, [_P, _P, _P, _P, c_int, "T", _P, _P, POINTER(c_int), POINTER(c_int), "PT"]),
    "smm_hip_bicgstab": (c_int, [_P, _P, _P, c_int, "T", _P, POINTER(c_int), POINTER(c_int), "PT"]),
    "smm_hip_bicgstab_dev": (c_int, [_P, _P, _P, c_int, "T", _P, _P, POINTER(c_int), POINTER(c_int), "PT"]),
    "smm_hip_bicgstab_functor": (c_int, [_P, _P, _P, c_int, "T", "APPLY", _P, POINTER(c_int), POINTER(c_int), "PT"]),
    "smm_hip_bicgsymmetric": (c_int, [_P, _P, _P, c_int, "T", POINTER(c_int), POINTER(c_int)]),
    "smm_hip_precond_apply": (c_int, [_P, _P, _P]),
    "smm_hip_precond_apply_dev": (c_int, [_P, _P, _P, _P]),
    "smm_hip_precond_apply_spmv": (c_int, [_P, _P, _P]),
    "smm_hip_precond_apply_spmv_dev": (c_int, [_P, _P, _P, _P]),
    "smm_hip_precond_values": (c_int, [_P, _P, c_size_t]),
    "smm_hip_gen_poisson2d_dev": (c_int, [c_int, c_int, _P, _P, _P, _P]),
    "smm_hip_gen_stencil3d_dev": (c_int, [c_int, c_int, c_int, "T", "T", "T", _P, _P, _P, _P]),
    "smm_hip_gen_banded_dev": (c_int, [c_int, c_int, c_ulonglong, c_int, "T", _P, _P, _P, _P]),
    "smm_hip_gen_banded_rows_dev": (c_int, [c_int, c_int, c_ulonglong, c_int, "T", c_int, c_int, _P, _P, _P, _P]),
    "smm_hip_spmv_fused_dev": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, _P, _P]),
    "smm_hip_spmv_fused_finish_dev": (c_int, [_P, c_int, _P, _P, _P, c_int, _P, _P, _P]),
    "smm_hip_bicgstab_ws_create": (c_int, [c_int, POINTER(_P)]),
    "smm_hip_bicgstab_ws_stage": (c_int, [_P, c_int, _P, "T", _P]),
    "smm_hip_cg_ws_stage": (c_int, [_P, c_int, _P, _P, "T", _P]),
    "smm_hip_bicgstab_ws_result": (c_int, [_P, _P, POINTER(c_int), POINTER(c_int), "PT"]),
    "smm_hip_dist_csr_create_dev": (c_int, [_P, c_int, POINTER(c_int), _P, _P, _P, POINTER(_P)]),
    "smm_hip_dist_spmv_dev": (c_int, [_P, c_int, _P, _P, _P, _P]),
    "smm_hip_dist_bicgstab_dev": (c_int, [_P, _P, _P, c_int, "T", _P, _P, POINTER(c_int), POINTER(c_int), "PT"]),
    "smm_hip_dist_cg_dev": (c_int, [_P, _P, _P, _P, c_int, "T", _P, POINTER(c_int), POINTER(c_int), "PT"]),
}



_PLAIN = {
    "smm_hip_init": (c_int, [c_int]),
    "smm_hip_shutdown": (c_int, []),
    "smm_hip_last_error": (c_char_p, []),
    "smm_hip_uses_std_fma": (c_int, []),
    "smm_hip_device_info": (c_int, [c_char_p, c_size_t, POINTER(c_int), POINTER(c_size_t)]),
    "smm_hip_stream_synchronize": (c_int, [_P]),
    "smm_hip_debug_fail_next_alloc": (c_int, [c_size_t]),
    "smm_hip_profile_enable": (c_int, [c_int]),
    "smm_hip_profile_read": (c_int, [POINTER(c_double), POINTER(c_longlong), c_int]),
    "smm_hip_profile_read_waits": (c_int, [POINTER(c_double), POINTER(c_longlong), c_int]),
    "smm_hip_csr_destroy": (c_int, [_P]),
    "smm_hip_csr_info": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "smm_hip_csr_set_kernel": (c_int, [_P, c_int, c_int]),
    "smm_hip_csr_get_kernel": (c_int, [_P, POINTER(c_int), POINTER(c_int)]),
    "smm_hip_csr_autotune": (c_int, [_P]),
    "smm_hip_csr_tile_info": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "smm_hip_csr_pattern_info": (c_int, [_P, POINTER(c_int), POINTER(c_int)]),
    "smm_hip_csr_kernel_desc": (c_int, [_P, c_char_p, c_int, POINTER(c_longlong)]),
    "smm_hip_set_march_min_rows": (c_int, [c_longlong, c_longlong]),
    "smm_hip_set_cg_lazy_x_min_bytes": (c_int, [c_longlong]),
    "smm_hip_set_cg_fuse_p": (c_int, [c_int]),
    "smm_hip_csr_pattern_allow_const": (c_int, [_P, c_int]),
    "smm_hip_precond_create": (c_int, [_P, c_int, POINTER(_P)]),
    "smm_hip_precond_create_block": (c_int, [_P, c_int, c_int, POINTER(_P)]),
    "smm_hip_precond_create_block_capped": (c_int, [_P, c_int, c_int, c_int, POINTER(_P)]),
    "smm_hip_precond_block_level_cap": (c_int, [_P, POINTER(c_int)]),
    "smm_hip_precond_create_block_ex": (c_int, [_P, c_int, c_int, c_int, c_int, POINTER(_P)]),
    "smm_hip_precond_block_rows": (c_int, [_P, _P, c_size_t, _P]),
    "smm_hip_precond_block_record_bytes": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "smm_hip_precond_block_count": (c_int, [_P, POINTER(c_int)]),
    "smm_hip_precond_block_bounds": (c_int, [_P, _P, c_size_t]),
    "smm_hip_precond_destroy": (c_int, [_P]),
    "smm_hip_precond_set_sweep": (c_int, [_P, c_int]),
    "smm_hip_precond_take_error": (c_int, [_P, _P]),
    "smm_hip_precond_info": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "smm_hip_gen_poisson2d_nnz": (c_longlong, [c_int, c_int]),
    "smm_hip_gen_stencil3d_nnz": (c_longlong, [c_int, c_int, c_int]),
    "smm_hip_gen_banded_nnz": (c_longlong, [c_int, c_int, c_ulonglong, c_int]),
    "smm_hip_gen_banded_row_start": (c_longlong, [c_int, c_int, c_ulonglong, c_int, c_int]),
    "smm_hip_partials_count": (c_int, []),
    "smm_hip_finish_len": (c_int, []),
    "smm_hip_finish_totals_offset": (c_int, []),
    "smm_hip_cg_resident": (c_int, [c_int]),
    "smm_hip_bicgstab_resident": (c_int, [c_int]),
    "smm_hip_bicgstab_ws_destroy": (c_int, [_P]),
    "smm_hip_cg_ws_status": (c_int, [_P, _P, POINTER(c_int)]),
    "smm_hip_bicgstab_ws_bind": (c_int, [_P, _P, _P, _P]),
    "smm_hip_bicgstab_ws_pointers": (c_int, [_P, POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P), POINTER(_P)]),
    "smm_hip_comm_unique_id": (c_int, [_P]),
    "smm_hip_comm_create_rccl": (c_int, [c_int, c_int, _P, POINTER(_P)]),
    "smm_hip_comm_create_host": (c_int, [c_int, c_int, HOST_ALLREDUCE_FN, HOST_SENDRECV_FN, _P, POINTER(_P)]),
    "smm_hip_comm_create_self": (c_int, [POINTER(_P)]),
    "smm_hip_comm_destroy": (c_int, [_P]),
    "smm_hip_comm_info": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "smm_hip_comm_selftest": (c_int, [_P]),
    "smm_hip_comm_rccl_ranks": (c_int, [_P, POINTER(c_int)]),
    "smm_hip_partition_rows_by_nnz": (c_int, [_P, c_int, c_int, _P]),
    "smm_hip_dist_csr_destroy": (c_int, [_P]),
    "smm_hip_dist_csr_info": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_longlong), POINTER(c_longlong)]),
    "smm_hip_dist_csr_local_block": (c_int, [_P, POINTER(_P), POINTER(_P)]),
    "smm_hip_dist_csr_halo_chunks": (c_int, [_P, POINTER(c_int)]),
    "smm_hip_dist_csr_options": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(ctypes.c_double)]),
    "smm_hip_dist_csr_matvec_forms": (c_int, [_P, POINTER(ctypes.c_longlong), POINTER(ctypes.c_longlong)]),
    "smm_hip_dist_csr_thin_remote": (c_int, [_P, POINTER(c_int), POINTER(ctypes.c_longlong)]),
    "smm_hip_dist_csr_cg_fused": (c_int, [_P, POINTER(ctypes.c_longlong)]),
    "smm_hip_dist_csr_split_wait": (c_int, [_P, POINTER(ctypes.c_double), c_int]),
    "smm_hip_dist_p2p_plan": (c_int, [c_int, c_int, POINTER(c_longlong), POINTER(c_int), c_int, c_double, POINTER(c_longlong), c_int, POINTER(c_int)]),
}


def exported_symbols():
    """Every symbol include/smm_hip.h declares (used by the CPU-side export test)."""
    names = list(_PLAIN)
    for base in _TYPED:
        names += [f"{base}_f32", f"{base}_f64"]
    return names


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's).  Two HIP
    runtimes in one process cannot both own the GPU, so when torch is installed its copy is loaded first: libsmm_hip.so's
    NEEDED libamdhip64.so.7 then binds to it by SONAME, whichever of the two is imported first.  Without torch (or with
    SMM_HIP_SYSTEM_RUNTIME=1) the library uses the system ROCm runtime."""
    if os.environ.get("SMM_HIP_SYSTEM_RUNTIME"):
        return
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load(fma=False):
    """Load the shared library (once) and declare the prototypes.  Raises OSError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path(fma)
    if not os.path.exists(path):
        raise OSError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  sparse_matrix_math_amd has no CPU fallback."
        )
    _share_hip_runtime_with_torch()
    lib = ctypes.CDLL(path)
    older_build = bool(os.environ.get("SMM_HIP_LIBRARY"))  # A/B measurements against an earlier build: entry points it lacks are skipped

    def symbol(name):
        try:
            return getattr(lib, name)
        except AttributeError:
            if older_build:
                return None
            raise

    for name, (res, args) in _PLAIN.items():
        fn = symbol(name)
        if fn is None:
            continue
        fn.restype = res
        fn.argtypes = args
    for base, (res, args) in _TYPED.items():
        for suf, ct in (("f32", c_float), ("f64", c_double)):
            fn = symbol(f"{base}_{suf}")
            if fn is None:
                continue
            fn.restype = res
            apply_t = ctypes.CFUNCTYPE(c_int, c_void_p, POINTER(ct), POINTER(ct))
            fn.argtypes = [ct if a == "T" else POINTER(ct) if a == "PT" else apply_t if a == "APPLY" else a for a in args]
            if "APPLY" in args:
                fn.apply_type = apply_t
    _lib = lib
    return lib


def check(status):
    if status != SMM_HIP_OK:
        msg = load().smm_hip_last_error().decode("utf-8", "replace")
        raise SmmHipError(status, msg)
