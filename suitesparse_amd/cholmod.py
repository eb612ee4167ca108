"""ctypes binding of libcholmod_amd.so -- the host-side mirror of the reference's
cholmod_l_* C API (include/cholmod.h) plus the engine shim (include/cholmod_hip.h).

This module is plumbing for tests and bench.py: it declares the C structs,
loads the in-tree shared library and fails loudly if it is missing.  All
numeric work happens in the library (HIP engine); nothing here computes.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHOLMOD_AMD_LIB") or os.path.join(_HERE, "lib", "libcholmod_amd.so")   # (override: A/B builds while tuning)
# the same library with the engine's test hooks compiled in (CHOLMOD_HIP_TEST_*): tests only, see csrc/Makefile
HOOKS_LIB_PATH = os.path.join(_HERE, "lib", "libcholmod_amd_testhooks.so")
CSRC = os.path.join(_HERE, "csrc")

CHOLMOD_MAXMETHODS = 9
CHOLMOD_HIP_NSTATS = 40

# constants (include/cholmod.h)
PATTERN, REAL, COMPLEX, ZOMPLEX = 0, 1, 2, 3
OK, NOT_INSTALLED, OUT_OF_MEMORY, TOO_LARGE, INVALID, GPU_PROBLEM = 0, -1, -2, -3, -4, -5
NOT_POSDEF = 1
NATURAL, GIVEN, POSTORDERED = 0, 1, 6
SIMPLICIAL, AUTO, SUPERNODAL = 0, 1, 2
SYS_A, SYS_LDLt, SYS_LD, SYS_DLt, SYS_L, SYS_Lt, SYS_D, SYS_P, SYS_Pt = range(9)
HIP_PLAN_HOST_ONLY = 2
# plan flags (include/cholmod_hip.h)
HIP_WIDE_OB, HIP_NO_FUSED_POTRF, HIP_NO_FUSED_TRSM, HIP_PHI_TWIN = 128, 512, 1024, 16384
HIP_INVALID = -4


class Method(C.Structure):
    _fields_ = [("lnz", C.c_double), ("fl", C.c_double), ("ordering", C.c_int)]


ERRFUNC = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_int, C.c_char_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p)


class Common(C.Structure):
    _fields_ = [
        ("supernodal", C.c_int), ("supernodal_switch", C.c_double),
        ("final_asis", C.c_int), ("final_super", C.c_int), ("final_ll", C.c_int),
        ("final_pack", C.c_int), ("final_monotonic", C.c_int), ("final_resymbol", C.c_int),
        ("zrelax", C.c_double * 3), ("nrelax", C.c_size_t * 3),
        ("prefer_upper", C.c_int), ("quick_return_if_not_posdef", C.c_int),
        ("print", C.c_int), ("nmethods", C.c_int), ("current", C.c_int), ("selected", C.c_int),
        ("method", Method * (CHOLMOD_MAXMETHODS + 1)),
        ("postorder", C.c_int), ("try_catch", C.c_int),
        ("error_handler", ERRFUNC),
        ("itype", C.c_int), ("dtype", C.c_int), ("status", C.c_int),
        ("fl", C.c_double), ("lnz", C.c_double), ("anz", C.c_double), ("modfl", C.c_double),
        ("malloc_count", C.c_size_t), ("memory_usage", C.c_size_t), ("memory_inuse", C.c_size_t),
        ("nrealloc_col", C.c_double), ("nrealloc_factor", C.c_double), ("ndbounds_hit", C.c_double),
        ("rowfacfl", C.c_double), ("aatfl", C.c_double),
        ("called_nd", C.c_int), ("blas_ok", C.c_int),
        ("useGPU", C.c_int), ("maxGpuMemBytes", C.c_size_t), ("maxGpuMemFraction", C.c_double),
        ("gpuMemorySize", C.c_size_t), ("gpuKernelTime", C.c_double), ("gpuFlops", C.c_int64),
        ("gpuNumKernelLaunches", C.c_int), ("devBuffSize", C.c_size_t), ("ibuffer", C.c_int),
        ("syrkStart", C.c_double),
        ("cholmod_cpu_gemm_time", C.c_double), ("cholmod_cpu_syrk_time", C.c_double),
        ("cholmod_cpu_trsm_time", C.c_double), ("cholmod_cpu_potrf_time", C.c_double),
        ("cholmod_gpu_gemm_time", C.c_double), ("cholmod_gpu_syrk_time", C.c_double),
        ("cholmod_gpu_trsm_time", C.c_double), ("cholmod_gpu_potrf_time", C.c_double),
        ("cholmod_assemble_time", C.c_double), ("cholmod_assemble_time2", C.c_double),
        ("cholmod_cpu_gemm_calls", C.c_size_t), ("cholmod_cpu_syrk_calls", C.c_size_t),
        ("cholmod_cpu_trsm_calls", C.c_size_t), ("cholmod_cpu_potrf_calls", C.c_size_t),
        ("cholmod_gpu_gemm_calls", C.c_size_t), ("cholmod_gpu_syrk_calls", C.c_size_t),
        ("cholmod_gpu_trsm_calls", C.c_size_t), ("cholmod_gpu_potrf_calls", C.c_size_t),
        ("hip_factor_on_device", C.c_int), ("hip_flags", C.c_int), ("hip_profile", C.c_int),
        ("hip_rank", C.c_int), ("hip_world", C.c_int),
        ("hip_allreduce", C.c_void_p), ("hip_allreduce_user", C.c_void_p),
        ("hip_cpu_fallback", C.c_int), ("prefer_zomplex", C.c_int), ("prefer_binary", C.c_int),
        ("hip_lazy_plan", C.c_int), ("hip_plan_seconds", C.c_double),
    ]


class Sparse(C.Structure):
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t),
                ("p", C.c_void_p), ("i", C.c_void_p), ("nz", C.c_void_p), ("x", C.c_void_p),
                ("z", C.c_void_p), ("stype", C.c_int), ("itype", C.c_int), ("xtype", C.c_int),
                ("dtype", C.c_int), ("sorted", C.c_int), ("packed", C.c_int)]


class Triplet(C.Structure):
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t), ("nnz", C.c_size_t),
                ("i", C.c_void_p), ("j", C.c_void_p), ("x", C.c_void_p), ("z", C.c_void_p),
                ("stype", C.c_int), ("itype", C.c_int), ("xtype", C.c_int), ("dtype", C.c_int)]


class Dense(C.Structure):
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t),
                ("d", C.c_size_t), ("x", C.c_void_p), ("z", C.c_void_p),
                ("xtype", C.c_int), ("dtype", C.c_int)]


class Factor(C.Structure):
    _fields_ = [("n", C.c_size_t), ("minor", C.c_size_t),
                ("Perm", C.c_void_p), ("ColCount", C.c_void_p), ("IPerm", C.c_void_p),
                ("nzmax", C.c_size_t),
                ("p", C.c_void_p), ("i", C.c_void_p), ("x", C.c_void_p), ("z", C.c_void_p),
                ("nz", C.c_void_p), ("next", C.c_void_p), ("prev", C.c_void_p),
                ("nsuper", C.c_size_t), ("ssize", C.c_size_t), ("xsize", C.c_size_t),
                ("maxcsize", C.c_size_t), ("maxesize", C.c_size_t),
                ("super", C.c_void_p), ("pi", C.c_void_p), ("px", C.c_void_p), ("s", C.c_void_p),
                ("ordering", C.c_int), ("is_ll", C.c_int), ("is_super", C.c_int),
                ("is_monotonic", C.c_int), ("itype", C.c_int), ("xtype", C.c_int),
                ("dtype", C.c_int), ("useGPU", C.c_int),
                ("hip_plan", C.c_void_p), ("hip_on_device", C.c_int), ("hip_host_valid", C.c_int),
                ("cx_twin", C.c_void_p), ("hip_apat_hash", C.c_uint64), ("hip_apat_nnz", C.c_size_t),
                ("hip_apat_valid", C.c_int), ("hip_apat_hash2", C.c_uint64), ("hip_is_twin", C.c_int),
                ("bset_work", C.c_void_p), ("hip_plan_ahead", C.c_int)]


# every symbol include/cholmod.h and include/cholmod_hip.h declare
API_SYMBOLS = [
    "cholmod_l_start", "cholmod_l_finish", "cholmod_l_defaults", "cholmod_l_error",
    "cholmod_l_malloc", "cholmod_l_calloc", "cholmod_l_free",
    "cholmod_l_allocate_sparse", "cholmod_l_free_sparse", "cholmod_l_copy_sparse",
    "cholmod_l_nnz", "cholmod_l_ptranspose", "cholmod_l_transpose",
    "cholmod_l_allocate_triplet", "cholmod_l_free_triplet", "cholmod_l_triplet_to_sparse",
    "cholmod_l_allocate_dense", "cholmod_l_zeros", "cholmod_l_ones", "cholmod_l_copy_dense",
    "cholmod_l_free_dense", "cholmod_l_free_factor",
    "cholmod_l_read_sparse", "cholmod_l_read_triplet", "cholmod_l_read_dense", "cholmod_l_read_matrix", "cholmod_l_check_factor", "cholmod_l_check_sparse",
    "cholmod_gpu_memorysize", "cholmod_gpu_probe", "cholmod_gpu_deallocate", "cholmod_gpu_end", "cholmod_gpu_allocate",
    "cholmod_l_gpu_stats", "cholmod_l_sdmult", "cholmod_l_norm_dense", "cholmod_l_norm_sparse",
    "cholmod_l_analyze", "cholmod_l_analyze_p", "cholmod_l_analyze_p2",
    "cholmod_l_factorize", "cholmod_l_factorize_p", "cholmod_l_solve", "cholmod_l_solve2",
    "cholmod_l_rcond", "cholmod_l_change_factor", "cholmod_l_realloc",
    "SuiteSparse_start", "SuiteSparse_finish", "SuiteSparse_malloc", "SuiteSparse_calloc", "SuiteSparse_realloc", "SuiteSparse_free",
    "cholmod_l_etree", "cholmod_l_postorder", "cholmod_l_rowcolcounts",
    "cholmod_l_super_symbolic", "cholmod_l_super_symbolic2", "cholmod_l_super_numeric",
    "cholmod_l_super_lsolve", "cholmod_l_super_ltsolve",
    "cholmod_l_gpu_memorysize", "cholmod_l_gpu_probe", "cholmod_l_gpu_deallocate",
    "cholmod_l_gpu_end", "cholmod_l_gpu_allocate",
    "cholmod_l_factor_to_host", "cholmod_l_hip_stats", "cholmod_l_refactorize_resident",
    "cholmod_l_gather_factor", "cholmod_l_hip_prepare",
]
HIP_SYMBOLS = [
    "cholmod_hip_probe", "cholmod_hip_memorysize", "cholmod_hip_set_device", "cholmod_hip_device_count",
    "cholmod_hip_plan_create", "cholmod_hip_plan_destroy", "cholmod_hip_factorize",
    "cholmod_hip_plan_create_dist", "cholmod_hip_set_allreduce", "cholmod_hip_get_partition",
    "cholmod_hip_get_groups", "cholmod_hip_get_batches", "cholmod_hip_progress_enable", "cholmod_hip_progress", "cholmod_hip_debug_schedule_hash", "cholmod_hip_diag_minmax", "cholmod_hip_values_staging", "cholmod_hip_values_push", "cholmod_hip_values_commit", "cholmod_hip_values_gather_index", "cholmod_hip_values_begin", "cholmod_hip_values_push_chunk", "cholmod_hip_debug_routing",
    "cholmod_hip_gather_factor",
    "cholmod_hip_upload_matrix", "cholmod_hip_factorize_resident",
    "cholmod_hip_set_value_map", "cholmod_hip_refresh_values",
    "cholmod_hip_download_factor", "cholmod_hip_download_even_columns", "cholmod_hip_upload_factor", "cholmod_hip_solve",
    "cholmod_hip_get_maps", "cholmod_hip_get_stats", "cholmod_hip_set_profiling",
    
    
    "cholmod_hip_dense_partial_factor", "cholmod_hip_factor_checks", "cholmod_hip_factor_checks_local", "cholmod_hip_get_launch_profile", "cholmod_hip_debug_thin_cycles", "cholmod_hip_debug_launch_regions",
    "cholmod_hip_rccl_unique_id", "cholmod_hip_rccl_attach", "cholmod_hip_rccl_detach",
    "cholmod_hip_version",
]

PROBES_PATH = os.path.join(_HERE, "lib", "libcholmod_amd_probes.so")
PROBE_SYMBOLS = [
    "cholmod_hip_bench_update_kernel", "cholmod_hip_bench_update_pair", "cholmod_hip_bench_mfma_peak", "cholmod_hip_bench_mfma_peak2",
    "cholmod_hip_bench_mixed", "cholmod_hip_debug_potrf_cycles", "cholmod_hip_debug_panel_cycles",
    "cholmod_hip_debug_latency", "cholmod_hip_bench_mfma_ceiling", "cholmod_hip_probe_cu_mask", "cholmod_hip_probe_overlap", "cholmod_hip_debug_update_diff", "cholmod_hip_debug_diag_cycles", "cholmod_hip_bench_handoff",
]

_probes = None


def build(force: bool = False) -> str:
    """Compile the host C layer and the HIP engine for gfx950 (in-tree)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j3"])
    return LIB_PATH


_libs = {}


def lib(hooks=None):
    """The product library; hooks=True: its twin with the engine's test hooks compiled in (tests that inject jitter,
    poison, failures).  hooks=None: the product library unless SSAMD_TEST_HOOKS_LIB=1 (worker processes of such tests)."""
    if hooks is None:
        hooks = os.environ.get("SSAMD_TEST_HOOKS_LIB") == "1"
    path = HOOKS_LIB_PATH if hooks else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no fallback path.")
    L = C.CDLL(path)
    vp, i64, dbl, sz = C.c_void_p, C.c_int64, C.c_double, C.c_size_t
    cm = C.POINTER(Common)
    sp, dn, fc = C.POINTER(Sparse), C.POINTER(Dense), C.POINTER(Factor)

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args

    sig("cholmod_l_start", C.c_int, [cm])
    sig("cholmod_l_finish", C.c_int, [cm])
    sig("cholmod_l_defaults", C.c_int, [cm])
    sig("cholmod_l_allocate_sparse", sp, [sz, sz, sz, C.c_int, C.c_int, C.c_int, C.c_int, cm])
    sig("cholmod_l_free_sparse", C.c_int, [C.POINTER(sp), cm])
    sig("cholmod_l_copy_sparse", sp, [sp, cm])
    sig("cholmod_l_nnz", i64, [sp, cm])
    sig("cholmod_l_ptranspose", sp, [sp, C.c_int, vp, vp, sz, cm])
    sig("cholmod_l_transpose", sp, [sp, C.c_int, cm])
    sig("cholmod_l_allocate_dense", dn, [sz, sz, sz, C.c_int, cm])
    sig("cholmod_l_zeros", dn, [sz, sz, C.c_int, cm])
    sig("cholmod_l_ones", dn, [sz, sz, C.c_int, cm])
    sig("cholmod_l_copy_dense", dn, [dn, cm])
    sig("cholmod_l_free_dense", C.c_int, [C.POINTER(dn), cm])
    sig("cholmod_l_free_factor", C.c_int, [C.POINTER(fc), cm])
    sig("cholmod_l_check_factor", C.c_int, [fc, cm])
    sig("cholmod_l_check_sparse", C.c_int, [sp, cm])
    sig("cholmod_l_gpu_stats", C.c_int, [cm])
    sig("cholmod_l_sdmult", C.c_int, [sp, C.c_int, C.POINTER(dbl * 2), C.POINTER(dbl * 2), dn, dn, cm])
    sig("cholmod_l_norm_dense", dbl, [dn, C.c_int, cm])
    sig("cholmod_l_norm_sparse", dbl, [sp, C.c_int, cm])
    sig("cholmod_l_analyze", fc, [sp, cm])
    sig("cholmod_l_analyze_p", fc, [sp, vp, vp, sz, cm])
    sig("cholmod_l_analyze_p2", fc, [C.c_int, sp, vp, vp, sz, cm])
    sig("cholmod_l_factorize", C.c_int, [sp, fc, cm])
    sig("cholmod_l_factorize_p", C.c_int, [sp, C.POINTER(dbl * 2), vp, sz, fc, cm])
    sig("cholmod_l_solve", dn, [C.c_int, fc, dn, cm])
    sig("cholmod_l_solve2", C.c_int, [C.c_int, fc, dn, sp, C.POINTER(dn), C.POINTER(sp), C.POINTER(dn), C.POINTER(dn), cm])
    sig("cholmod_l_rcond", dbl, [fc, cm])
    sig("cholmod_l_change_factor", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fc, cm])
    sig("cholmod_l_etree", C.c_int, [sp, vp, cm])
    sig("cholmod_l_postorder", i64, [vp, sz, vp, vp, cm])
    sig("cholmod_l_rowcolcounts", C.c_int, [sp, vp, sz, vp, vp, vp, vp, vp, vp, cm])
    sig("cholmod_l_super_symbolic", C.c_int, [sp, sp, vp, fc, cm])
    sig("cholmod_l_super_numeric", C.c_int, [sp, sp, C.POINTER(dbl * 2), fc, cm])
    sig("cholmod_l_super_lsolve", C.c_int, [fc, dn, dn, cm])
    sig("cholmod_l_super_ltsolve", C.c_int, [fc, dn, dn, cm])
    sig("cholmod_l_gpu_memorysize", C.c_int, [C.POINTER(sz), C.POINTER(sz), cm])
    sig("cholmod_l_gpu_probe", C.c_int, [cm])
    sig("cholmod_l_gpu_allocate", C.c_int, [cm])
    sig("cholmod_l_gpu_deallocate", C.c_int, [cm])
    sig("cholmod_l_factor_to_host", C.c_int, [fc, cm])
    sig("cholmod_l_hip_stats", C.c_int, [fc, C.POINTER(dbl * CHOLMOD_HIP_NSTATS), cm])
    sig("cholmod_l_refactorize_resident", C.c_int, [C.POINTER(dbl * 2), fc, cm])
    sig("cholmod_l_read_sparse", sp, [vp, cm])
    sig("cholmod_l_read_dense", dn, [vp, cm])
    sig("cholmod_l_read_triplet", vp, [vp, cm])
    sig("cholmod_l_free_triplet", C.c_int, [C.POINTER(vp), cm])
    sig("cholmod_l_read_matrix", vp, [vp, C.c_int, C.POINTER(C.c_int), cm])
    # engine shim
    sig("cholmod_hip_probe", C.c_int, [])
    sig("cholmod_hip_set_device", C.c_int, [C.c_int])
    sig("cholmod_hip_memorysize", C.c_int, [C.POINTER(sz), C.POINTER(sz)])
    sig("cholmod_hip_plan_create", vp, [i64, i64, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int)])
    sig("cholmod_hip_plan_destroy", None, [vp])
    sig("cholmod_hip_plan_create_dist", vp, [i64, i64, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_int)])
    sig("cholmod_hip_set_allreduce", C.c_int, [vp, ALLREDUCE_FN, vp])
    sig("cholmod_hip_get_partition", C.c_int, [vp, vp])
    sig("cholmod_hip_get_groups", C.c_int, [vp, vp, vp])
    sig("cholmod_hip_get_batches", i64, [vp, vp, vp])
    sig("cholmod_hip_progress_enable", C.c_int, [vp, C.c_int])
    sig("cholmod_hip_progress", C.c_int, [vp, vp])
    sig("cholmod_hip_debug_schedule_hash", C.c_int, [vp, vp])
    sig("cholmod_hip_debug_routing", C.c_int64, [vp, C.c_int64, vp, vp, vp, vp])
    sig("cholmod_hip_gather_factor", C.c_int, [vp])
    sig("cholmod_l_gather_factor", C.c_int, [fc, cm])
    sig("cholmod_l_hip_prepare", C.c_int, [fc, cm])
    sig("cholmod_hip_factorize", C.c_int, [vp, vp, vp, vp, vp, dbl, C.c_int, vp, C.POINTER(i64)])
    sig("cholmod_hip_upload_matrix", C.c_int, [vp, vp, vp, vp, vp])
    sig("cholmod_hip_factorize_resident", C.c_int, [vp, dbl, C.c_int, C.POINTER(i64)])
    sig("cholmod_hip_download_factor", C.c_int, [vp, vp])
    sig("cholmod_hip_upload_factor", C.c_int, [vp, vp])
    sig("cholmod_hip_solve", C.c_int, [vp, C.c_int, vp, i64, i64])
    sig("cholmod_hip_get_maps", C.c_int, [vp, vp, vp, vp])
    sig("cholmod_hip_get_stats", C.c_int, [vp, vp])
    sig("cholmod_hip_set_profiling", C.c_int, [vp, C.c_int])
    sig("cholmod_hip_dense_partial_factor", C.c_int, [vp, i64, i64, C.c_int, C.POINTER(i64)])
    sig("cholmod_hip_factor_checks", C.c_int, [vp, vp])
    sig("cholmod_hip_factor_checks_local", C.c_int, [vp, vp])
    sig("cholmod_hip_get_launch_profile", i64, [vp, i64, vp, vp, vp, vp, vp, vp])
    sig("cholmod_hip_debug_thin_cycles", C.c_int, [vp, i64, vp])
    sig("cholmod_hip_debug_launch_regions", i64, [vp, i64, i64, vp])
    sig("cholmod_hip_rccl_unique_id", C.c_int, [vp])
    sig("cholmod_hip_rccl_attach", C.c_int, [vp, vp])
    sig("cholmod_hip_rccl_detach", C.c_int, [vp])
    sig("cholmod_hip_version", C.c_char_p, [])
    _libs[path] = L
    return L


def probes():
    """Micro-benchmarks / tuning probes (include/cholmod_hip_probes.h): a library of
    their own, not part of the product."""
    global _probes
    if _probes is not None:
        return _probes
    if not os.path.exists(PROBES_PATH):
        raise RuntimeError(f"{PROBES_PATH} is missing: build it with __graft_entry__.build()")
    L = C.CDLL(PROBES_PATH)
    vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
    for name, res, args in (
            ("cholmod_hip_bench_update_kernel", dbl, [i64, i64, i64, C.c_int, C.c_int]),
            ("cholmod_hip_bench_update_pair", dbl, [i64, i64, i64, i64, i64, C.c_int, C.c_int]),
            ("cholmod_hip_bench_mfma_peak", dbl, [C.c_int, C.c_int]),
            ("cholmod_hip_bench_mfma_peak2", dbl, [C.c_int, C.c_int, C.c_int, C.c_int]),
            ("cholmod_hip_bench_mixed", dbl, [C.c_int, C.c_int, C.c_int]),
            ("cholmod_hip_bench_mfma_ceiling", dbl, [C.c_int, C.c_int, C.c_int, C.c_int, vp]),
            ("cholmod_hip_probe_cu_mask", dbl, [vp, C.c_int, C.c_int, C.c_int, vp]),
            ("cholmod_hip_probe_overlap", C.c_int, [vp, vp, C.c_int, i64, i64, C.c_int, C.c_int, C.c_int, vp]),
            ("cholmod_hip_debug_update_diff", dbl, [i64, i64, i64, C.c_int, C.c_int, C.c_int]),
            ("cholmod_hip_debug_diag_cycles", C.c_int, [vp, C.c_int, C.c_int]),
            ("cholmod_hip_debug_potrf_cycles", C.c_int, [vp]),
            ("cholmod_hip_debug_panel_cycles", C.c_int, [vp]),
            ("cholmod_hip_debug_latency", C.c_int, [vp, C.c_int])):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _probes = L
    return L


def _view(ptr, n, ctype, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(int(n),))


class Session:
    """One cholmod_common plus convenience wrappers (tests / bench harness)."""

    def __init__(self, supernodal=SUPERNODAL, use_gpu=1, print_level=0, postorder=True,
                 factor_on_device=False, hip_flags=0, rank=0, world=1, allreduce=None,
                 ordering="natural", hooks=None):
        """ordering (used by analyze() when no permutation is passed): "natural"
        (the harness default: tests pin results to the oracle's natural order),
        "default" (cholmod_l_start's strategy: the built-in nested dissection) or
        "nesdis".  hooks=True: the library with the engine's test hooks (lib ())."""
        self.L = lib(hooks)
        self.cm = Common()
        self.L.cholmod_l_start(C.byref(self.cm))
        if ordering == "natural":
            self.cm.nmethods = 1
            self.cm.method[0].ordering = 0          # CHOLMOD_NATURAL
        elif ordering == "nesdis":
            self.cm.nmethods = 1
            self.cm.method[0].ordering = 4          # CHOLMOD_NESDIS
        elif ordering != "default":
            raise ValueError(ordering)
        self.cm.supernodal = supernodal
        self.cm.useGPU = use_gpu
        self.cm.print = print_level
        self.cm.postorder = int(postorder)
        self.cm.hip_factor_on_device = int(factor_on_device)
        self.cm.hip_flags = hip_flags
        self._keep = []
        if world > 1 or allreduce is not None:
            # allreduce: a ctypes ALLREDUCE_FN (see suitesparse_amd/dist.py); with
            # world == 1 it is only used by the CHOLMOD_HIP_SHARE_AS_WORLD self test
            self.cm.hip_rank, self.cm.hip_world = rank, world
            self._keep.append(allreduce)
            self.cm.hip_allreduce = C.cast(allreduce, C.c_void_p)

    @property
    def status(self):
        return self.cm.status

    def finish(self):
        self.L.cholmod_l_finish(C.byref(self.cm))

    # ---- object helpers
    def sparse(self, n, Ap, Ai, Ax, stype, zomplex=False):
        """Copy numpy CSC arrays into a library-owned cholmod_sparse.  Complex values
        (numpy complex128) make a CHOLMOD_COMPLEX matrix, or CHOLMOD_ZOMPLEX on request."""
        nz = int(Ap[-1])
        cx = np.iscomplexobj(Ax)
        xtype = (ZOMPLEX if zomplex else COMPLEX) if cx else REAL
        A = self.L.cholmod_l_allocate_sparse(n, n, max(nz, 1), 1, 1, stype, xtype, C.byref(self.cm))
        if not A:
            raise MemoryError("cholmod_l_allocate_sparse")
        a = A.contents
        _view(a.p, n + 1, C.c_int64, np.int64)[:] = Ap
        if nz:
            _view(a.i, nz, C.c_int64, np.int64)[:] = Ai
            if not cx:
                _view(a.x, nz, C.c_double, np.float64)[:] = Ax
            elif zomplex:
                _view(a.x, nz, C.c_double, np.float64)[:] = np.real(Ax)
                _view(a.z, nz, C.c_double, np.float64)[:] = np.imag(Ax)
            else:
                _view(a.x, 2 * nz, C.c_double, np.float64)[:] = np.ascontiguousarray(Ax, dtype=np.complex128).view(np.float64)
        return A

    def dense(self, arr, zomplex=False):
        cx = np.iscomplexobj(arr)
        arr = np.asarray(arr, dtype=np.complex128 if cx else np.float64)
        n = arr.shape[-1] if arr.ndim > 1 else arr.shape[0]
        nrhs = arr.shape[0] if arr.ndim > 1 else 1
        xtype = (ZOMPLEX if zomplex else COMPLEX) if cx else REAL
        X = self.L.cholmod_l_allocate_dense(n, nrhs, n, xtype, C.byref(self.cm))
        if not cx:
            _view(X.contents.x, n * nrhs, C.c_double, np.float64)[:] = arr.reshape(-1)
        elif zomplex:
            _view(X.contents.x, n * nrhs, C.c_double, np.float64)[:] = arr.real.reshape(-1)
            _view(X.contents.z, n * nrhs, C.c_double, np.float64)[:] = arr.imag.reshape(-1)
        else:
            _view(X.contents.x, 2 * n * nrhs, C.c_double, np.float64)[:] = arr.reshape(-1).view(np.float64)
        return X

    def dense_to_numpy(self, X):
        x = X.contents
        if x.xtype == COMPLEX:
            out = _view(x.x, 2 * x.d * x.ncol, C.c_double, np.float64).copy().view(np.complex128)
        elif x.xtype == ZOMPLEX:
            out = (_view(x.x, x.d * x.ncol, C.c_double, np.float64)
                   + 1j * _view(x.z, x.d * x.ncol, C.c_double, np.float64))
        else:
            out = _view(x.x, x.d * x.ncol, C.c_double, np.float64).copy()
        return out.reshape(x.ncol, x.d)[:, :x.nrow] if x.ncol > 1 else out[:x.nrow]

    def free_sparse(self, A):
        self.L.cholmod_l_free_sparse(C.byref(A), C.byref(self.cm))

    def free_dense(self, X):
        self.L.cholmod_l_free_dense(C.byref(X), C.byref(self.cm))

    def free_factor(self, Lf):
        self.L.cholmod_l_free_factor(C.byref(Lf), C.byref(self.cm))

    # ---- the path
    def analyze(self, A, perm=None):
        if perm is None:
            Lf = self.L.cholmod_l_analyze(A, C.byref(self.cm))
        else:
            p = np.ascontiguousarray(perm, dtype=np.int64)
            Lf = self.L.cholmod_l_analyze_p(A, p.ctypes.data, None, 0, C.byref(self.cm))
        if not Lf:
            raise RuntimeError(f"cholmod_l_analyze failed, status {self.cm.status}")
        return Lf

    def factorize(self, A, Lf, beta=0.0):
        b = (C.c_double * 2)(beta, 0.0)
        return self.L.cholmod_l_factorize_p(A, C.byref(b), None, 0, Lf, C.byref(self.cm))

    def refactorize_resident(self, Lf, beta=0.0):
        b = (C.c_double * 2)(beta, 0.0)
        return self.L.cholmod_l_refactorize_resident(C.byref(b), Lf, C.byref(self.cm))

    def solve(self, Lf, b, sys=SYS_A, zomplex=False):
        B = self.dense(b, zomplex=zomplex)
        X = self.L.cholmod_l_solve(sys, Lf, B, C.byref(self.cm))
        self.free_dense(B)
        if not X:
            raise RuntimeError(f"cholmod_l_solve failed, status {self.cm.status}")
        out = self.dense_to_numpy(X)
        self.free_dense(X)
        return out

    def solve_subset(self, Lf, b, bset, sys=SYS_A, handles=None):
        """cholmod_l_solve2 with a sparse right-hand side: b (length n, only its entries at the indices `bset` are
        read).  Returns (x, xset): x of length n -- defined at the indices xset only, NaN elsewhere when the handles
        are fresh -- and xset in the order the library produced it.  `handles`: a dict kept by the caller between calls
        (the X / Xset / Y workspaces of the reference's interface); free with free_subset_handles."""
        h = handles if handles is not None else {}
        n = int(Lf.contents.n)
        B = self.dense(b)
        bset = np.ascontiguousarray(bset, dtype=np.int64)
        Bs = self.L.cholmod_l_allocate_sparse(n, 1, max(len(bset), 1), 0, 1, 0, PATTERN, C.byref(self.cm))
        _view(Bs.contents.p, 2, C.c_int64, np.int64)[:] = [0, len(bset)]
        if len(bset):
            _view(Bs.contents.i, len(bset), C.c_int64, np.int64)[:] = bset
        X = h.get("X") or C.POINTER(Dense)()
        Xs = h.get("Xset") or C.POINTER(Sparse)()
        Y = h.get("Y") or C.POINTER(Dense)()
        E = C.POINTER(Dense)()
        fresh = not bool(X)
        ok = self.L.cholmod_l_solve2(sys, Lf, B, Bs, C.byref(X), C.byref(Xs), C.byref(Y), C.byref(E), C.byref(self.cm))
        self.free_dense(B)
        self.free_sparse(Bs)
        h.update(X=X, Xset=Xs, Y=Y)
        try:
            if not ok:
                raise RuntimeError(f"cholmod_l_solve2 (Bset) failed, status {self.cm.status}")
            k = int(_view(Xs.contents.p, 2, C.c_int64, np.int64)[1])
            xset = _view(Xs.contents.i, max(k, 1), C.c_int64, np.int64)[:k].copy()
            xall = self.dense_to_numpy(X)
            x = np.full(n, np.nan, dtype=xall.dtype) if fresh else xall.copy()
            x[xset] = xall[xset]
            return x, xset
        finally:
            if handles is None:
                self.free_subset_handles(h)

    def free_subset_handles(self, h):
        for k, fr in (("X", self.free_dense), ("Y", self.free_dense), ("Xset", self.free_sparse)):
            if h.get(k):
                fr(h[k])
            h.pop(k, None)

    def hip_stats(self, Lf):
        s = (C.c_double * CHOLMOD_HIP_NSTATS)()
        self.L.cholmod_l_hip_stats(Lf, C.byref(s), C.byref(self.cm))
        return np.array(list(s))

    def factor_checks(self, Lf):
        """Invariants of the device-resident factor (cholmod_hip_factor_checks):
        dict(half_logdet, upper_nonzeros, nonfinite, fro2, nonpositive_diag)."""
        out = np.zeros(5)
        rc = self.L.cholmod_hip_factor_checks(Lf.contents.hip_plan, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"cholmod_hip_factor_checks failed: {rc}")
        return dict(half_logdet=out[0], upper_nonzeros=int(out[1]), nonfinite=int(out[2]),
                    fro2=out[3], nonpositive_diag=int(out[4]))

    def factor_checks_local(self, Lf):
        """This rank's share of the invariants of a distributed factor (cholmod_hip_factor_checks_local):
        the five numbers as an array; their sums over the ranks are those of the complete factor."""
        out = np.zeros(5)
        f = Lf.contents
        plan = C.cast(f.cx_twin, C.POINTER(Factor)).contents.hip_plan if f.cx_twin else f.hip_plan
        rc = self.L.cholmod_hip_factor_checks_local(plan, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"cholmod_hip_factor_checks_local failed: {rc}")
        return out

    def launch_profile(self, Lf):
        """Per-launch (kind, grid, aux, ms, flops, bytes) of the last profiled factorization."""
        plan = Lf.contents.hip_plan
        nl = self.L.cholmod_hip_get_launch_profile(plan, 0, None, None, None, None, None, None)
        kind = np.zeros(nl, dtype=np.int32); grid = np.zeros(nl, dtype=np.int32); aux = np.zeros(nl, dtype=np.int32)
        ms = np.zeros(nl); fl = np.zeros(nl); by = np.zeros(nl)
        self.L.cholmod_hip_get_launch_profile(plan, nl, kind.ctypes.data, grid.ctypes.data, aux.ctypes.data,
                                              ms.ctypes.data, fl.ctypes.data, by.ctypes.data)
        return dict(kind=kind, grid=grid, aux=aux, ms=ms, flops=fl, bytes=by)

    def set_profiling(self, Lf, on=True):
        if Lf.contents.hip_plan:
            self.L.cholmod_hip_set_profiling(Lf.contents.hip_plan, int(on))


class FactorView:
    """numpy views of a cholmod_factor's supernodal fields."""

    def __init__(self, Lf):
        f = Lf.contents
        self.n, self.nsuper = int(f.n), int(f.nsuper)
        self.ssize, self.xsize = int(f.ssize), int(f.xsize)
        self.maxcsize, self.maxesize, self.minor = int(f.maxcsize), int(f.maxesize), int(f.minor)
        self.ordering, self.is_super, self.is_ll, self.xtype = f.ordering, f.is_super, f.is_ll, f.xtype
        self.useGPU = f.useGPU
        self.Perm = _view(f.Perm, self.n, C.c_int64, np.int64)
        self.ColCount = _view(f.ColCount, self.n, C.c_int64, np.int64)
        self.super = _view(f.super, self.nsuper + 1, C.c_int64, np.int64)
        self.pi = _view(f.pi, self.nsuper + 1, C.c_int64, np.int64)
        self.px = _view(f.px, self.nsuper + 1, C.c_int64, np.int64)
        self.s = _view(f.s, self.ssize, C.c_int64, np.int64)
        if f.x and f.xtype == COMPLEX:      # interleaved (re, im) pairs
            self.x = _view(f.x, 2 * self.xsize, C.c_double, np.float64).view(np.complex128)
        else:
            self.x = _view(f.x, self.xsize, C.c_double, np.float64) if f.x else None
        self.hip_plan = f.hip_plan
        self.cx_twin = f.cx_twin
