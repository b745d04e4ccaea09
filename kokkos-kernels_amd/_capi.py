"""ctypes signatures of libkkamd.so (include/kkamd.h).  Pointer plumbing only: every compute call
goes to the HIP library; there is no Python or CPU implementation behind these functions."""
import ctypes as C

F32, F64 = 0, 1
I32, I64 = 0, 1
OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_ALLOC, ERR_STATE = range(6)
SPMV_DEFAULT, SPMV_FAST_SETUP, SPMV_NATIVE, SPMV_MERGE_PATH, SPMV_NATIVE_MERGE_PATH = range(5)

EXPORTS = [
    "kkamd_last_error", "kkamd_version", "kkamd_device_info", "kkamd_trace_push", "kkamd_trace_pop", "kkamd_spmv_plan_create", "kkamd_spmv_plan_create_knobs", "kkamd_release_scratch", "kkamd_spmv_plan_destroy",
    "kkamd_spmv", "kkamd_spmv_mv", "kkamd_spmv_struct", "kkamd_sort_and_merge", "kkamd_transpose", "kkamd_spmv_plan_set", "kkamd_spmv_plan_values_changed", "kkamd_spmv_plan_query", "kkamd_spmv_plan_export", "kkamd_set_default", "kkamd_spgemm_create",
    "kkamd_spgemm_destroy", "kkamd_spgemm_set", "kkamd_spgemm_symbolic", "kkamd_spgemm_numeric", "kkamd_spgemm_get", "kkamd_spgemm_get_hint", "kkamd_sort_crs",
    "kkamd_exclusive_scan", "kkamd_gen_laplace", "kkamd_gen_laplace_rows", "kkamd_bench_read",
    "kkamd_dist_unique_id", "kkamd_dist_transport_selftest", "kkamd_dist_spmv_create", "kkamd_dist_spmv_destroy", "kkamd_dist_spmv_x_local", "kkamd_dist_spmv_apply",
    "kkamd_dist_spmv_query",
    "kkamd_dist_spgemm_partition", "kkamd_dist_spgemm_create", "kkamd_dist_spgemm_destroy", "kkamd_dist_spgemm_handle", "kkamd_dist_spgemm_symbolic",
    "kkamd_dist_spgemm_numeric", "kkamd_dist_spgemm_query",
]

ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int,
                          C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_void_p)


class Transport(C.Structure):
    """kkamd_transport_t: how the multi-GPU SpMV moves x entries when it is not the built-in RCCL"""
    _fields_ = [("ctx", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("exchange", EXCHANGE_FN)]


class CrsDesc(C.Structure):
    _fields_ = [("num_rows", C.c_int64), ("num_cols", C.c_int64), ("nnz", C.c_int64), ("d_row_map", C.c_void_p),
                ("d_entries", C.c_void_p), ("d_values", C.c_void_p), ("offset_type", C.c_int), ("value_type", C.c_int)]


class KkamdError(RuntimeError):
    """Non-zero kkamd_status.  .status holds the code; the C++ shim maps INVALID_ARG/HIP to
    std::runtime_error and STATE to std::invalid_argument like the reference."""

    def __init__(self, status, msg):
        super().__init__("kkamd status %d: %s" % (status, msg))
        self.status = status


def bind(lib):
    vp, i64, ci, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    lib.kkamd_last_error.restype = C.c_char_p
    lib.kkamd_version.restype = ci
    lib.kkamd_device_info.argtypes = [C.c_char_p, ci, C.POINTER(ci), C.POINTER(ci)]
    lib.kkamd_trace_push.argtypes = [C.c_char_p]
    lib.kkamd_trace_pop.argtypes = []
    lib.kkamd_spmv_plan_create.argtypes = [C.POINTER(vp), C.POINTER(CrsDesc), ci, vp]
    lib.kkamd_spmv_plan_create_knobs.argtypes = [C.POINTER(vp), C.POINTER(CrsDesc), ci, C.POINTER(C.c_char_p), C.POINTER(ci), ci, vp]
    lib.kkamd_release_scratch.argtypes = []
    lib.kkamd_spmv_plan_destroy.argtypes = [vp]
    lib.kkamd_spmv.argtypes = [vp, C.POINTER(CrsDesc), C.c_char, dbl, vp, dbl, vp, ci, vp]
    lib.kkamd_spmv_mv.argtypes = [vp, C.POINTER(CrsDesc), C.c_char, dbl, vp, i64, i64, dbl, vp, i64, i64, i64, ci, vp]
    lib.kkamd_spmv_struct.argtypes = [C.POINTER(CrsDesc), C.c_char, ci, ci, C.POINTER(i64), dbl, vp, dbl, vp, ci, vp]
    lib.kkamd_spmv_plan_set.argtypes = [vp, C.c_char_p, ci]
    lib.kkamd_spmv_plan_values_changed.argtypes = [vp]
    lib.kkamd_set_default.argtypes = [C.c_char_p, ci]
    lib.kkamd_spmv_plan_query.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    lib.kkamd_spmv_plan_export.argtypes = [vp, C.c_char_p, vp, i64]
    lib.kkamd_spgemm_create.argtypes = [C.POINTER(vp)]
    lib.kkamd_spgemm_destroy.argtypes = [vp]
    lib.kkamd_spgemm_set.argtypes = [vp, C.c_char_p, dbl]
    lib.kkamd_spgemm_symbolic.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, ci, C.POINTER(i64), vp]
    lib.kkamd_spgemm_numeric.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    lib.kkamd_spgemm_get.argtypes = [vp, ci, C.POINTER(i64)]
    lib.kkamd_spgemm_get_hint.argtypes = [vp, C.c_char_p, C.POINTER(dbl)]
    lib.kkamd_sort_crs.argtypes = [i64, vp, vp, vp, ci, ci, vp]
    lib.kkamd_exclusive_scan.argtypes = [vp, i64, ci, vp]
    lib.kkamd_sort_and_merge.argtypes = [i64, vp, vp, vp, ci, ci, vp, vp, vp, C.POINTER(i64), vp]
    lib.kkamd_transpose.argtypes = [i64, i64, i64, vp, vp, vp, ci, ci, vp, vp, vp, vp]
    lib.kkamd_gen_laplace.argtypes = [ci, ci, i64, i64, i64, vp, vp, vp, ci, ci, C.POINTER(i64), vp]
    lib.kkamd_gen_laplace_rows.argtypes = [ci, ci, i64, i64, i64, i64, i64, vp, vp, vp, ci, ci, C.POINTER(i64), vp]
    lib.kkamd_bench_read.argtypes = [vp, i64, ci, ci, ci, vp, vp]
    lib.kkamd_dist_unique_id.argtypes = [vp]
    lib.kkamd_dist_transport_selftest.argtypes = [i64, vp]
    lib.kkamd_dist_spmv_create.argtypes = [C.POINTER(vp), C.POINTER(CrsDesc), C.POINTER(i64), ci, ci, vp, C.POINTER(Transport), ci, ci, ci, ci, vp]
    lib.kkamd_dist_spmv_destroy.argtypes = [vp]
    lib.kkamd_dist_spmv_x_local.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.kkamd_dist_spmv_apply.argtypes = [vp, dbl, vp, dbl, vp, ci, vp]
    lib.kkamd_dist_spmv_query.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    lib.kkamd_dist_spgemm_partition.argtypes = [i64, vp, vp, vp, ci, ci, C.POINTER(i64), C.POINTER(i64), vp]
    lib.kkamd_dist_spgemm_create.argtypes = [C.POINTER(vp), ci, ci, C.POINTER(i64)]
    lib.kkamd_dist_spgemm_destroy.argtypes = [vp]
    lib.kkamd_dist_spgemm_handle.argtypes = [vp]
    lib.kkamd_dist_spgemm_symbolic.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, ci, C.POINTER(i64), vp]
    lib.kkamd_dist_spgemm_numeric.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    lib.kkamd_dist_spgemm_query.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name == "kkamd_dist_spgemm_handle":
            fn.restype = vp
        elif name not in ("kkamd_last_error",):
            fn.restype = ci
    return lib


def check(lib, status):
    if status != OK:
        raise KkamdError(status, lib.kkamd_last_error().decode(errors="replace"))
