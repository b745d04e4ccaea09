"""CPU-side checks of the boundary: libkkamd.so loads (no GPU needed for dlopen) and exports every symbol that
include/kkamd.h declares; argument validation that happens before any device work returns the documented
status codes."""
import ctypes as C
import os
import re

import kk_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    kk = kk_loader.load()
    hdr = open(os.path.join(ROOT, "include", "kkamd.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*|kkamd_spgemm_handle_t\*)\s+(kkamd_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert len(declared) >= 18
    lib = C.CDLL(kk.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, "declared in kkamd.h but not exported: %s" % missing
    assert set(kk._capi.EXPORTS) <= declared
    assert lib.kkamd_version() >= 100


def test_argument_validation_without_device_work():
    kk = kk_loader.load()
    lib = kk.lib()
    d = kk._capi.CrsDesc()
    d.num_rows, d.num_cols, d.nnz = -1, 4, 0
    assert lib.kkamd_spmv(None, C.byref(d), b"N", 1.0, None, 0.0, None, 1, None) == kk._capi.ERR_INVALID_ARG
    assert b"negative" in lib.kkamd_last_error()
    d.num_rows = 4; d.offset_type = 7
    assert lib.kkamd_spmv(None, C.byref(d), b"N", 1.0, None, 0.0, None, 1, None) == kk._capi.ERR_INVALID_ARG
    d.offset_type = 0; d.value_type = 9
    assert lib.kkamd_spmv(None, C.byref(d), b"N", 1.0, None, 0.0, None, 1, None) == kk._capi.ERR_UNSUPPORTED
    d.value_type = 1
    d.d_row_map = 8  # never dereferenced: mode is rejected first
    assert lib.kkamd_spmv(None, C.byref(d), b"Q", 1.0, None, 0.0, None, 1, None) == kk._capi.ERR_INVALID_ARG
    assert b"Invalid transpose mode" in lib.kkamd_last_error()
    p = C.c_void_p()
    assert lib.kkamd_spmv_plan_create(C.byref(p), C.byref(d), 99, None) == kk._capi.ERR_INVALID_ARG
    assert lib.kkamd_spgemm_numeric(None, 1, 1, 1, None, None, None, None, None, None, None, None, None, 0, 1, None) == kk._capi.ERR_STATE
    v = C.c_int64()
    assert lib.kkamd_spgemm_get(None, 0, C.byref(v)) == kk._capi.ERR_INVALID_ARG
    assert lib.kkamd_exclusive_scan(None, -1, 0, None) == kk._capi.ERR_INVALID_ARG
    assert lib.kkamd_gen_laplace(4, 0, 4, 4, 4, None, None, None, 0, 1, None, None) == kk._capi.ERR_INVALID_ARG
