"""kokkos-kernels_amd -- MI355X (gfx950) native KokkosSparse::spmv / spgemm hot path.

The product is libkkamd.so (hand-written HIP, C ABI in include/kkamd.h) plus the C++ drop-in headers
under host/.  This Python package is plumbing for tests and benchmarks: it loads the library, keeps
arrays in HBM as torch tensors, hands raw device pointers across the C ABI, and runs the 1-D
row-partitioned multi-GPU SpMV over torch.distributed (RCCL).  It contains no CPU implementation and
fails loudly when the HIP library is missing.
"""
import ctypes as _C
import os as _os

from . import _capi
from ._capi import KkamdError  # noqa: F401

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.path.join(_HERE, "libkkamd.so")
# tools/ only: KKAMD_LIBRARY=libkkamd_ablate.so selects the measurement build (csrc: `make ablate`, -DKK_ABLATE), whose extra
# knobs switch parts of the kernels off.  Tests, bench.py and the drop-in headers always use libkkamd.so.
if _os.environ.get("KKAMD_LIBRARY"):
    LIB_PATH = _os.path.join(_HERE, _os.path.basename(_os.environ["KKAMD_LIBRARY"]))
_lib = None


def build(verbose=False, ablate=False):
    """Compile csrc/*.hip for gfx950 into libkkamd.so (hipcc cross-compiles without a GPU); ablate=True also builds the
    measurement library libkkamd_ablate.so."""
    import subprocess
    cmd = ["make", "-C", _os.path.join(_HERE, "csrc"), "-j8"] + (["all", "ablate"] if ablate else []) + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """The loaded HIP library.  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not _os.path.exists(LIB_PATH):
            raise RuntimeError("libkkamd.so is missing (%s): run __graft_entry__.build(); "
                               "kokkos-kernels_amd has no CPU fallback" % LIB_PATH)
        # Load order: torch's wheel carries its own HIP runtime, and a process that maps libkkamd.so (linked against /opt/rocm's) BEFORE
        # torch ends up with a runtime that finds no device ("no ROCm-capable device is detected" from the first kkamd call -- seen with
        # `python __graft_entry__.py smoke`, which builds, loads the library and only then imports torch).  torch is imported first only
        # when this process uses it anyway (already imported, or KKAMD_IMPORT_TORCH=1 / the torch backend asks for it): a host without
        # torch pays nothing, and a process in the bad order gets an error that names the cause (device_check below).
        import sys as _sys
        if "torch" not in _sys.modules and _os.environ.get("KKAMD_IMPORT_TORCH", "") == "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _lib = _capi.bind(_C.CDLL(LIB_PATH))
    return _lib


def device_check():
    """Raises with the load-order explanation when the library's HIP runtime sees no device although the machine has one."""
    import ctypes as C
    name = C.create_string_buffer(64); gfx = C.c_int(0); cus = C.c_int(0)
    rc = lib().kkamd_device_info(name, 64, C.byref(gfx), C.byref(cus))
    if rc != 0 or cus.value <= 0:
        raise RuntimeError("libkkamd.so sees no GPU (%s).  If this process also uses PyTorch-ROCm, torch must be imported BEFORE the library is "
                           "loaded (its wheel carries its own HIP runtime; see INTEGRATION.md, 'load order'): import torch first, or set "
                           "KKAMD_IMPORT_TORCH=1." % lib().kkamd_last_error().decode(errors="replace"))
    return name.value.decode(), bool(gfx.value), cus.value


from .sparse import (CrsMatrix, SPMVHandle, KokkosKernelsHandle, spmv, spmv_struct, sort_and_merge_matrix, transpose_matrix, spgemm_symbolic, spgemm_numeric, spgemm,  # noqa: E402,F401
                     sort_crs_matrix, laplace_matrix, Backend, torch_backend)
from . import io  # noqa: E402,F401  (matrix file formats: .mtx / .bin / .crs)
