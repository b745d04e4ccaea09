"""Thin, reference-shaped front end over the C ABI (names follow KokkosSparse: CrsMatrix with
graph.row_map / graph.entries / values, SPMVHandle, KokkosKernelsHandle + spgemm_symbolic/numeric).

Arrays live wherever the Backend puts them: `torch_backend()` = HBM tensors on the current CUDA/HIP
device (the product path).  tests/emu provides a numpy Backend bound to the emulator build for
logic tests on machines without a GPU; nothing in this file computes anything itself.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import F32, F64, I32, I64, CrsDesc, check


class Backend:
    """How arrays are allocated and how their device pointer / stream are obtained."""

    def __init__(self, lib, empty, ptr, stream, to_numpy, from_numpy, name):
        self.lib, self.empty, self.ptr, self.stream = lib, empty, ptr, stream
        self.to_numpy, self.from_numpy, self.name = to_numpy, from_numpy, name


_TORCH_BACKEND = None


def torch_backend():
    global _TORCH_BACKEND
    if _TORCH_BACKEND is None:
        import torch                     # (before the library is loaded: see lib() on the load order)
        from . import lib, device_check
        if torch.cuda.is_available():
            device_check()               # the library's own runtime must see the device too (it does not when it was loaded before torch)
        if not torch.cuda.is_available():
            raise RuntimeError("kokkos-kernels_amd needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU path")
        tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32,
               np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}

        def empty(n, dtype):
            return torch.empty(int(n), dtype=tdt[np.dtype(dtype)], device="cuda")

        def from_numpy(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")

        _TORCH_BACKEND = Backend(lib(), empty, lambda t: None if t is None else t.data_ptr(),
                                 lambda: torch.cuda.current_stream().cuda_stream,
                                 lambda t: t.detach().cpu().numpy(), from_numpy, "torch")
    return _TORCH_BACKEND


def _np_dtype(a):
    s = str(a.dtype).replace("torch.", "")
    return np.dtype(s)


def _scalar_type(a):
    return {np.dtype(np.float32): F32, np.dtype(np.float64): F64}[_np_dtype(a)]


def _offset_type(a):
    return {np.dtype(np.int32): I32, np.dtype(np.int64): I64}[_np_dtype(a)]


class _Graph:
    def __init__(self, row_map, entries):
        self.row_map, self.entries = row_map, entries


class CrsMatrix:
    """KokkosSparse::CrsMatrix as the hot path sees it (sparse/src/KokkosSparse_CrsMatrix.hpp:317-388):
    graph.row_map (offset[numRows+1]), graph.entries (int32[nnz]), values (scalar[nnz]), numCols."""

    def __init__(self, num_rows, num_cols, row_map, entries, values, backend=None):
        self.backend = backend or torch_backend()
        self._nrows, self._ncols = int(num_rows), int(num_cols)
        self.graph = _Graph(row_map, entries)
        self.values = values
        self._nnz = int(entries.shape[0]) if entries is not None else 0

    @classmethod
    def from_host(cls, num_rows, num_cols, row_map, entries, values, offset_dtype=np.int32, backend=None):
        be = backend or torch_backend()
        return cls(num_rows, num_cols, be.from_numpy(np.asarray(row_map).astype(offset_dtype)),
                   be.from_numpy(np.asarray(entries).astype(np.int32)),
                   None if values is None else be.from_numpy(np.asarray(values)), backend=be)

    def numRows(self): return self._nrows
    def numCols(self): return self._ncols
    def nnz(self): return self._nnz

    def desc(self):
        be = self.backend
        d = CrsDesc()
        d.num_rows, d.num_cols, d.nnz = self._nrows, self._ncols, self._nnz
        d.d_row_map = be.ptr(self.graph.row_map)
        d.d_entries = be.ptr(self.graph.entries) if self._nnz else None
        d.d_values = be.ptr(self.values) if (self._nnz and self.values is not None) else None
        d.offset_type = _offset_type(self.graph.row_map)
        d.value_type = _scalar_type(self.values) if self.values is not None else F64
        return d

    def to_host(self):
        be = self.backend
        return (be.to_numpy(self.graph.row_map), be.to_numpy(self.graph.entries),
                None if self.values is None else be.to_numpy(self.values))


_ALGOS = {"SPMV_DEFAULT": 0, "SPMV_FAST_SETUP": 1, "SPMV_NATIVE": 2, "SPMV_MERGE_PATH": 3, "SPMV_NATIVE_MERGE_PATH": 4}


class SPMVHandle:
    """KokkosSparse::SPMVHandle (sparse/src/KokkosSparse_spmv_handle.hpp:280-349): carries the algorithm
    choice; the per-matrix plan is created lazily by the first spmv call and freed with the handle."""

    def __init__(self, algo="SPMV_DEFAULT", backend=None):
        if algo not in _ALGOS:
            raise ValueError("SPMVHandle: algorithm %s cannot be used if A is a CrsMatrix" % algo)
        self.algo = algo
        self.backend = backend
        self._plan = None
        self._pending = {}

    def get_algorithm(self): return self.algo

    def set(self, key, value):
        """expert knobs (the reference exposes team_size / vector_length / ... as public members)"""
        if self._plan is None:
            self._pending[key] = value
        else:
            check(self.backend.lib, self.backend.lib.kkamd_spmv_plan_set(self._plan, key.encode(), int(value)))

    def values_changed(self):
        """the caller wrote to A.values: re-ordered copies the plan keeps (cached transpose, column-slab copy) copy them again at the next
        call (kkamd_spmv_plan_values_changed; required under the knob values_tracking = 1, optional otherwise)"""
        if self._plan is not None:
            check(self.backend.lib, self.backend.lib.kkamd_spmv_plan_values_changed(self._plan))

    def query(self, key):
        """what the analysis produced (kkamd_spmv_plan_query); None before the first spmv call"""
        if self._plan is None: return None
        v = C.c_int64(0)
        check(self.backend.lib, self.backend.lib.kkamd_spmv_plan_query(self._plan, key.encode(), C.byref(v)))
        return int(v.value)

    def export(self, what, count):
        """a per-tile array of the analysis ("tile_first_row", "tile_mode") as a host int32 array"""
        out = np.zeros(int(count), dtype=np.int32)
        check(self.backend.lib, self.backend.lib.kkamd_spmv_plan_export(self._plan, what.encode(), out.ctypes.data_as(C.c_void_p), int(count)))
        return out

    def _ensure(self, A, rank2=False):
        if self._plan is None:
            if rank2 and "defer_rank1" not in self._pending:
                # the reference's handle is set up by its first spmv call, for that call's rank (spmv_handle.hpp:280-349): a handle that
                # begins with rank 2 leaves the rank-1 analysis to the first rank-1 call, if one ever comes
                self._pending["defer_rank1"] = 1
            self.backend = A.backend
            lib = A.backend.lib
            p = C.c_void_p()
            d = A.desc()
            n = len(self._pending)                 # the handle's knobs go in with the creation: they shape the analysis
            keys = (C.c_char_p * max(n, 1))(*[k.encode() for k in self._pending])
            vals = (C.c_int * max(n, 1))(*[int(v) for v in self._pending.values()])
            check(lib, lib.kkamd_spmv_plan_create_knobs(C.byref(p), C.byref(d), _ALGOS[self.algo], keys, vals, n, A.backend.stream()))
            self._plan = p
        return self._plan

    def __del__(self):
        try:
            if self._plan is not None and self.backend is not None:
                self.backend.lib.kkamd_spmv_plan_destroy(self._plan)
        except Exception:
            pass
        self._plan = None


def _strides(a):
    if hasattr(a, "stride") and callable(a.stride):
        return tuple(a.stride())
    return tuple(s // a.itemsize for s in a.strides)


def spmv(*args):
    """KokkosSparse::spmv.  spmv(mode, alpha, A, x, beta, y) or spmv(handle, mode, alpha, A, x, beta, y)
    (the execution-space overloads map to the backend's current stream).  x, y rank 1 or rank 2.
    Dimension checks and messages follow sparse/src/KokkosSparse_spmv.hpp:126-142."""
    if isinstance(args[0], SPMVHandle):
        handle, mode, alpha, A, x, beta, y = args
    else:
        handle = None
        mode, alpha, A, x, beta, y = args
    be, lib = A.backend, A.backend.lib
    if len(x.shape) != len(y.shape) or len(x.shape) not in (1, 2):
        raise RuntimeError("KokkosSparse::spmv: Vector ranks do not match.")
    m, n = A.numRows(), A.numCols()
    xr, yr = x.shape[0], y.shape[0]
    xc = x.shape[1] if len(x.shape) == 2 else 1
    yc = y.shape[1] if len(y.shape) == 2 else 1
    trans = mode[0] in "TtHh"
    if mode[0] not in "NnCcTtHh":
        raise RuntimeError("Invalid transpose mode %s for KokkosSparse::spmv()" % mode)
    if xc != yc or (not trans and (n != xr or m != yr)) or (trans and (m != xr or n != yr)):
        raise RuntimeError("KokkosSparse::spmv: Dimensions do not match%s: , A: %d x %d, x: %d x %d, y: %d x %d"
                           % (" (transpose)" if trans else "", m, n, xr, xc, yr, yc))
    plan = handle._ensure(A, rank2=(len(x.shape) == 2 and xc > 1)) if handle is not None else None
    d = A.desc()
    vt = _scalar_type(y)
    if len(x.shape) == 1:
        check(lib, lib.kkamd_spmv(plan, C.byref(d), mode[0].encode(), float(alpha), be.ptr(x), float(beta), be.ptr(y),
                                  vt, be.stream()))
    else:
        xs, ys = _strides(x), _strides(y)
        check(lib, lib.kkamd_spmv_mv(plan, C.byref(d), mode[0].encode(), float(alpha), be.ptr(x), xs[0], xs[1],
                                     float(beta), be.ptr(y), ys[0], ys[1], xc, vt, be.stream()))
    return y


def spmv_struct(mode, stencil_type, structure, alpha, A, x, beta, y):
    """KokkosSparse::Experimental::spmv_struct (sparse/src/KokkosSparse_spmv.hpp:478-848).  structure: the grid extents
    (ni[, nj[, nk]]) as a host sequence; stencil_type 1 = FD (3/5/7-pt), 2 = FE (3/9/27-pt).  Rank-2 x/y with one
    column take the structured path, more columns fall through to spmv (:803-831)."""
    be, lib = A.backend, A.backend.lib
    if len(x.shape) != len(y.shape) or len(x.shape) not in (1, 2):
        raise RuntimeError("KokkosSparse::spmv_struct: Vector ranks do not match.")
    m, n = A.numRows(), A.numCols()
    xr, yr = x.shape[0], y.shape[0]
    xc = x.shape[1] if len(x.shape) == 2 else 1
    yc = y.shape[1] if len(y.shape) == 2 else 1
    if mode[0] not in "NnCcTtHh":
        raise RuntimeError("Invalid transpose mode %s for KokkosSparse::spmv_struct()" % mode)
    trans = mode[0] in "TtHh"
    # the reference only requires the vectors to be long enough here (:504-523)
    if xc != yc or (not trans and (n > xr or m > yr)) or (trans and (n > yr or m > xr)):
        raise RuntimeError("KokkosSparse::spmv_struct: Dimensions do not match%s: , A: %d x %d, x: %d x %d, y: %d x %d"
                           % (" (transpose)" if trans else "", m, n, xr, xc, yr, yc))
    if len(x.shape) == 2:
        if xc != 1:
            return spmv(mode, alpha, A, x, beta, y)
        xs, ys = _strides(x), _strides(y)
        if xs[0] != 1 or ys[0] != 1:
            return spmv(mode, alpha, A, x, beta, y)
    st = (C.c_int64 * len(structure))(*[int(v) for v in structure])
    d = A.desc()
    check(lib, lib.kkamd_spmv_struct(C.byref(d), mode[0].encode(), int(stencil_type), len(structure), st, float(alpha),
                                     be.ptr(x), float(beta), be.ptr(y), _scalar_type(y), be.stream()))
    return y


class _SpgemmHandle:
    def __init__(self, backend):
        self.backend = backend
        self.h = C.c_void_p()
        check(backend.lib, backend.lib.kkamd_spgemm_create(C.byref(self.h)))

    def get(self, what):
        v = C.c_int64()
        check(self.backend.lib, self.backend.lib.kkamd_spgemm_get(self.h, what, C.byref(v)))
        return int(v.value)

    def set(self, key, value):
        """SPGEMMHandle / KokkosKernelsHandle option setters (kkamd_spgemm_set): "algorithm", "accumulator", "compression",
        "compression_cut_off", "verbose" act; the reference's team / shared-memory / hash-scale hints are accepted and recorded
        (get_hint), without effect; unknown keys raise KkamdError(INVALID_ARG)"""
        check(self.backend.lib, self.backend.lib.kkamd_spgemm_set(self.h, key.encode(), float(value)))

    def get_hint(self, key):
        v = C.c_double()
        check(self.backend.lib, self.backend.lib.kkamd_spgemm_get_hint(self.h, key.encode(), C.byref(v)))
        return float(v.value)

    def get_c_nnz(self): return self.get(0)
    def is_symbolic_called(self): return bool(self.get(4))
    def is_numeric_called(self): return bool(self.get(5))

    def destroy(self):
        if self.h:
            self.backend.lib.kkamd_spgemm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


_SPGEMM_ALGOS = {"SPGEMM_KK": 0, "SPGEMM_KK_DENSE": 1, "SPGEMM_KK_MEMORY": 2, "SPGEMM_KK_LP": 3, "SPGEMM_DEFAULT": 4, "SPGEMM_DEBUG": 5,
                 "SPGEMM_SERIAL": 6, "SPGEMM_KK_SPEED": 7, "SPGEMM_KK_MEMSPEED": 8}


class KokkosKernelsHandle:
    """The slice of KokkosKernels::Experimental::KokkosKernelsHandle the SpGEMM path uses
    (sparse/src/KokkosKernels_Handle.hpp:385-482): create_spgemm_handle / get_spgemm_handle /
    destroy_spgemm_handle."""

    def __init__(self, backend=None):
        self.backend = backend
        self._spgemm = None

    def create_spgemm_handle(self, algo="SPGEMM_KK"):
        """algo: a KokkosSparse::SPGEMMAlgorithm name (sparse/src/KokkosSparse_spgemm_handle.hpp:44-93).  SPGEMM_DEBUG / SPGEMM_SERIAL
        are host-sequential in the reference and raise here (no CPU path); SPGEMM_KK_DENSE selects the dense-accumulator numeric."""
        self.backend = self.backend or torch_backend()
        if algo not in _SPGEMM_ALGOS:
            raise RuntimeError("Invalid SPGEMMAlgorithm name")
        self._spgemm = _SpgemmHandle(self.backend)
        self._spgemm.algo = algo
        self._spgemm.set("algorithm", _SPGEMM_ALGOS[algo])

    def get_spgemm_handle(self): return self._spgemm

    def destroy_spgemm_handle(self):
        if self._spgemm is not None:
            self._spgemm.destroy()
        self._spgemm = None


def spgemm_symbolic(kh, A, transposeA, B, transposeB, Cmat=None, allocate=True):
    """Matrix-level KokkosSparse::spgemm_symbolic (sparse/src/KokkosSparse_spgemm.hpp:40-61): allocates
    row_map C, runs the symbolic phase, allocates entries/values of get_c_nnz() and returns C.
    allocate=False (view-level behaviour, :spgemm_symbolic with row maps only) returns the row_map of C alone."""
    if transposeA or transposeB:
        raise RuntimeError("KokkosSparse::spgemm_symbolic: transposing A or B is not yet supported")
    sh = kh.get_spgemm_handle() if kh is not None else None
    if sh is None:
        raise ValueError("KokkosSparse::spgemm_symbolic: the given KernelHandle does not have an SpGEMM handle "
                         "associated with it.")
    if A.numCols() != B.numRows():
        raise RuntimeError("KokkosSparse::spgemm: A.numCols() != B.numRows()")
    be, lib = A.backend, A.backend.lib
    m, n, k = A.numRows(), A.numCols(), B.numCols()
    odt = _np_dtype(A.graph.row_map)
    rmC = be.empty(m + 1, odt)
    nnz = C.c_int64()
    check(lib, lib.kkamd_spgemm_symbolic(sh.h, m, n, k, be.ptr(A.graph.row_map), be.ptr(A.graph.entries),
                                         be.ptr(B.graph.row_map), be.ptr(B.graph.entries), be.ptr(rmC),
                                         _offset_type(A.graph.row_map), C.byref(nnz), be.stream()))
    if not allocate:
        return rmC
    vdt = _np_dtype(A.values) if A.values is not None else np.dtype(np.float64)
    return CrsMatrix(m, k, rmC, be.empty(nnz.value, np.int32), be.empty(nnz.value, vdt), backend=be)


def spgemm_numeric(kh, A, transposeA, B, transposeB, Cmat):
    if transposeA or transposeB:
        raise RuntimeError("KokkosSparse::spgemm_numeric: transposing A or B is not yet supported")
    sh = kh.get_spgemm_handle() if kh is not None else None
    if sh is None:
        raise ValueError("KokkosSparse::spgemm_numeric: the given KernelHandle does not have an SpGEMM handle "
                         "associated with it.")
    be, lib = A.backend, A.backend.lib
    # the library keeps entries(C) across numeric calls into the SAME arrays; whether these are the same it can only judge by address,
    # and a caching allocator hands a freed address out again: arrays this wrapper has not yet seen through THIS handle are told to be
    # new ("entries_computed" 0), whatever their address
    seen = (id(sh), id(Cmat.graph.entries), be.ptr(Cmat.graph.entries))
    if getattr(Cmat, "_kk_numeric_seen", None) != seen:
        sh.set("entries_computed", 0)
    try:
        check(lib, lib.kkamd_spgemm_numeric(sh.h, A.numRows(), A.numCols(), B.numCols(), be.ptr(A.graph.row_map),
                                            be.ptr(A.graph.entries), be.ptr(A.values), be.ptr(B.graph.row_map),
                                            be.ptr(B.graph.entries), be.ptr(B.values), be.ptr(Cmat.graph.row_map),
                                            be.ptr(Cmat.graph.entries), be.ptr(Cmat.values),
                                            _offset_type(A.graph.row_map), _scalar_type(A.values), be.stream()))
    except _capi.KkamdError as e:
        if e.status == _capi.ERR_STATE:
            raise ValueError(str(e))   # std::invalid_argument in the reference
        raise
    Cmat._kk_numeric_seen = seen
    return Cmat


def spgemm(A, transposeA, B, transposeB):
    """No-reuse KokkosSparse::spgemm<CMatrix>(A, false, B, false) (sparse/src/KokkosSparse_spgemm.hpp:170-218)."""
    kh = KokkosKernelsHandle(A.backend)
    kh.create_spgemm_handle()
    try:
        Cm = spgemm_symbolic(kh, A, transposeA, B, transposeB)
        spgemm_numeric(kh, A, transposeA, B, transposeB, Cm)
    finally:
        kh.destroy_spgemm_handle()
    return Cm


def sort_crs_matrix(A):
    be, lib = A.backend, A.backend.lib
    check(lib, lib.kkamd_sort_crs(A.numRows(), be.ptr(A.graph.row_map), be.ptr(A.graph.entries),
                                  be.ptr(A.values) if A.values is not None else None,
                                  _offset_type(A.graph.row_map), _scalar_type(A.values) if A.values is not None else F64,
                                  be.stream()))
    return A


def sort_and_merge_matrix(A):
    """KokkosSparse::sort_and_merge_matrix (sparse/src/KokkosSparse_SortCrs.hpp:304-400): returns a new matrix whose rows
    are sorted with duplicate columns summed; A itself is left sorted (as in the reference, which sorts its input views)."""
    be, lib = A.backend, A.backend.lib
    m = A.numRows()
    rm_out = be.empty(m + 1, _np_dtype(A.graph.row_map))
    ot = _offset_type(A.graph.row_map)
    vt = _scalar_type(A.values) if A.values is not None else F64
    n = C.c_int64()
    vals = be.ptr(A.values) if A.values is not None else None
    check(lib, lib.kkamd_sort_and_merge(m, be.ptr(A.graph.row_map), be.ptr(A.graph.entries), vals, ot, vt, be.ptr(rm_out), None, None,
                                        C.byref(n), be.stream()))
    ent = be.empty(max(n.value, 1), np.int32)
    val = be.empty(max(n.value, 1), _np_dtype(A.values)) if A.values is not None else None
    check(lib, lib.kkamd_sort_and_merge(m, be.ptr(A.graph.row_map), be.ptr(A.graph.entries), vals, ot, vt, be.ptr(rm_out), be.ptr(ent),
                                        be.ptr(val) if val is not None else None, C.byref(n), be.stream()))
    return CrsMatrix(m, A.numCols(), rm_out, ent[:n.value], val[:n.value] if val is not None else None, backend=be)


def transpose_matrix(A):
    """KokkosSparse::Impl::transpose_matrix (sparse/src/KokkosSparse_Utils.hpp:381-398); rows of the result are sorted."""
    be, lib = A.backend, A.backend.lib
    m, n, nnz = A.numRows(), A.numCols(), A.nnz()
    t_rm = be.empty(n + 1, _np_dtype(A.graph.row_map))
    t_ent = be.empty(max(nnz, 1), np.int32)
    t_val = be.empty(max(nnz, 1), _np_dtype(A.values)) if A.values is not None else None
    check(lib, lib.kkamd_transpose(m, n, nnz, be.ptr(A.graph.row_map), be.ptr(A.graph.entries),
                                   be.ptr(A.values) if A.values is not None else None, _offset_type(A.graph.row_map),
                                   _scalar_type(A.values) if A.values is not None else F64, be.ptr(t_rm), be.ptr(t_ent),
                                   be.ptr(t_val) if t_val is not None else None, be.stream()))
    return CrsMatrix(n, m, t_rm, t_ent[:nnz], t_val[:nnz] if t_val is not None else None, backend=be)


def laplace_matrix(stencil, nx, ny, nz=None, offset_dtype=np.int32, value_dtype=np.float64, backend=None, rows=None):
    """Structured Laplacian (every BC = 1) generated in place on the device; bit-identical to the
    reference's generate_structured_matrix2D/3D (test_common/KokkosKernels_Test_Structured_Matrix.hpp).
    rows=(begin, count) builds only that row slab (local row_map, global columns)."""
    be = backend or torch_backend()
    lib = be.lib
    dim = 2 if nz is None else 3
    s = {"FD": 0, "FE": 1}[stencil]
    n = nx * ny * (nz or 1)
    r0, cnt = rows if rows is not None else (0, n)
    rm = be.empty(cnt + 1, offset_dtype)
    nnz = C.c_int64()
    ot = I64 if np.dtype(offset_dtype) == np.dtype(np.int64) else I32
    vt = F64 if np.dtype(value_dtype) == np.dtype(np.float64) else F32
    check(lib, lib.kkamd_gen_laplace_rows(dim, s, nx, ny, nz or 1, r0, cnt, be.ptr(rm), None, None, ot, vt, C.byref(nnz),
                                          be.stream()))
    ent = be.empty(nnz.value, np.int32)
    val = be.empty(nnz.value, value_dtype)
    check(lib, lib.kkamd_gen_laplace_rows(dim, s, nx, ny, nz or 1, r0, cnt, be.ptr(rm), be.ptr(ent), be.ptr(val), ot, vt,
                                          C.byref(nnz), be.stream()))
    return CrsMatrix(cnt, n, rm, ent, val, backend=be)
