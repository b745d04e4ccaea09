"""ctypes/numpy front-end of libkkoracle.so (test infrastructure, see kk_oracle.h)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile the C restatement (gcc).  Called by __graft_entry__.build()."""
    so = os.path.join(_HERE, "libkkoracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in ("kk_oracle.c", "kk_oracle_omp.c", "kk_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libkkoracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.kko_spgemm_symbolic.restype = C.c_int64
        _LIB.kko_spgemm_mults.restype = C.c_int64
        _LIB.kko_sort_and_merge.restype = C.c_int64
        _LIB.kko_laplace2d_nnz.restype = C.c_int64
        _LIB.kko_laplace3d_nnz.restype = C.c_int64
        _LIB.kko_hash_value_1_50.restype = C.c_double
        _LIB.kko_spgemm_kkmem_omp.restype = C.c_int64
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i64(v):
    return C.c_int64(int(v))


class Crs:
    """Host CSR triple: row_map int64[nrows+1], entries int32[nnz], values float64[nnz]."""

    def __init__(self, nrows, ncols, row_map, entries, values):
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.row_map = np.ascontiguousarray(row_map, dtype=np.int64)
        self.entries = np.ascontiguousarray(entries, dtype=np.int32)
        self.values = None if values is None else np.ascontiguousarray(values)
        assert self.row_map.shape == (self.nrows + 1,)

    @property
    def nnz(self):
        return int(self.row_map[-1]) if self.nrows > 0 else 0

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.values, self.entries, self.row_map), shape=(self.nrows, self.ncols))


# --------------------------------------------------------------------------- SpMV
def spmv_serial(mode, A, alpha, x, beta, y):
    """Kokkos::Serial SpMV restatement; y is updated in place and returned."""
    L = lib()
    if A.values.dtype == np.float64 and x.dtype == np.float64:
        fn, cf = L.kko_spmv_serial, C.c_double
    elif A.values.dtype == np.float32 and x.dtype == np.float64:
        fn, cf = L.kko_spmv_serial_f32a, C.c_double
    elif A.values.dtype == np.float32 and x.dtype == np.float32:
        fn, cf = L.kko_spmv_serial_f32, C.c_float
    else:
        raise TypeError("unsupported dtype combination")
    assert y.dtype == x.dtype and y.flags.c_contiguous and x.flags.c_contiguous
    rc = fn(C.c_char(mode.encode()), _i64(A.nrows), _i64(A.ncols), _p(A.row_map), _p(A.entries), _p(A.values),
            cf(alpha), _p(x), cf(beta), _p(y))
    if rc != 0:
        raise ValueError("invalid mode %r" % mode)
    return y


def spmv_sequential(mode, A, alpha, x, beta, y):
    rc = lib().kko_spmv_sequential(C.c_char(mode.encode()), _i64(A.nrows), _i64(A.ncols), _p(A.row_map),
                                   _p(A.entries), _p(A.values), C.c_double(alpha), _p(x), C.c_double(beta), _p(y))
    if rc != 0:
        raise ValueError("invalid mode %r" % mode)
    return y


def spmv_struct(mode, stencil_type, structure, A, alpha, x, beta, y):
    """KokkosSparse::Experimental::spmv_struct, host path; structure = (ni[, nj[, nk]])."""
    st = np.asarray(structure, dtype=np.int64)
    rc = lib().kko_spmv_struct(C.c_char(mode.encode()), C.c_int(stencil_type), C.c_int(len(st)), _p(st), _i64(A.nrows),
                               _i64(A.ncols), _p(A.row_map), _p(A.entries), _p(A.values), C.c_double(alpha), _p(x),
                               C.c_double(beta), _p(y))
    if rc != 0:
        raise ValueError("kko_spmv_struct rc=%d" % rc)
    return y


def spmv_mv_serial(mode, A, alpha, X, beta, Y):
    """X: (ncols|nrows) x nvec, Y likewise; any strides (numpy order C or F)."""
    assert X.dtype == np.float64 and Y.dtype == np.float64 and X.ndim == 2 and Y.ndim == 2
    e = X.itemsize
    rc = lib().kko_spmv_mv_serial(C.c_char(mode.encode()), _i64(A.nrows), _i64(A.ncols), _i64(X.shape[1]),
                                  _p(A.row_map), _p(A.entries), _p(A.values), C.c_double(alpha), _p(X),
                                  _i64(X.strides[0] // e), _i64(X.strides[1] // e), C.c_double(beta), _p(Y),
                                  _i64(Y.strides[0] // e), _i64(Y.strides[1] // e))
    if rc != 0:
        raise ValueError("invalid mode %r" % mode)
    return Y


def spmv_max_error(A, alpha, beta, max_x=1.0, max_y=1.0, max_val=1.0):
    """Reference test tolerance: 10*eps*(beta*max_y + alpha*max_nnz_row*max_val*max_x)
    (sparse/unit_test/Test_Sparse_spmv.hpp:84-91,181,432)."""
    mx = int(np.max(np.diff(A.row_map))) if A.nrows else 0
    return 10 * np.finfo(np.float64).eps * (abs(beta) * max_y + abs(alpha) * mx * max_val * max_x)


# --------------------------------------------------------------------------- SpGEMM
def spgemm_symbolic(A, B):
    rmC = np.zeros(A.nrows + 1, dtype=np.int64)
    nnz = lib().kko_spgemm_symbolic(C.c_int32(A.nrows), C.c_int32(A.ncols), C.c_int32(B.ncols), _p(A.row_map),
                                    _p(A.entries), _p(B.row_map), _p(B.entries), _p(rmC))
    assert nnz >= 0
    return rmC, int(nnz)


def spgemm(A, B, sort=True):
    """SPGEMM_DEBUG symbolic + numeric (+ the post-numeric sort_crs_matrix)."""
    rmC, nnz = spgemm_symbolic(A, B)
    entC = np.zeros(nnz, dtype=np.int32)
    valC = np.zeros(nnz, dtype=np.float64)
    rc = lib().kko_spgemm_numeric(C.c_int32(A.nrows), C.c_int32(A.ncols), C.c_int32(B.ncols), _p(A.row_map),
                                  _p(A.entries), _p(A.values), _p(B.row_map), _p(B.entries), _p(B.values), _p(rmC),
                                  _p(entC), _p(valC))
    assert rc == 0
    Cm = Crs(A.nrows, B.ncols, rmC, entC, valC)
    if sort:
        sort_crs(Cm)
    return Cm


def sort_crs(A):
    rc = lib().kko_sort_crs(_i64(A.nrows), _p(A.row_map), _p(A.entries),
                            _p(A.values) if A.values is not None else None)
    assert rc == 0
    return A


def sort_and_merge(A):
    """KokkosSparse::sort_and_merge_matrix: a new Crs with sorted rows and duplicate columns summed (A is sorted in place)."""
    rm_out = np.zeros(A.nrows + 1, dtype=np.int64)
    n = lib().kko_sort_and_merge(_i64(A.nrows), _p(A.row_map), _p(A.entries), _p(A.values), _p(rm_out), None, None)
    ent = np.zeros(n, dtype=np.int32); val = np.zeros(n)
    lib().kko_sort_and_merge(_i64(A.nrows), _p(A.row_map), _p(A.entries), _p(A.values), _p(rm_out), _p(ent), _p(val))
    return Crs(A.nrows, A.ncols, rm_out, ent, val)


def spgemm_mults(A, B):
    mx = C.c_int64(0)
    t = lib().kko_spgemm_mults(C.c_int32(A.nrows), _p(A.row_map), _p(A.entries), _p(B.row_map), C.byref(mx))
    return int(t), int(mx.value)


def transpose(A):
    rm = np.zeros(A.ncols + 1, dtype=np.int64)
    ent = np.zeros(A.nnz, dtype=np.int32)
    val = np.zeros(A.nnz, dtype=np.float64)
    rc = lib().kko_transpose(C.c_int32(A.nrows), C.c_int32(A.ncols), _p(A.row_map), _p(A.entries), _p(A.values),
                             _p(rm), _p(ent), _p(val))
    assert rc == 0
    return Crs(A.ncols, A.nrows, rm, ent, val)


def is_same_matrix(C1, C2, eps=1e-7):
    """Reference comparator: dims, nnz, row_map and entries identical, values
    |a-b|/(|a|+|b|) <= eps (sparse/unit_test/Test_Sparse_Utils.hpp:39-127,
    common/src/KokkosKernels_SimpleUtils.hpp:291-310)."""
    if (C1.nrows, C1.ncols, C1.nnz) != (C2.nrows, C2.ncols, C2.nnz):
        return False, "dims/nnz differ"
    if not np.array_equal(C1.row_map, C2.row_map):
        return False, "row_map differs"
    if not np.array_equal(C1.entries, C2.entries):
        return False, "entries differ"
    a, b = np.asarray(C1.values, dtype=np.float64), np.asarray(C2.values, dtype=np.float64)
    den = np.abs(a) + np.abs(b)
    rel = np.where(den > 0, np.abs(a - b) / np.where(den > 0, den, 1), 0.0)
    if rel.size and rel.max() > eps:
        return False, "values differ: max rel %g" % rel.max()
    return True, ""


# --------------------------------------------------------------------------- generators
def laplace1d(nx, bc=(1, 1)):
    nnz = 3 * (nx - 2) + 4
    rm = np.zeros(nx + 1, dtype=np.int64); ent = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
    rc = lib().kko_gen_laplace1d(_i64(nx), C.c_int(bc[0]), C.c_int(bc[1]), _p(rm), _p(ent), _p(val))
    if rc != 0:
        raise ValueError("kko_gen_laplace1d rc=%d" % rc)
    return Crs(nx, nx, rm, ent, val)


def laplace2d(stencil, nx, ny, bc=(1, 1, 1, 1)):
    s = {"FD": 0, "FE": 1}[stencil]
    nnz = lib().kko_laplace2d_nnz(s, _i64(nx), _i64(ny))
    rm = np.zeros(nx * ny + 1, dtype=np.int64); ent = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
    rc = lib().kko_gen_laplace2d(s, _i64(nx), _i64(ny), (C.c_int * 4)(*bc), _p(rm), _p(ent), _p(val))
    if rc != 0:
        raise ValueError("kko_gen_laplace2d rc=%d" % rc)
    return Crs(nx * ny, nx * ny, rm, ent, val)


def laplace3d(stencil, nx, ny, nz):
    s = {"FD": 0, "FE": 1}[stencil]
    nnz = lib().kko_laplace3d_nnz(s, _i64(nx), _i64(ny), _i64(nz))
    n = nx * ny * nz
    rm = np.zeros(n + 1, dtype=np.int64); ent = np.zeros(nnz, dtype=np.int32); val = np.zeros(nnz)
    rc = lib().kko_gen_laplace3d(s, _i64(nx), _i64(ny), _i64(nz), _p(rm), _p(ent), _p(val))
    if rc != 0:
        raise ValueError("kko_gen_laplace3d rc=%d" % rc)
    return Crs(n, n, rm, ent, val)


def rmat(scale, edgefactor=16, seed=17312837, value_seed=13718):
    """R-MAT CSR: duplicates merged, self loops kept, rows column-sorted, values in [1,50)."""
    n = 1 << scale
    ne = edgefactor * n
    keys = np.zeros(ne, dtype=np.uint64)
    rc = lib().kko_gen_rmat_keys(C.c_int(scale), _i64(ne), C.c_uint64(seed), _p(keys))
    assert rc == 0
    keys = np.unique(keys)
    rows = (keys >> np.uint64(32)).astype(np.int64)
    cols = (keys & np.uint64(0xFFFFFFFF)).astype(np.int32)
    rm = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=rm[1:])
    vals = hash_values_1_50(keys, value_seed)
    return Crs(n, n, rm, cols, vals)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_values_1_50(keys, seed):
    with np.errstate(over="ignore"):
        s = _splitmix64(np.array([seed], dtype=np.uint64))[0]
        h = _splitmix64(keys.astype(np.uint64) ^ s)
    return 1.0 + 49.0 * ((h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0))


def random_crs(nrows, ncols, nnz_per_row, variance=0, seed=0, bandwidth=None, sorted_rows=False, dtype=np.float64):
    """Seeded random CSR for parity tests (structure: uniform columns inside a band, possible duplicates,
    like kk_generate_sparse_matrix, sparse/src/KokkosSparse_IOUtils.hpp:30-82; values uniform in [0,1))."""
    rng = np.random.default_rng(seed)
    if variance:
        lens = np.clip(nnz_per_row + rng.integers(-variance, variance + 1, size=nrows), 0, None)
    else:
        lens = np.full(nrows, nnz_per_row)
    if ncols == 0:
        lens[:] = 0
    rm = np.zeros(nrows + 1, dtype=np.int64)
    np.cumsum(lens, out=rm[1:])
    nnz = int(rm[-1])
    rows = np.repeat(np.arange(nrows), lens)
    if bandwidth is None:
        cols = rng.integers(0, max(ncols, 1), size=nnz)
    else:
        cols = (rows * ncols // max(nrows, 1) + rng.integers(-bandwidth, bandwidth + 1, size=nnz)) % max(ncols, 1)
    cols = cols.astype(np.int32)
    vals = rng.random(nnz).astype(dtype)
    A = Crs(nrows, ncols, rm, cols, vals)
    if sorted_rows:
        v64 = A.values.astype(np.float64)
        A.values = v64
        sort_crs(A)
        A.values = A.values.astype(dtype)
    return A


# --------------------------------------------------------------------------- merge matrix (merge-path SpMV, K4)
def merge_matrix_diagonal(a, b, d):
    """Entries of the merge matrix M[i, j] = (a[i] > b[j]) along diagonal d, from the +a end towards the +b end
    (MergeMatrixDiagonal::operator() and diag_to_a_b, sparse/impl/KokkosSparse_merge_matrix.hpp:140-196)."""
    na, nb = len(a), len(b)
    if d <= na and d <= nb: size = d
    elif d > na and d > nb: size = na + nb - d
    else: size = min(na, nb)
    out = []
    for di in range(size):
        ai = (d - 1) - di if d < na else na - 1 - di
        bi = di if d < na else d + di - na
        out.append(1 if ai >= na else (0 if bi >= nb else int(a[ai] > b[bi])))
    return out


def diagonal_search(a, b, d):
    """(ai, bi): the first position on diagonal d whose merge-matrix entry is not 1 (diagonal_search + MergeMatrixDiagonal::position,
    sparse/impl/KokkosSparse_merge_matrix.hpp:118-135,196-227): a lower bound over the diagonal with predicate "equals 1"."""
    na = len(a)
    if d == 0:
        return 0, 0
    entries = merge_matrix_diagonal(a, b, d)
    lo, hi = 0, len(entries)                # lower_bound_thread with Equal<bool>: first idx with entries[idx] != 1
    while lo < hi:
        mid = (lo + hi) // 2
        if entries[mid] == 1: lo = mid + 1
        else: hi = mid
    ai = (d - 1) - lo if d < na else na - 1 - lo
    bi = lo if d < na else d + lo - na
    return ai + 1, bi


# --------------------------------------------------------------------------- CPU baselines
def first_touch(a):
    """copy of `a` whose pages were first touched by the OpenMP team (NUMA spread, like a Kokkos::View)"""
    out = np.empty_like(a)
    lib().kko_first_touch_copy(_p(out), _p(np.ascontiguousarray(a)), _i64(a.nbytes))
    return out


def omp_threads():
    return int(lib().kko_omp_max_threads())


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container with
    cpu.max = "1600000 100000" gets 16 CPUs' worth of time however many cores it can see -- 128 OpenMP threads under
    such a quota run 8x SLOWER than 16)."""
    import os
    n = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        a = len(os.sched_getaffinity(0))
        if a > 1:            # a mask of ONE cpu is what OMP_PROC_BIND leaves on the main thread once an OpenMP runtime
            n = min(n, a)    # (e.g. torch's) has started -- not a limit of the process
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def set_omp_threads(n):
    return int(lib().kko_omp_set_threads(C.c_int(int(n))))


def spmv_omp(A_row_map, entries, values, alpha, x, beta, y):
    L = lib()
    fn = L.kko_spmv_omp_i32 if A_row_map.dtype == np.int32 else L.kko_spmv_omp
    fn(_i64(len(A_row_map) - 1), _p(A_row_map), _p(entries), _p(values), C.c_double(alpha), _p(x),
       C.c_double(beta), _p(y))
    return y


def spgemm_kkmem_omp(A, B, sort=True, timings=None):
    """SPGEMM_KK on the host: OpenMP port of the KKMEM hash-accumulator kernels (impl_kkmem.hpp:196-272) -- the CPU baseline
    beside the GPU SpGEMM.  timings (dict) receives symbolic / numeric / sort seconds."""
    import time
    L = lib()
    rmC = np.zeros(A.nrows + 1, dtype=np.int64)
    args = (C.c_int32(A.nrows), C.c_int32(A.ncols), C.c_int32(B.ncols), _p(A.row_map), _p(A.entries), _p(A.values), _p(B.row_map), _p(B.entries), _p(B.values))
    t0 = time.perf_counter()
    nnz = L.kko_spgemm_kkmem_omp(C.c_int(0), *args, _p(rmC), None, None)
    t1 = time.perf_counter()
    assert nnz >= 0
    entC = np.zeros(nnz, dtype=np.int32); valC = np.zeros(nnz, dtype=np.float64)
    t2 = time.perf_counter()
    rc = L.kko_spgemm_kkmem_omp(C.c_int(1), *args, _p(rmC), _p(entC), _p(valC))
    t3 = time.perf_counter()
    assert rc == nnz
    Cm = Crs(A.nrows, B.ncols, rmC, entC, valC)
    if sort:                       # rows sorted in parallel, as the reference's sort_crs_matrix does
        rc = L.kko_sort_crs_omp(_i64(Cm.nrows), _p(Cm.row_map), _p(Cm.entries), _p(Cm.values))
        assert rc == 0
    if timings is not None:
        timings.update(symbolic_s=t1 - t0, numeric_s=t3 - t2, sort_s=time.perf_counter() - t3)
    return Cm


def spgemm_symbolic_kkmem_omp(A, B):
    """row_map of C = A*B by the OpenMP port of the KKMEM symbolic kernel alone (counts, no entries): what the full-size
    SpGEMM test compares against where the host cannot hold C itself"""
    rmC = np.zeros(A.nrows + 1, dtype=np.int64)
    nnz = lib().kko_spgemm_kkmem_omp(C.c_int(0), C.c_int32(A.nrows), C.c_int32(A.ncols), C.c_int32(B.ncols), _p(A.row_map), _p(A.entries), None,
                                     _p(B.row_map), _p(B.entries), None, _p(rmC), None, None)
    assert nnz >= 0 and nnz == rmC[-1]
    return rmC


def spmv_mv_omp(row_map_i32, entries, values, alpha, X, beta, Y):
    e = 8
    lib().kko_spmv_mv_omp_i32(_i64(len(row_map_i32) - 1), _i64(X.shape[1]), _p(row_map_i32), _p(entries), _p(values),
                              C.c_double(alpha), _p(X), _i64(X.strides[0] // e), _i64(X.strides[1] // e),
                              C.c_double(beta), _p(Y), _i64(Y.strides[0] // e), _i64(Y.strides[1] // e))
    return Y
