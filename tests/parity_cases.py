"""Parity checks shared by the emulator tests (CPU, small sizes) and the GPU tests (`-m gpu`, through
the C ABI of libkkamd.so).  Every check compares the library under test with the CPU oracle on the same
seeded inputs, with the reference's own comparators and tolerances:
  SpMV   |expected - y| <= 10*eps*(beta*max_y + alpha*max_row*max_val*max_x), NaN mismatch fails
         (sparse/unit_test/Test_Sparse_spmv.hpp:84-91,181,432)
  SpGEMM row_map and entries IDENTICAL, values |a-b|/(|a|+|b|) <= 1e-7
         (sparse/unit_test/Test_Sparse_Utils.hpp:39-127)
"""
import numpy as np

import kk_loader
import oracle

kk = kk_loader.load()
EPS_F = float(np.finfo(np.float32).eps)


def dev(be, A0, offset_dtype=np.int32, value_dtype=None):
    vals = A0.values if value_dtype is None else A0.values.astype(value_dtype)
    return kk.CrsMatrix.from_host(A0.nrows, A0.ncols, A0.row_map, A0.entries, vals, offset_dtype=offset_dtype, backend=be)


def fspmv_ok(expected, got, tol):
    nan_mismatch = np.isnan(expected) ^ np.isnan(got)
    err = np.abs(expected - got)
    bad = nan_mismatch | (err > tol)
    return not bad.any(), (float(np.nanmax(err)) if err.size else 0.0)


def check_spmv(be, A0, mode="N", alpha=1.0, beta=0.0, algo=None, nans=False, seed=0, offset_dtype=np.int32,
               max_val=1.0, knobs=None, value_dtype=None, vec_dtype=np.float64, expect=None):
    """one check_spmv() of the reference test (Test_Sparse_spmv.hpp:168-216)"""
    rng = np.random.default_rng(seed)
    trans = mode in "TH"
    nin, nout = (A0.nrows, A0.ncols) if trans else (A0.ncols, A0.nrows)
    x = rng.random(nin).astype(vec_dtype)
    y0 = rng.random(nout).astype(vec_dtype)
    if nans:
        y0[::19] = np.nan
    A = dev(be, A0, offset_dtype, value_dtype)
    xd, yd = be.from_numpy(x), be.from_numpy(y0)
    if algo is None:
        kk.spmv(mode, alpha, A, xd, beta, yd)
    else:
        h = kk.SPMVHandle(algo)
        for k_, v_ in (knobs or {}).items():
            h.set(k_, v_)
        kk.spmv(h, mode, alpha, A, xd, beta, yd)
        kk.spmv(h, mode, alpha, A, xd, beta, yd) if beta == 0.0 else None   # handle reuse
        for k_, v_ in (expect or {}).items():                               # what the analysis must have produced
            assert h.query(k_) == v_, "plan query %s: %r, expected %r" % (k_, h.query(k_), v_)
    got = be.to_numpy(yd).astype(np.float64)
    Ao = A0 if value_dtype is None else oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, A0.values.astype(value_dtype))
    if vec_dtype == np.float64:
        exp = oracle.spmv_serial(mode, Ao, alpha, x, beta, y0.copy())
        eps_scale = 1.0
    else:
        exp = oracle.spmv_serial(mode, Ao, alpha, x, beta, y0.copy()).astype(np.float64)
        eps_scale = EPS_F / np.finfo(np.float64).eps
    tol = oracle.spmv_max_error(A0, alpha, beta, max_val=max_val) * eps_scale
    if trans and A0.nnz:      # a transposed product accumulates per COLUMN: scale the reference's bound by the longest column instead
        longest_col = int(np.bincount(A0.entries, minlength=A0.ncols).max()); longest_row = int(np.diff(A0.row_map).max())
        tol *= max(1.0, longest_col / max(longest_row, 1))
    ok, err = fspmv_ok(exp, got, max(tol, 1e-300))
    assert ok, "spmv mismatch mode=%s alpha=%g beta=%g algo=%s: max err %g > tol %g" % (mode, alpha, beta, algo, err, tol)
    return h if algo is not None else None


def check_spmv_mv(be, A0, nvec, mode="N", alpha=1.0, beta=0.0, x_order="F", y_order="F", algo=None, seed=0, knobs=None, expect=None,
                  max_val=1.0, nans=False, offset_dtype=np.int32, value_dtype=None, x_special=None):
    """x_special: {row: value} written into every column of X after the random fill (Inf / NaN propagation)"""
    rng = np.random.default_rng(seed)
    trans = mode in "TH"
    nin, nout = (A0.nrows, A0.ncols) if trans else (A0.ncols, A0.nrows)
    X = np.asarray(rng.random((nin, nvec)), order=x_order)
    for r_, v_ in (x_special or {}).items():
        X[r_, :] = v_
    Y0 = np.asarray(rng.random((nout, nvec)), order=y_order)
    if nans:
        Y0[::7, :] = np.nan
    if value_dtype is not None:
        A0 = oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, A0.values.astype(value_dtype).astype(np.float64))
    A = dev(be, A0, offset_dtype, value_dtype)
    Xd, Yd = _to_dev_2d(be, X), _to_dev_2d(be, Y0)
    h = None
    if algo is None:
        kk.spmv(mode, alpha, A, Xd, beta, Yd)
    else:
        h = kk.SPMVHandle(algo)
        for k_, v_ in (knobs or {}).items():
            h.set(k_, v_)
        kk.spmv(h, mode, alpha, A, Xd, beta, Yd)
        if beta == 0.0:
            kk.spmv(h, mode, alpha, A, Xd, beta, Yd)                            # handle reuse
        for k_, v_ in (expect or {}).items():
            assert h.query(k_) == v_, "plan query %s: %r, expected %r" % (k_, h.query(k_), v_)
    got = _to_host_2d(be, Yd)
    exp = oracle.spmv_mv_serial(mode, A0, alpha, X, beta, Y0.copy(order="K"))
    tol = oracle.spmv_max_error(A0, alpha, beta, max_val=max_val)
    if trans and A0.nnz:      # a transposed product accumulates per COLUMN: scale the reference's bound by the longest column instead (as check_spmv does)
        longest_col = int(np.bincount(A0.entries, minlength=A0.ncols).max()); longest_row = int(np.diff(A0.row_map).max())
        tol *= max(1.0, longest_col / max(longest_row, 1))
    assert not (np.isnan(exp) ^ np.isnan(got)).any(), "spmv_mv NaN mismatch nvec=%d" % nvec
    inf = np.isinf(exp)
    assert np.array_equal(inf, np.isinf(got)) and np.array_equal(exp[inf], got[inf]), "spmv_mv Inf mismatch nvec=%d" % nvec
    exp = np.where(inf, 0.0, exp); got = np.where(inf, 0.0, got)
    err = np.nanmax(np.abs(exp - got)) if got.size else 0.0
    assert err <= max(tol, 1e-300), "spmv_mv mismatch nvec=%d mode=%s orders=%s%s: %g > %g" % (nvec, mode, x_order, y_order, err, tol)
    return h


def check_mv5(be, light=False):
    combos = ((16, "C", "C", 1.5, 0.5, np.int32), (16, "F", "F", 1.0, 0.0, np.int64), (5, "C", "F", 2.0, 0.0, np.int32), (37, "F", "C", -1.0, 1.0, np.int32),
              (32, "C", "C", 1.0, 0.0, np.int32))
    for ci, (name, A0, tiles, other) in enumerate(mv5_cases()):
        for nvec, xo, yo, alpha, beta, off in (combos if (ci < 3 or not light) else combos[:2]):
            h = check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", max_val=1.5, nans=(beta == 0.0), offset_dtype=off)
            got = h.query("mv5_tiles")
            assert (got > 0) if tiles is None else (got == tiles), (name, nvec, got, tiles)
            if other is not None and got > 0:
                assert h.query("mv5_other_rows") == other, (name, h.query("mv5_other_rows"))
    name, A0, _, _ = mv5_cases()[0]
    h = check_spmv_mv(be, A0, 16, "N", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", max_val=1.5, value_dtype=np.float32)
    assert h.query("mv5_tiles") > 0 and h.query("mv5_fill_permille") == 1000
    # Inf / NaN in X: the rows that hold the column get it, the other rows of the tile do not (block 0: rows 0..31 hold column 3;
    # the band matrix: rows 90..110 hold column 100)
    for yo in ("C", "F"):
        h = check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", yo, algo="SPMV_DEFAULT", max_val=1.5, nans=True, x_special={3: np.inf, 40: np.nan, 1279: -np.inf})
        assert h.query("mv5_tiles") > 0
        h = check_spmv_mv(be, mv5_cases()[3][1], 7, "N", 1.0, 0.0, "F", yo, algo="SPMV_DEFAULT", max_val=1.5, nans=True, x_special={100: np.inf, 499: np.nan})
        assert h.query("mv5_tiles") > 0
    # forced (mv5 = 2: no fill threshold) on matrices it would not take, off (mv5 = 0), the gather kernel asked for, no analysis
    sparse = oracle.random_crs(800, 800, 9, variance=3, seed=5, sorted_rows=True)
    h = check_spmv_mv(be, sparse, 16, "N", 1.5, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=1.5, nans=True, knobs={"mv5": 2})
    assert h.query("mv5_tiles") > 10 and h.query("mv5_fill_permille") < 100, (h.query("mv5_tiles"), h.query("mv5_fill_permille"))
    h = check_spmv_mv(be, oracle.laplace2d("FE", 130, 41), 21, "N", 1.0, 1.0, "F", "F", algo="SPMV_DEFAULT", max_val=32.0, knobs={"mv5": 2})
    assert h.query("mv5_tiles") > 0
    # tiles whose union takes several chunks of 64 column blocks: 16 rows x 100 entries, no column shared (400 blocks per tile)
    wide = oracle.random_crs(48, 200000, 100, variance=0, seed=9, sorted_rows=True)
    for nvec, yo in ((16, "C"), (32, "F")):
        h = check_spmv_mv(be, wide, nvec, "N", 1.5, 0.0, "C", yo, algo="SPMV_DEFAULT", max_val=1.5, nans=True, knobs={"mv5": 2})
        assert h.query("mv5_tiles") >= 2 and h.query("mv5_blocks") > 2 * 300, (h.query("mv5_tiles"), h.query("mv5_blocks"))
    for algo, knobs in (("SPMV_DEFAULT", {"mv5": 0}), ("SPMV_DEFAULT", {"mv_kernel": 2}), ("SPMV_FAST_SETUP", None)):
        h = check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo=algo, knobs=knobs, max_val=1.5)
        assert h.query("mv5_tiles") == 0


def mv6_cases():
    """(name, matrix) for the nonzero-split rank-2 kernel: chunks of 128 entries; rows cut by one and by many chunk boundaries, runs
    that end exactly at a boundary, empty rows (leading, trailing, in between), fewer entries than a chunk, one row only"""
    out = [("hubs", hub_matrix(3000, 9000, 6, {5: 7000, 17: 1500, 1234: 1025, 2999: 4000, 40: 1024}, seed=3))]
    rng = np.random.default_rng(31)
    def from_lens(lens, ncols, seed):
        lens = np.asarray(lens, dtype=np.int64)
        rm = np.zeros(lens.size + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
        r_ = np.random.default_rng(seed)
        ent = np.concatenate([np.sort(r_.choice(ncols, size=int(l), replace=False)) for l in lens] + [np.empty(0, np.int64)]).astype(np.int32)
        return oracle.Crs(lens.size, ncols, rm, ent, r_.random(int(rm[-1])) + 0.5)
    # every row exactly one chunk; rows of 64 (two per chunk); 128 + 1 (every boundary cuts a row one entry in)
    out.append(("rows of 128", from_lens([128] * 40, 4000, 1)))
    out.append(("rows of 64", from_lens([64] * 50, 4000, 2)))
    out.append(("rows of 129", from_lens([129] * 33, 4000, 3)))
    # empty rows everywhere, a row of 1000 in the middle, short rows
    lens = rng.integers(0, 9, 600); lens[:5] = 0; lens[-7:] = 0; lens[100:140] = 0; lens[300] = 1000; lens[301] = 0; lens[302] = 300
    out.append(("empty rows + long", from_lens(lens, 5000, 4)))
    out.append(("tiny", from_lens([3, 0, 5, 1], 50, 5)))
    out.append(("one row", from_lens([700], 2000, 6)))
    return out


def check_mv6(be, light=False):
    combos = ((16, "C", "C", 1.5, 0.5, np.int32), (16, "F", "F", 1.0, 0.0, np.int64), (5, "C", "F", 2.0, 0.0, np.int32), (33, "F", "C", -1.0, 1.0, np.int32))
    for ci, (name, A0) in enumerate(mv6_cases()):
        for nvec, xo, yo, alpha, beta, off in (combos if (ci < 2 or not light) else combos[:2]):
            h = check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", max_val=50.0, nans=(beta == 0.0), offset_dtype=off, knobs={"mv6": 2})
            assert h.query("mv6_chunks") == -(-A0.nnz // 128), (name, h.query("mv6_chunks"))
            assert h.query("mv6_empty_rows") == int((np.diff(A0.row_map) == 0).sum()), name
    name, A0 = mv6_cases()[0]
    # by default on matrices with a tenth of their entries in long rows; fp32 values; Inf / NaN in X
    h = check_spmv_mv(be, A0, 16, "N", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", max_val=50.0, value_dtype=np.float32)
    assert h.query("mv6_chunks") > 0 and h.query("mv_long_nnz") == 7000 + 1500 + 1025 + 4000 + 1024
    h = check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=50.0, nans=True, x_special={int(A0.entries[0]): np.inf, 40: np.nan, 8999: -np.inf})
    assert h.query("mv6_chunks") > 0
    # not its matrices / off / the gather kernel asked for / no analysis
    h = check_spmv_mv(be, randomized(oracle.random_crs(2000, 2000, 9, variance=3, seed=5)), 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT")
    assert h.query("mv6_chunks") == 0
    for algo, knobs in (("SPMV_DEFAULT", {"mv6": 0}), ("SPMV_DEFAULT", {"mv_kernel": 2}), ("SPMV_FAST_SETUP", None)):
        h = check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo=algo, knobs=knobs, max_val=50.0)
        assert h.query("mv6_chunks") == 0


def check_transpose_plan_inherits_knobs(be):
    """Modes T / H of an analysed handle run the mode-N dispatch on the plan of its cached transpose: what the caller set on the handle --
    a forced rank-2 kernel, the nonzero-split kernel switched off -- must hold there too, whether it was set before the transpose exists
    (replayed at its creation) or afterwards (forwarded); round-4 advisor finding."""
    lib = be.lib
    kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 0))
    try:
        name, A0 = mv6_cases()[0]
        At0 = oracle.transpose(A0)
        # the transpose of this matrix is one the nonzero-split kernel takes by default ...
        h = check_spmv_mv(be, A0, 16, "T", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=50.0, knobs={"mv6": 2})
        assert h.query("transpose_plan_mv6_chunks") == -(-At0.nnz // 128), h.query("transpose_plan_mv6_chunks")
        # ... a handle with the gather kernel forced BEFORE the first transposed call keeps it away from the transpose's plan ...
        h = check_spmv_mv(be, A0, 16, "T", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", max_val=50.0, knobs={"mv6": 2, "mv_kernel": 2})
        assert h.query("transpose_plan_tile") > 0 and h.query("transpose_plan_mv6_chunks") == 0
        # ... and setting it AFTERWARDS reaches the existing transposed plan
        h = check_spmv_mv(be, A0, 16, "T", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=50.0, knobs={"mv6": 2})
        assert h.query("transpose_plan_mv6_chunks") > 0
        h.set("mv6", 0)
        assert h.query("transpose_plan_mv6_chunks") == 0
        assert h.query("transpose_bytes") >= At0.nnz * (4 + 4 + 8 + 8)          # structure, permutation, values and the shadow copy of exact tracking
    finally:
        kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 1000))


def check_rank2_first_handle_defers_rank1(be):
    """A handle whose first call is rank 2 (more than one column) leaves the rank-1 analysis -- tiles, window codes, pattern records -- to
    the first rank-1 call (knob defer_rank1, set by SPMVHandle; the reference's handle is set up by its first call, for that call's rank,
    KokkosSparse_spmv_handle.hpp:280-349): rank 2, rank 1, rank 2 again on one handle, modes N and T, each against the oracle; the rank-1
    plan does not exist before the rank-1 call and does afterwards; a handle that begins with rank 1 is analysed at once, as before; the
    knob can be switched off on the handle."""
    lib = be.lib
    kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 0))
    try:
        for A0 in (oracle.laplace3d("FE", 9, 8, 7), oracle.random_crs(900, 700, 14, variance=6, seed=21)):
            A = dev(be, A0)
            rng = np.random.default_rng(5)
            for mode in ("N", "T"):
                trans = mode == "T"
                nin, nout = (A0.nrows, A0.ncols) if trans else (A0.ncols, A0.nrows)
                X = np.asarray(rng.random((nin, 6)), order="C"); Y0 = np.asarray(rng.random((nout, 6)), order="C")
                x = rng.random(nin); y0 = rng.random(nout)
                tol2 = oracle.spmv_max_error(A0, 1.5, 0.5) * (max(1.0, np.bincount(A0.entries, minlength=A0.ncols).max() / max(np.diff(A0.row_map).max(), 1)) if trans else 1.0)
                h = kk.SPMVHandle("SPMV_DEFAULT")
                key = "transpose_plan_tiles" if trans else "tiles"
                def rank2():
                    Yd = _to_dev_2d(be, Y0)
                    kk.spmv(h, mode, 1.5, A, _to_dev_2d(be, X), 0.5, Yd)
                    exp = oracle.spmv_mv_serial(mode, A0, 1.5, X, 0.5, Y0.copy(order="K"))
                    assert np.abs(_to_host_2d(be, Yd) - exp).max() <= tol2, (mode, "rank 2")
                def rank1():
                    yd = be.from_numpy(y0)
                    kk.spmv(h, mode, 1.5, A, be.from_numpy(x), 0.5, yd)
                    exp = oracle.spmv_serial(mode, A0, 1.5, x, 0.5, y0.copy())
                    assert np.abs(be.to_numpy(yd) - exp).max() <= tol2, (mode, "rank 1")
                rank2()
                assert h.query("tile") > 0 and h.query("tiles") == 0, (h.query("tile"), h.query("tiles"))      # analysed handle, no rank-1 plan yet
                if trans: assert h.query("transpose_plan_tile") > 0 and h.query(key) == 0
                rank1()
                assert h.query(key) > 0, (mode, h.query(key))
                rank2(); rank1()
                # rank 1 first: analysed at creation
                h = kk.SPMVHandle("SPMV_DEFAULT")
                rank1()
                assert h.query(key) > 0
                rank2()
                # the knob switched off by the caller
                h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("defer_rank1", 0)
                rank2()
                assert h.query("tiles") > 0
    finally:
        kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 1000))


def check_colslab_deterministic(be):
    """The DETERMINISTIC column-slab form of rank-1 mode N (round 5; `colslab` 4 forces it, 3 -- the default -- chooses it by rule): per-slab
    partial sums of every row, stored by exactly one writer, cut runs summed in chunk order, slabs added in ascending order -- no atomics.
    Cases: rows longer than a wave's chunk of 512 entries inside ONE slab (runs cut by several chunk boundaries: head / tail / whole-chunk
    pieces), empty rows, rows with one entry per slab, duplicate columns, a matrix wider than tall, the last chunk shorter than a round,
    beta != 0, NaN in y with beta = 0, fp32 values / vectors, 64-bit offsets, value changes under exact tracking -- and the SAME BITS
    from two handles and from repeated calls."""
    import struct
    rng = np.random.default_rng(4)
    n, k = 1200, 5000
    lens = rng.integers(0, 12, size=n); lens[3] = 1700; lens[4] = 0; lens[5] = 3000; lens[700] = 4999; lens[n - 1] = 530
    rm = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = np.concatenate([np.sort(rng.choice(k, size=l, replace=False)) for l in lens]).astype(np.int32)
    ent[rm[5]:rm[5] + 2000] = np.sort(rng.integers(0, 60, size=2000))                 # 2000 entries (duplicates) of one row inside the first slab: a run over four chunks
    A0 = oracle.Crs(n, k, rm, ent, 1 + 49 * rng.random(rm[-1]))
    wide = oracle.random_crs(300, 9000, 40, variance=10, seed=6)
    for M in (A0, wide):
        for shift in (6, 9, 13):
            kn = {"colslab": 4, "colslab_shift": shift}
            h = check_spmv(be, M, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs=kn, max_val=50.0, expect={"colslab": 1, "colslab_deterministic": 1})
            assert h.query("colslab_slabs") <= 64
            check_spmv(be, M, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs=kn, max_val=50.0, nans=True)
        check_spmv(be, M, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"colslab": 4, "colslab_shift": 7}, max_val=50.0, offset_dtype=np.int64, value_dtype=np.float32)
        check_spmv(be, M, "N", 1.0, 1.0, "SPMV_DEFAULT", knobs={"colslab": 4, "colslab_shift": 7, "colslab_const": 1}, max_val=50.0, value_dtype=np.float32, vec_dtype=np.float32)
    # the same bits: two handles, repeated calls, after a value change and back
    A = dev(be, A0)
    x = rng.random(k); xd = be.from_numpy(x)
    outs = []
    for rep in range(2):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("colslab", 4); h.set("colslab_shift", 6)
        yd = be.from_numpy(np.zeros(n))
        for call in range(2):
            kk.spmv(h, "N", 1.0, A, xd, 0.0, yd)
            outs.append(be.to_numpy(yd).copy())
        assert h.query("colslab_deterministic") == 1
    for o in outs[1:]:
        assert o.tobytes() == outs[0].tobytes(), "the deterministic column-slab form gave different bits"
    exp = oracle.spmv_serial("N", A0, 1.0, x, 0.0, np.zeros(n))
    np.testing.assert_allclose(outs[0], exp, rtol=1e-11, atol=1e-9)
    v = A0.values.copy(); v[::7] *= -2.0
    A.values[:] = be.from_numpy(v)                                # exact tracking (default): the copy follows
    kk.spmv(h, "N", 1.0, A, xd, 0.0, yd)
    exp2 = oracle.spmv_serial("N", oracle.Crs(n, k, rm, ent, v), 1.0, x, 0.0, np.zeros(n))
    np.testing.assert_allclose(be.to_numpy(yd), exp2, rtol=1e-11, atol=1e-9)
    # one handle, BOTH vector types (round-5 advice: the partial-sum buffers were sized by the first call's type and the other type got
    # KKAMD_ERR_STATE; the CRS kernels take either on one handle): a float matrix with fp64 vectors, then fp32 vectors, then fp64 again
    v32 = v.astype(np.float32)
    A32 = dev(be, oracle.Crs(n, k, rm, ent, v32.astype(np.float64)), value_dtype=np.float32)
    h32 = kk.SPMVHandle("SPMV_DEFAULT"); h32.set("colslab", 4); h32.set("colslab_shift", 6)
    exp64 = oracle.spmv_serial("N", oracle.Crs(n, k, rm, ent, v32.astype(np.float64)), 1.0, x, 0.0, np.zeros(n))
    x32 = x.astype(np.float32)
    exp32 = oracle.spmv_serial("N", oracle.Crs(n, k, rm, ent, v32.astype(np.float64)), 1.0, x32.astype(np.float64), 0.0, np.zeros(n))
    for vec32 in (False, True, False):
        if vec32:
            y32 = be.from_numpy(np.zeros(n, dtype=np.float32))
            kk.spmv(h32, "N", 1.0, A32, be.from_numpy(x32), 0.0, y32)
            np.testing.assert_allclose(be.to_numpy(y32).astype(np.float64), exp32, rtol=2e-4, atol=1e-2)
        else:
            kk.spmv(h32, "N", 1.0, A32, xd, 0.0, yd)
            np.testing.assert_allclose(be.to_numpy(yd), exp64, rtol=1e-11, atol=1e-9)
        assert h32.query("colslab_deterministic") == 1
    # the rule of the default handle says no on a small matrix (x is not several L2s large): nothing is built
    check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=50.0, expect={"colslab": 0, "colslab_tried": 1})


def check_values_tracking(be):
    """The re-ordered value copies of a plan (cached transpose, column-slab copy) under the three "values_tracking" policies: 0 exact
    (default), 1 notify (SPMVHandle.values_changed), 2 fingerprints.  A.values is rewritten in place between calls: one value, every value,
    two values swapped, a whole 4096-value tile negated and another doubled (the changes the round-3 linear fingerprint was blind or
    nearly blind to), then nothing."""
    lib = be.lib
    kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 0))
    try:
        A0 = oracle.random_crs(2500, 2300, 10, variance=3, seed=8, sorted_rows=True)
        rng = np.random.default_rng(1)
        for what, mode, knobs in (("transpose", "T", {"explicit_transpose": 1}), ("colslab", "N", {"colslab": 2, "colslab_shift": 5})):
            nin, nout = (A0.nrows, A0.ncols) if mode == "T" else (A0.ncols, A0.nrows)
            x = rng.random(nin)
            for tracking in (0, 1, 2):
                A = dev(be, A0)
                h = kk.SPMVHandle("SPMV_DEFAULT")
                for k_, v_ in knobs.items(): h.set(k_, v_)
                h.set("values_tracking", tracking)
                xd, yd = be.from_numpy(x), be.from_numpy(np.zeros(nout))
                v = A0.values.copy()
                def run_and_check(tag):
                    kk.spmv(h, mode, 1.0, A, xd, 0.0, yd)
                    exp = oracle.spmv_sequential(mode, oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, v), 1.0, x, 0.0, np.zeros(nout))
                    np.testing.assert_allclose(be.to_numpy(yd), exp, rtol=1e-12, atol=1e-12, err_msg="%s tracking %d after %s" % (what, tracking, tag))
                run_and_check("the first call")
                assert h.query("transpose_cached" if mode == "T" else "colslab") == 1
                steps = (("one value", lambda: v.__setitem__(5000, -7.25)), ("every value", lambda: v.__imul__(3.0)),
                         ("a swap", lambda: v.__setitem__([0, len(v) - 1], v[[len(v) - 1, 0]])),
                         ("a tile negated", lambda: v.__setitem__(slice(4096, 8192), -v[4096:8192])),
                         ("a tile doubled", lambda: v.__setitem__(slice(8192, 12288), 2.0 * v[8192:12288])), ("nothing", lambda: None))
                for tag, change in steps:
                    change()
                    A.values[:] = be.from_numpy(v)
                    if tracking == 1 and tag != "nothing": h.values_changed()
                    run_and_check(tag)
                # a notification is harmless under the other policies, and a change of policy starts over
                h.values_changed(); run_and_check("a notification without a change")
                h.set("values_tracking", (tracking + 1) % 3)
                v[17] = 0.125; A.values[:] = be.from_numpy(v)
                if (tracking + 1) % 3 == 1: h.values_changed()
                run_and_check("a change of policy")
    finally:
        kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 1000))


def check_entries_guard(be):
    """knob check_entries: a column array edited in place under a live handle is reported, an untouched one is not"""
    import pytest
    A0 = oracle.laplace3d("FE", 12, 11, 10)
    A = dev(be, A0)
    h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("check_entries", 1)
    x, y = be.from_numpy(np.ones(A0.ncols)), be.from_numpy(np.zeros(A0.nrows))
    kk.spmv(h, "N", 1.0, A, x, 0.0, y); kk.spmv(h, "N", 1.0, A, x, 0.0, y)
    A.values[:] = be.from_numpy(A0.values * 2.0)                     # values may change
    kk.spmv(h, "N", 1.0, A, x, 0.0, y)
    e = be.to_numpy(A.graph.entries).copy(); e[5], e[6] = e[6], e[5]
    A.graph.entries[:] = be.from_numpy(e)                            # the structure may not
    with pytest.raises(kk.KkamdError) as ei:
        kk.spmv(h, "N", 1.0, A, x, 0.0, y)
    assert ei.value.status == kk._capi.ERR_STATE
    h2 = kk.SPMVHandle("SPMV_DEFAULT")                               # without the knob nothing is checked (and nothing is paid)
    kk.spmv(h2, "N", 1.0, A, x, 0.0, y)


def check_mv_transpose_cached(be, light=False):
    """Rank 2, modes T / H of an analysed handle run the mode-N dispatch on the cached transpose (no atomics): the four layout pairs,
    a value update between the calls, a lattice matrix (its transpose takes the plane-marching kernel), a rectangular one, 64-bit
    offsets; explicit_transpose 0 keeps the reference's atomic scatter."""
    lib = be.lib
    kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 0))
    try:
        mats = ((oracle.random_crs(900, 700, 9, variance=3, seed=7, sorted_rows=True), 1.0), (oracle.laplace3d("FE", 34, 9, 8), 32.0))
        combos = list(zip((("C", "C"), ("F", "F"), ("C", "F"), ("F", "C")), (16, 5, 21, 32), (np.int32, np.int64, np.int32, np.int32)))
        if light:                                                  # the emulator: the unstructured matrix, two layout pairs (the GPU suite runs them all)
            mats, combos = mats[:1], combos[:2]
        for A0, maxv in mats:
            for (xo, yo), nvec, off in combos:
                h = check_spmv_mv(be, A0, nvec, "T", 1.5, 0.0, xo, yo, algo="SPMV_DEFAULT", max_val=maxv, nans=True, offset_dtype=off)
                assert h.query("transpose_cached") == 1
                check_spmv_mv(be, A0, nvec, "H", 1.0, -0.5, xo, yo, algo="SPMV_DEFAULT", max_val=maxv, offset_dtype=off)
        A0 = oracle.random_crs(900, 700, 9, variance=3, seed=7, sorted_rows=True)
        A = dev(be, A0)
        h = kk.SPMVHandle("SPMV_DEFAULT")
        rng = np.random.default_rng(4)
        X = rng.random((A0.nrows, 16)); v = A0.values.copy()
        for rep in range(3):
            Yd = _to_dev_2d(be, np.zeros((A0.ncols, 16)))
            kk.spmv(h, "T", 1.0, A, _to_dev_2d(be, X), 0.0, Yd)
            exp = oracle.spmv_mv_serial("T", oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, v), 1.0, X, 0.0, np.zeros((A0.ncols, 16)))
            np.testing.assert_allclose(_to_host_2d(be, Yd), exp, rtol=1e-12, atol=1e-12)
            v = rng.random(A0.nnz) - rep
            A.values[:] = be.from_numpy(v)
        h = check_spmv_mv(be, A0, 16, "T", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", knobs={"explicit_transpose": 0})
        assert h.query("transpose_cached") == 0
    finally:
        kk._capi.check(lib, lib.kkamd_set_default(b"explicit_transpose_min_knnz", 1000))


def _to_dev_2d(be, M):
    """device 2-D array with the same logical layout (Fortran order kept through a transposed view)"""
    if be.name == "emu":
        return np.array(M, order="K", copy=True)
    import torch
    if M.flags.f_contiguous and not M.flags.c_contiguous:
        return torch.from_numpy(np.ascontiguousarray(M.T)).to("cuda").t()
    return torch.from_numpy(np.ascontiguousarray(M)).to("cuda")


def _to_host_2d(be, M):
    if be.name == "emu":
        return np.asarray(M)
    return M.detach().cpu().numpy()


def check_spgemm(be, A0, B0, offset_dtype=np.int32, reuse=True, value_dtype=np.float64, algo="SPGEMM_KK", options=None, expect_compressed=None):
    A, B = dev(be, A0, offset_dtype, value_dtype), dev(be, B0, offset_dtype, value_dtype)
    kh = kk.KokkosKernelsHandle(be)
    kh.create_spgemm_handle(algo)
    for k_, v_ in (options or {}).items():
        kh.get_spgemm_handle().set(k_, v_)
    Cm = kk.spgemm_symbolic(kh, A, False, B, False)
    Cgold = oracle.spgemm(A0, B0)
    rmC = be.to_numpy(Cm.graph.row_map).astype(np.int64)
    assert kh.get_spgemm_handle().get_c_nnz() == Cgold.nnz
    assert np.array_equal(rmC, Cgold.row_map), "symbolic row_map differs"
    if expect_compressed is not None:
        assert bool(kh.get_spgemm_handle().get(6)) == expect_compressed, "compression decision: %d (work %d of %d)" % (
            kh.get_spgemm_handle().get(6), kh.get_spgemm_handle().get(7), kh.get_spgemm_handle().get(1))
    kk.spgemm_numeric(kh, A, False, B, False, Cm)
    rm, ent, val = Cm.to_host()
    got = oracle.Crs(A0.nrows, B0.ncols, rm.astype(np.int64), ent, val.astype(np.float64))
    eps = 1e-7 if value_dtype == np.float64 else 3.7e-3
    if value_dtype != np.float64:
        Ag = oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, A0.values.astype(np.float32).astype(np.float64))
        Bg = oracle.Crs(B0.nrows, B0.ncols, B0.row_map, B0.entries, B0.values.astype(np.float32).astype(np.float64))
        Cgold = oracle.spgemm(Ag, Bg)
    ok, msg = oracle.is_same_matrix(got, Cgold, eps)
    assert ok, "spgemm: " + msg
    if reuse and A0.nnz and B0.nnz:
        # numeric again with new values on the same handle (Test_Sparse_spgemm.hpp:243-252)
        rng = np.random.default_rng(99)
        A2 = oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, 1 + 49 * rng.random(A0.nnz))
        Ad2 = kk.CrsMatrix(A0.nrows, A0.ncols, A.graph.row_map, A.graph.entries, be.from_numpy(A2.values.astype(value_dtype)), backend=be)
        kk.spgemm_numeric(kh, Ad2, False, B, False, Cm)
        rm, ent, val = Cm.to_host()
        got = oracle.Crs(A0.nrows, B0.ncols, rm.astype(np.int64), ent, val.astype(np.float64))
        if value_dtype != np.float64:
            A2 = oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, A2.values.astype(np.float32).astype(np.float64))
            Cg2 = oracle.spgemm(A2, Bg)
        else:
            Cg2 = oracle.spgemm(A2, B0)
        ok, msg = oracle.is_same_matrix(got, Cg2, eps)
        assert ok, "spgemm numeric reuse: " + msg
    kh.destroy_spgemm_handle()
    return got


def check_spgemm_kept_structure(be):
    """The symbolic phase keeps, for the first numeric call, the bitmaps of its densest rows (>= k/32 entries) and the ENTRY LISTS of the
    other dense rows (round 4; lists above 6144 entries leave LDS in several rounds): structure identical to the oracle's, both sources
    used, numeric reuse keeps the entries, and with the lists switched off the same C comes out"""
    import fuzz_cases as fz
    rng = np.random.default_rng(5)
    n, k = 60, 400000
    B = fz.hubby(rng, n, k, 3000, 0, 3000)            # rows of B: 0 .. 6000 entries
    A = fz.hubby(rng, 10, n, 3, 2, 20)                # rows of A: 0 .. 6 entries, two of about 20
    gold = oracle.spgemm(A, B)
    # rows of C that are 30 .. 100 % dense (kept bitmaps whose 64-word steps hold more than a wave's 1024 staging slots: halves, quarters)
    kd = 8192
    lens = rng.integers(1500, 6000, size=30)
    rmd = np.zeros(31, dtype=np.int64); np.cumsum(lens, out=rmd[1:])
    Bd0 = oracle.Crs(30, kd, rmd, np.concatenate([np.sort(rng.choice(kd, size=l, replace=False)) for l in lens]).astype(np.int32), 1 + 49 * rng.random(rmd[-1]))
    rows_d = [np.sort(rng.choice(30, size=c, replace=False)) for c in (1, 2, 3, 6, 12, 30)]
    armd = np.zeros(len(rows_d) + 1, dtype=np.int64); np.cumsum([len(r) for r in rows_d], out=armd[1:])
    Ad0 = oracle.Crs(len(rows_d), 30, armd, np.concatenate(rows_d).astype(np.int32), 1 + 49 * rng.random(armd[-1]))
    got_d = check_spgemm(be, Ad0, Bd0)
    assert np.diff(got_d.row_map).max() > 8000 and np.diff(got_d.row_map).min() >= 1500
    kh = kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
    Cd = kk.spgemm_symbolic(kh, dev(be, Ad0), False, dev(be, Bd0), False)
    kk.spgemm_numeric(kh, dev(be, Ad0), False, dev(be, Bd0), False, Cd)
    assert kh.get_spgemm_handle().get(12) >= 4, kh.get_spgemm_handle().get(12)          # written from kept bitmaps
    kh.destroy_spgemm_handle()
    for lists in (1, 0):
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_keep_lists", lists))
        try:
            kh = kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
            Ad, Bd = dev(be, A), dev(be, B)
            Cm = kk.spgemm_symbolic(kh, Ad, False, Bd, False)
            kk.spgemm_numeric(kh, Ad, False, Bd, False, Cm)
            sh = kh.get_spgemm_handle()
            assert sh.get(12) > 0 and (sh.get(14) > 0) == bool(lists), (sh.get(12), sh.get(14))
            rm, ent, val = Cm.to_host()
            ok, msg = oracle.is_same_matrix(oracle.Crs(A.nrows, B.ncols, rm.astype(np.int64), ent, val.astype(np.float64)), gold)
            assert ok, msg
            kk.spgemm_numeric(kh, Ad, False, Bd, False, Cm)            # reuse: entries kept
            assert sh.get(11) == 1
            rm2, ent2, val2 = Cm.to_host()
            assert np.array_equal(ent, ent2) and np.allclose(val, val2, rtol=1e-13)
            kh.destroy_spgemm_handle()
        finally:
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_keep_lists", 1))


def check_spgemm_units(be, light=False):
    """Round 6: the symbolic phase counts its dense class by UNITS (row of C, window of 2^unit_bits columns; an index of B at window
    granularity cuts every list).  Windows of 64 .. 2^18 columns over products whose rows are sparse, dense, a single list, hundreds of
    lists (several chunks of 256), lists shorter than a quad, empty pieces in most windows; 32- and 64-bit offsets; with the bitmaps,
    the lists or both switched off; against the one-workgroup-per-row kernel (spgemm_sym_units 0).  Structure identical to the oracle's
    every time (check_spgemm), and the queries say the units were used."""
    import fuzz_cases as fz
    rng = np.random.default_rng(61)
    setd = lambda key, v: kk._capi.check(be.lib, be.lib.kkamd_set_default(key, v))
    # (a) hub lists next to short ones, k not a multiple of any window; (b) dense rows; (c) an A row of 700 lists (three chunks), B rows of 0 .. 30 entries
    n, k = 60, 100003
    B = fz.hubby(rng, n, k, 2500, 0, 2500)
    A = fz.hubby(rng, 10, n, 3, 2, 20)
    kd = 8192
    lens = rng.integers(1500, 6000, size=30)
    rmd = np.zeros(31, dtype=np.int64); np.cumsum(lens, out=rmd[1:])
    Bd = oracle.Crs(30, kd, rmd, np.concatenate([np.sort(rng.choice(kd, size=l, replace=False)) for l in lens]).astype(np.int32), 1 + 49 * rng.random(rmd[-1]))
    rows_d = [np.sort(rng.choice(30, size=c, replace=False)) for c in (1, 2, 3, 6, 12, 30)]
    armd = np.zeros(len(rows_d) + 1, dtype=np.int64); np.cumsum([len(r) for r in rows_d], out=armd[1:])
    Ad = oracle.Crs(len(rows_d), 30, armd, np.concatenate(rows_d).astype(np.int32), 1 + 49 * rng.random(armd[-1]))
    Bs = randomized(oracle.random_crs(900, 30000, 14, variance=14, seed=23, sorted_rows=True))
    cols = np.sort(rng.choice(900, size=700, replace=False)).astype(np.int32)
    As = oracle.Crs(3, 900, np.array([0, 700, 703, 1100]), np.concatenate([cols, [1, 5, 9], np.sort(rng.choice(900, size=397, replace=False))]).astype(np.int32), 1 + 49 * rng.random(1100))
    cases = [(A, B, "hubs"), (Ad, Bd, "dense"), (As, Bs, "many short lists")]
    try:
        for bits in ((6, 12, 18) if light else (6, 9, 12, 15, 18)):
            setd(b"spgemm_unit_bits", bits)
            for A0, B0, name in cases:
                if bits == 6 and name == "hubs" : continue               # 1563 windows x 10 rows of one word each: slow under the emulator, nothing new
                for odt in (np.int32, np.int64):
                    got = check_spgemm(be, A0, B0, offset_dtype=odt, reuse=(bits == 12))
                kh = kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
                Am, Bm = dev(be, A0), dev(be, B0)
                Cm = kk.spgemm_symbolic(kh, Am, False, Bm, False)
                sh = kh.get_spgemm_handle()
                nwin = -(-B0.ncols // (1 << bits))
                assert sh.get(19) > 0 and sh.get(19) % nwin == 0, (name, bits, sh.get(19), nwin)          # class rows x windows
                assert sh.get(21) > 0 and sh.get(13) == sh.get(21), (name, bits, sh.get(21), sh.get(13))
                if name == "dense" and bits >= 9: assert sh.get(20) > 0, (bits, sh.get(20))               # units at least 1/32 dense keep bitmaps
                kk.spgemm_numeric(kh, Am, False, Bm, False, Cm)
                assert sh.get(12) > 0 and sh.get(13) == 0, (name, bits, sh.get(12), sh.get(13))
                kh.destroy_spgemm_handle()
        setd(b"spgemm_unit_bits", 10)
        for bm, ls in ((0, 1), (1, 0), (0, 0)):
            setd(b"spgemm_keep_bitmaps", bm); setd(b"spgemm_keep_lists", ls)
            for A0, B0, name in cases:
                check_spgemm(be, A0, B0, offset_dtype=np.int64, reuse=False)
        setd(b"spgemm_keep_bitmaps", 1); setd(b"spgemm_keep_lists", 1)
        setd(b"spgemm_sym_units", 0)                                      # the row-by-row kernel is still there (B with unsorted rows under several windows takes it)
        for A0, B0, name in cases:
            check_spgemm(be, A0, B0, reuse=False)
        setd(b"spgemm_sym_units", 1)
        # B with unsorted rows: one window -> units without an index; several windows -> row by row
        ent, val = B.entries.copy(), B.values.copy()
        for i in range(B.nrows):
            lo, hi = B.row_map[i], B.row_map[i + 1]
            q = rng.permutation(hi - lo)
            ent[lo:hi], val[lo:hi] = ent[lo:hi][q], val[lo:hi][q]
        Bu = oracle.Crs(B.nrows, B.ncols, B.row_map, ent, val)
        for bits in (18, 12):
            setd(b"spgemm_unit_bits", bits)
            check_spgemm(be, A, Bu, reuse=False)
        # a cap on the store (knob spgemm_store_cap_mb, for hosts that interleave their own allocations): room goes to the heaviest rows, the
        # others walk their products again in the numeric phase -- the same C; the queries report what the process-wide store holds and has held
        setd(b"spgemm_unit_bits", 12); setd(b"spgemm_store_cap_mb", 1)
        kk._capi.check(be.lib, be.lib.kkamd_release_scratch())
        kh = kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
        Am, Bm = dev(be, Ad), dev(be, Bd)
        Cm = kk.spgemm_symbolic(kh, Am, False, Bm, False)
        sh = kh.get_spgemm_handle()
        assert 0 < sh.get(22) <= 1 << 20 and sh.get(23) >= sh.get(22), (sh.get(22), sh.get(23))
        assert 0 < sh.get(21) < Ad.nrows, sh.get(21)                       # some rows complete, not all (the dense product wants 1.5 MB)
        kk.spgemm_numeric(kh, Am, False, Bm, False, Cm)
        rm_, ent_, val_ = Cm.to_host()
        ok, msg = oracle.is_same_matrix(oracle.Crs(Ad.nrows, Bd.ncols, rm_.astype(np.int64), ent_, val_.astype(np.float64)), oracle.spgemm(Ad, Bd))
        assert ok, msg
        kh.destroy_spgemm_handle()
    finally:
        setd(b"spgemm_store_cap_mb", 0)
        setd(b"spgemm_unit_bits", 18); setd(b"spgemm_keep_bitmaps", 1); setd(b"spgemm_keep_lists", 1); setd(b"spgemm_sym_units", 1)


def check_spgemm_pool_release(be):
    """The process-wide store of bitmaps / entry lists (GBs on large products) goes back to the device when the LAST SpGEMM handle is
    destroyed (round-4 review: a plain C caller kept it for ever); with "spgemm_pool_keep" 1 it outlives the handles and
    kkamd_release_scratch returns it.  Also: the store is sized by what the product can need, not by a flat share of the free memory."""
    import torch
    import fuzz_cases as fz
    rng = np.random.default_rng(5)
    n, k = 60, 400000
    B = fz.hubby(rng, n, k, 3000, 0, 3000)
    A = fz.hubby(rng, 10, n, 3, 2, 20)
    Ad, Bd = dev(be, A), dev(be, B)
    kk._capi.check(be.lib, be.lib.kkamd_release_scratch())
    torch.cuda.synchronize(); torch.cuda.empty_cache()

    def run(keep_a_second_handle=False):
        kh = kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
        other = None
        if keep_a_second_handle:
            other = kk.KokkosKernelsHandle(be); other.create_spgemm_handle("SPGEMM_KK")
        Cm = kk.spgemm_symbolic(kh, Ad, False, Bd, False)
        held = torch.cuda.mem_get_info()[0]                   # free memory while the handle holds its store
        kk.spgemm_numeric(kh, Ad, False, Bd, False, Cm)
        assert kh.get_spgemm_handle().get(12) + kh.get_spgemm_handle().get(14) > 0           # the store was used
        del Cm
        kh.destroy_spgemm_handle()
        return held, other

    kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_pool_keep", 2))       # always returned with the last handle
    try:
        free0 = torch.cuda.mem_get_info()[0]
        held, _ = run()
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        free1 = torch.cuda.mem_get_info()[0]
        assert free0 - held < (1 << 30), "store of %d MB for a product with 6e4 multiplications" % ((free0 - held) >> 20)    # (it was a tenth of the free HBM)
        assert free0 - free1 < (8 << 20), "the last handle is gone and %d MB are still held" % ((free0 - free1) >> 20)
        # a second live handle keeps the pool (the next product of the job will use it) until it is destroyed too
        _, other = run(keep_a_second_handle=True)
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        free2 = torch.cuda.mem_get_info()[0]
        other.destroy_spgemm_handle()
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        free3 = torch.cuda.mem_get_info()[0]
        assert free3 >= free2 and free0 - free3 < (8 << 20), (free0, free2, free3)
        # 1 = it outlives the handles until kkamd_release_scratch
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_pool_keep", 1))
        run()
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        free4 = torch.cuda.mem_get_info()[0]
        kk._capi.check(be.lib, be.lib.kkamd_release_scratch())
        torch.cuda.synchronize()
        free5 = torch.cuda.mem_get_info()[0]
        assert free5 >= free4 and free0 - free5 < (8 << 20), (free0, free4, free5)
        # 0 (default): whatever the process did before, after a release the next product's store is returned or kept -- and kkamd_release_scratch
        # always returns it
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_pool_keep", 0))
        run(); run()
        kk._capi.check(be.lib, be.lib.kkamd_release_scratch())
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        assert free0 - torch.cuda.mem_get_info()[0] < (8 << 20)
    finally:
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_pool_keep", 0))


def galerkin_operands(n):
    """27-point FE Laplacian n^3 (n even), P of a 2 x 2 x 2 aggregation (one entry per row), R = P^T and A P (by the oracle)"""
    A = oracle.laplace3d("FE", n, n, n)
    i = np.arange(n); c = n // 2
    ci = (i[:, None, None] // 2) * c * c + (i[None, :, None] // 2) * c + (i[None, None, :] // 2)
    nf = n ** 3
    P = oracle.Crs(nf, c ** 3, np.arange(nf + 1, dtype=np.int64), ci.reshape(-1).astype(np.int32), np.ones(nf))
    order = np.argsort(P.entries, kind="stable")                                       # R = P^T: coarse node -> its (up to) eight fine nodes
    rrm = np.zeros(c ** 3 + 1, dtype=np.int64); np.cumsum(np.bincount(P.entries, minlength=c ** 3), out=rrm[1:])
    R = oracle.Crs(c ** 3, nf, rrm, order.astype(np.int32), np.ones(nf))
    return A, P, R, oracle.spgemm(A, P)


def check_spgemm_galerkin(be):
    """The two products of a Galerkin coarse operator R A P (Test_Sparse_spgemm.hpp:491-504 lists rectangular shapes; the perf driver
    KokkosSparse_spgemm.cpp:372-423 takes any pair): A P -- 27 products and 1 .. 8 entries per row, a rectangular B with one entry per
    row -- and R (A P) -- up to 216 products and 27 entries per row: a product whose wave bin mixes the four-rows-per-wave kernels with the
    one-row kernels.  Structure bit-exact, values to 1e-7, numeric reuse, both offset widths."""
    A, P, R, AP = galerkin_operands(12)
    got = check_spgemm(be, A, P)
    assert got.nnz == AP.nnz
    C = check_spgemm(be, R, AP, offset_dtype=np.int64)
    assert 8 <= np.diff(C.row_map).min() and np.diff(C.row_map).max() == 27
    check_spgemm(be, R, AP, value_dtype=np.float32)


def check_spgemm_block_kernel(be):
    """Column-block value kernel (round 5, `spgemm_block`): rows of C that are dense, or have more lists than the flat kernel's shapes
    hold, accumulate per (row, column block) in a direct-indexed LDS accumulator, with the pieces of every list taken from an index of
    B.  Cases: k not a multiple of the block width, blocks with no entry of a row, an A row of 2500 lists (three chunks of lists into
    one accumulator), a row that is 100 % dense, lists that end B's arrays (guarded 16-byte loads), empty B rows inside an A row,
    numeric reuse (the index of B is kept), 64-bit offsets and fp32.  Block widths 256 (many blocks), 4096 and the default."""
    rng = np.random.default_rng(23)
    n, k = 2600, 9000 + 37
    lens = rng.integers(0, 40, size=n); lens[:6] = (k, 3000, 2500, 0, 1, 700); lens[-1] = 1501
    rm = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = np.concatenate([np.sort(rng.choice(k, size=l, replace=False)) for l in lens]).astype(np.int32)
    B = oracle.Crs(n, k, rm, ent, 1 + 49 * rng.random(rm[-1]))
    rows = [np.arange(0, 2500),                          # 2500 lists, 100 % dense (contains row 0 of B)
            np.array([1, 2, 5, n - 1]),                  # four long lists: ~50 % dense
            np.arange(6, 700),                           # 694 short lists (more than the flat shapes take), ~2 entries per 256-column block ...
            np.array([3, 4]), np.array([n - 1]), np.array([], dtype=np.int64),
            np.sort(rng.choice(n, size=600, replace=False))]
    arm = np.zeros(len(rows) + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=arm[1:])
    A = oracle.Crs(len(rows), n, arm, np.concatenate(rows).astype(np.int32), 1 + 49 * rng.random(arm[-1]))
    try:
        # items: groups of blocks with a position-indexed accumulator (cap: entries per group; 64 forces the dense blocks to stay column-indexed
        # items next to groups), or (items 0) one workgroup per (row, block)
        for w, odt, vdt, items, cap in ((256, np.int32, np.float64, 1, 6144), (256, np.int32, np.float64, 1, 64), (256, np.int64, np.float64, 1, 200),
                                        (4096, np.int64, np.float64, 1, 1000), (16384, np.int32, np.float32, 1, 6144), (256, np.int64, np.float32, 1, 100),
                                        (256, np.int32, np.float64, 0, 6144), (4096, np.int64, np.float32, 0, 6144)):
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_block_w", w))
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_items", items))
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_item_cap", cap))
            check_spgemm(be, A, B, offset_dtype=odt, value_dtype=vdt)
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_item_cap", 100))
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_items", 1))
        # the rows really went there (and stay away with the knob off: same C)
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_block_w", 1024))
        for on in (1, 0):
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_block", on))
            kh = kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
            Ad, Bd = dev(be, A), dev(be, B)
            Cm = kk.spgemm_symbolic(kh, Ad, False, Bd, False)
            kk.spgemm_numeric(kh, Ad, False, Bd, False, Cm)
            assert (kh.get_spgemm_handle().get(16) >= 3) == bool(on), kh.get_spgemm_handle().get(16)
            if on: assert kh.get_spgemm_handle().get(17) + kh.get_spgemm_handle().get(18) > 0, (kh.get_spgemm_handle().get(17), kh.get_spgemm_handle().get(18))
            rm_, ent_, val_ = Cm.to_host()
            ok, msg = oracle.is_same_matrix(oracle.Crs(A.nrows, B.ncols, rm_.astype(np.int64), ent_, val_.astype(np.float64)), oracle.spgemm(A, B))
            assert ok, msg
            if on:      # the (process-wide) knobs change under a live handle: numeric reuse rebuilds its indices and items for the new grid
                for w2, cap2 in ((256, 64), (4096, 6144)):
                    kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_block_w", w2))
                    kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_item_cap", cap2))
                    kk.spgemm_numeric(kh, Ad, False, Bd, False, Cm)
                    rm_, ent_, val_ = Cm.to_host()
                    ok, msg = oracle.is_same_matrix(oracle.Crs(A.nrows, B.ncols, rm_.astype(np.int64), ent_, val_.astype(np.float64)), oracle.spgemm(A, B))
                    assert ok, (w2, cap2, msg)
                kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_block_w", 1024))
                kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_item_cap", 100))
            kh.destroy_spgemm_handle()
    finally:
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_block", 1))
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_block_w", 16384))
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_items", 1))
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_item_cap", 6144))


def check_spgemm_val_steps(be):
    """Flat value kernel (vector walk: units of eight entries of one list, 16-byte loads) on rows of C whose windows hold several steps of
    products (lists that overlap heavily), in the lightest shape (128 work-items), the 512-work-item shape (more than 256 lists) and next to
    rows whose windows end inside the first step.  The last row of B is one of the long lists and nnz(B) is not a multiple of 4: the 16-byte
    walks of the symbolic phase (unit kernel) and of the bitmap kernel take the array's last, partial quad from their tail registers.
    (The scalar walks with 1 .. 3 steps in flight, knob spgemm_val_steps, measured neutral in round 5, are gone in round 6.)"""
    rng = np.random.default_rng(17)
    n, k = 400, 12000
    rows = [list(range(23)) + [n - 1],                    # 24 lists of ~6000 over 12000 columns: ~12 products per entry of C
            [0, 30, 31],                                  # one long list and two short ones
            list(range(20, 320)),                         # 300 lists: the 512-work-item shape
            [n - 1], [5], []]
    arm = np.zeros(len(rows) + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=arm[1:])
    A = oracle.Crs(len(rows), n, arm, np.concatenate([np.array(r, dtype=np.int32) for r in rows]), 1 + 49 * rng.random(arm[-1]))

    def make_b(last_len):
        lens = np.concatenate([np.full(23, 6000), np.full(n - 24, 150), [last_len]])
        rm = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
        ent = np.concatenate([np.sort(rng.choice(k, size=l, replace=False)) for l in lens]).astype(np.int32)
        return oracle.Crs(n, k, rm, ent, 1 + 49 * rng.random(rm[-1]))
    for last_len in (6001, 6007, 6002):
        B = make_b(last_len)
        assert B.nnz % 4 == last_len % 4 != 0
        got = check_spgemm(be, A, B)
        assert np.diff(got.row_map)[0] > 5461
        for bits in (10, 12):                             # the unit kernel's windows cut the last list: its last piece ends the arrays
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_unit_bits", bits))
            try: check_spgemm(be, A, B, reuse=False)
            finally: kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_unit_bits", 18))
    check_spgemm(be, A, B, offset_dtype=np.int64, value_dtype=np.float32)


def check_spgemm_sorted_emission(be):
    """Rows of C with more than 256 entries out of at most 2048 products: entries(C) sorted in LDS (`spgemm_emit_sort`, default) instead
    of walking the products through a 2^20-column bitmap.  A rows of more than 256 entries (two chunks of lists), duplicates among
    the products, rows above the product limit next to them (bitmap kernel), short rows (wave kernel); same C with the knob off."""
    rng = np.random.default_rng(23)
    nb, k = 3000, 50000
    lens = np.concatenate([np.full(1000, 3), np.full(2000, 20)])
    rm = np.zeros(nb + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = np.concatenate([np.sort(rng.choice(k, size=l, replace=False)) for l in lens]).astype(np.int32)
    B = oracle.Crs(nb, k, rm, ent, 1 + 49 * rng.random(rm[-1]))
    rows = []
    for i in range(40):
        rows.append(np.sort(rng.choice(np.arange(1000, 3000), size=int(rng.integers(20, 90)), replace=False)))    # 400 .. 1800 products
    rows.append(np.sort(rng.choice(1000, size=300, replace=False)))                                            # 300 lists of 3: two chunks
    rows.append(np.sort(rng.choice(np.arange(1000, 3000), size=102, replace=False)))                           # 2040 products: just inside
    rows.append(np.sort(rng.choice(np.arange(1000, 3000), size=104, replace=False)))                           # 2080: the bitmap kernel
    rows.append(np.sort(rng.choice(np.arange(1000, 3000), size=400, replace=False)))                           # 8000 products
    rows.append(np.array([5, 1500]))                                                                            # 23 products: wave kernel
    rows.append(np.array([], dtype=np.int64))
    rows.append(np.concatenate([np.full(1, 1200), np.sort(rng.choice(np.arange(1000, 3000), size=30, replace=False))]))   # (unsorted A row, maybe a duplicate column)
    arm = np.zeros(len(rows) + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=arm[1:])
    A = oracle.Crs(len(rows), nb, arm, np.concatenate(rows).astype(np.int32), 1 + 49 * rng.random(arm[-1]))
    gold = oracle.spgemm(A, B)
    sizes = np.diff(gold.row_map)
    assert (sizes[:42] > 256).all()
    for on in (1, 0):
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_emit_sort", on))
        try:
            for odt in (np.int32, np.int64):
                kh = kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
                Ad, Bd = dev(be, A, offset_dtype=odt), dev(be, B, offset_dtype=odt)
                Cm = kk.spgemm_symbolic(kh, Ad, False, Bd, False)
                kk.spgemm_numeric(kh, Ad, False, Bd, False, Cm)
                sh = kh.get_spgemm_handle()
                assert (sh.get(15) >= 42) == bool(on), sh.get(15)
                rm_, ent_, val_ = Cm.to_host()
                ok, msg = oracle.is_same_matrix(oracle.Crs(A.nrows, B.ncols, rm_.astype(np.int64), ent_, val_.astype(np.float64)), gold)
                assert ok, msg
                kk.spgemm_numeric(kh, Ad, False, Bd, False, Cm)            # reuse: entries kept
                assert sh.get(11) == 1
                rm2, ent2, val2 = Cm.to_host()
                assert np.array_equal(ent_, ent2) and np.allclose(val_, val2, rtol=1e-13)
                kh.destroy_spgemm_handle()
        finally:
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_emit_sort", 1))


def check_spgemm_quad_rows(be):
    """Wave-per-row kernels with four rows of the list per wave (`spgemm_quad_rows`): stencil products whose rows all have a few
    dozen products (16 lanes per row), lists whose length is no multiple of four, waves that mix small and larger rows (they fall back to
    one row after the other), A rows of more than 16 entries on 1-entry rows of B (two chunks per group), empty rows; same C with the knob off."""
    rng = np.random.default_rng(29)
    cases = []
    A7 = randomized(oracle.laplace3d("FD", 9, 7, 5)); cases.append((A7, A7))                      # 315 rows, 49 products at most
    A9 = randomized(oracle.laplace2d("FE", 13, 11)); cases.append((A9, A9))                       # 143 rows, 81 products: above the limit, wave path
    # restriction-like P (1 .. 3 entries per row) and A P: 7 .. 81 products per row, 3 .. 40 entries
    A27 = randomized(oracle.laplace3d("FE", 9, 7, 5))
    lens = rng.integers(1, 4, size=A27.ncols)
    rm = np.zeros(A27.ncols + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    P = oracle.Crs(A27.ncols, 60, rm, np.concatenate([np.sort(rng.choice(60, size=l, replace=False)) for l in lens]).astype(np.int32), 1 + 49 * rng.random(rm[-1]))
    cases.append((A27, P))
    # mixed: every fifth row of A is long (200 entries on 1-entry rows of B: 200 products, two chunks of 16 do not apply -- wave path), the others tiny;
    # and a row of 40 entries on 1-entry rows (40 products, three chunks of a group)
    nb = 500
    B1 = oracle.Crs(nb, 4000, np.arange(nb + 1, dtype=np.int64), rng.integers(0, 4000, size=nb).astype(np.int32), 1 + 49 * rng.random(nb))
    rows = []
    for i in range(41):
        if i % 5 == 4: rows.append(np.sort(rng.choice(nb, size=200, replace=False)))
        elif i == 7: rows.append(np.sort(rng.choice(nb, size=40, replace=False)))
        elif i == 11: rows.append(np.array([], dtype=np.int64))
        else: rows.append(np.sort(rng.choice(nb, size=int(rng.integers(1, 17)), replace=False)))
    arm = np.zeros(len(rows) + 1, dtype=np.int64); np.cumsum([len(r) for r in rows], out=arm[1:])
    Am = oracle.Crs(len(rows), nb, arm, np.concatenate(rows).astype(np.int32), 1 + 49 * rng.random(arm[-1]))
    cases.append((Am, B1))
    try:
        for on in (2, 1, 0):         # 2: always (waves with a larger row fall back), 1: only when every row of the product is small (default), 0: never
            kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_quad_rows", on))
            for A, B in cases:
                check_spgemm(be, A, B)
            check_spgemm(be, A7, A7, offset_dtype=np.int64, value_dtype=np.float32)
    finally:
        kk._capi.check(be.lib, be.lib.kkamd_set_default(b"spgemm_quad_rows", 1))


def randomized(A0, seed=5):
    """values re-drawn in [1,50) as the reference's SpGEMM tests do (Test_Sparse_spgemm.hpp:62-72)"""
    rng = np.random.default_rng(seed)
    return oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, 1 + 49 * rng.random(A0.nnz))


def hub_matrix(n, ncols, base_nnz, hubs, seed=0):
    """sorted random matrix with a few very long rows: rows i in `hubs` get hubs[i] entries"""
    rng = np.random.default_rng(seed)
    lens = np.full(n, base_nnz)
    for i, l in hubs.items():
        lens[i] = l
    rm = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = np.empty(rm[-1], dtype=np.int32)
    for i in range(n):
        ent[rm[i]:rm[i + 1]] = np.sort(rng.choice(ncols, size=lens[i], replace=False))
    return oracle.Crs(n, ncols, rm, ent, 1 + 49 * rng.random(rm[-1]))


# --- spmv_struct: the reference's own cases (sparse/unit_test/Test_Sparse_spmv.hpp:609-768, 1096-1104) -----------------
STRUCT_CASES_1D = [(10,)]
STRUCT_CASES_2D = [(25, 21), (20, 25), (22, 22)]
STRUCT_CASES_3D = [(20, 20, 20), (22, 22, 22), (25, 10, 20), (10, 20, 25), (10, 24, 20)]


def struct_matrix(dims, stencil_type):
    if len(dims) == 1:
        return oracle.laplace1d(dims[0])
    name = "FD" if stencil_type == 1 else "FE"
    return oracle.laplace2d(name, *dims) if len(dims) == 2 else oracle.laplace3d(name, *dims)


def check_spmv_struct(be, dims, stencil_type, mode="N", offset_dtype=np.int32, value_dtype=None, vec_dtype=np.float64,
                      seed=13718, A0=None, rank2=False):
    """check_spmv_struct of the reference test (:263-296): alpha/beta in (1,0), (0,1), (1,1) against the oracle's
    restatement of the host path (itself checked against sequential_spmv in test_oracle.py)."""
    A0 = A0 if A0 is not None else struct_matrix(dims, stencil_type)
    rng = np.random.default_rng(seed)
    trans = mode in "TH"
    nin, nout = (A0.nrows, A0.ncols) if trans else (A0.ncols, A0.nrows)
    x = rng.random(nin).astype(vec_dtype); y0 = rng.random(nout).astype(vec_dtype)
    A = dev(be, A0, offset_dtype, value_dtype)
    Ao = A0 if value_dtype is None else oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries,
                                                   A0.values.astype(value_dtype).astype(np.float64))
    max_val = float(np.abs(A0.values).max())
    for alpha, beta in ((1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (-2.5, 0.5)):
        xd, yd = be.from_numpy(x.reshape(-1, 1) if rank2 else x), be.from_numpy(y0.reshape(-1, 1) if rank2 else y0)
        kk.spmv_struct(mode, stencil_type, dims, alpha, A, xd, beta, yd)
        got = be.to_numpy(yd).astype(np.float64).reshape(-1)
        exp = oracle.spmv_struct(mode, stencil_type, dims, Ao, alpha, x.astype(np.float64), beta, y0.astype(np.float64).copy())
        eps_scale = 1.0 if vec_dtype == np.float64 else EPS_F / np.finfo(np.float64).eps
        tol = oracle.spmv_max_error(A0, alpha, beta, max_val=max_val) * eps_scale
        ok, err = fspmv_ok(exp, got, max(tol, 1e-300))
        assert ok, "spmv_struct mismatch dims=%s stencil=%d mode=%s alpha=%g beta=%g: max err %g > tol %g" % (
            dims, stencil_type, mode, alpha, beta, err, tol)
    if not trans:
        # beta == 0 overwrites y: the incoming y (here all NaN) is never read (BLAS convention of the CRS kernels)
        xd = be.from_numpy(x.reshape(-1, 1) if rank2 else x)
        yd = be.from_numpy(np.full(y0.reshape(-1, 1).shape if rank2 else y0.shape, np.nan, dtype=vec_dtype))
        kk.spmv_struct(mode, stencil_type, dims, 1.0, A, xd, 0.0, yd)
        got = be.to_numpy(yd).astype(np.float64).reshape(-1)
        exp = oracle.spmv_struct(mode, stencil_type, dims, Ao, 1.0, x.astype(np.float64), 0.0, np.zeros(nout))
        ok, err = fspmv_ok(exp, got, max(oracle.spmv_max_error(A0, 1.0, 0.0, max_val=max_val) * eps_scale, 1e-300))
        assert ok and np.isfinite(got).all(), "spmv_struct beta=0 must not read y (dims=%s stencil=%d)" % (dims, stencil_type)


def window_code_cases():
    """(name, matrix, window codes expected) for stream_variant 6: column sets that need 1, several, exactly 16 and more
    than 16 windows of 4096 columns per tile, plus a ragged last tile."""
    rng = np.random.default_rng(77)
    out = []
    # multi-diagonal: offsets spread over 7 windows, columns clipped to the matrix
    n, nc = 5000, 400000
    offs = np.array([0, 1, 2, 5000, 5001, 60000, 60007, 130000, 200000, 200001, 300000, 390000])
    cols = (np.arange(n)[:, None] * 2 + offs[None, :])
    keep = cols < nc
    lens = keep.sum(1)
    rm = np.zeros(n + 1, np.int64); rm[1:] = np.cumsum(lens)
    out.append(("diagonals", oracle.Crs(n, nc, rm, cols[keep].astype(np.int32), rng.random(int(rm[-1]))), 1))
    # exactly 16 column clusters per row (16 windows needed), then 17 (falls back to plain entries)
    for k, ok in ((16, 1), (17, 0)):
        n = 1500
        cols = (np.arange(k)[None, :] * 20000 + rng.integers(0, 1000, (n, k))).astype(np.int32)
        cols.sort(axis=1)
        rm = np.arange(n + 1, dtype=np.int64) * k
        out.append(("clusters%d" % k, oracle.Crs(n, 20000 * k, rm, cols.reshape(-1), rng.random(n * k)), ok))
    # uniformly random columns over a wide range: no tile is coverable
    out.append(("random-wide", oracle.random_crs(3000, 3000000, 9, variance=3, seed=5), 0))
    # 27-pt stencil (9 x-lines per tile) with nnz not a multiple of any tile size
    out.append(("27pt", oracle.laplace3d("FE", 24, 19, 13), 1))
    return out


def check_pattern_direct(be, cases=None):
    """Round 6: the row-pattern records straight from the matrix (pat_direct_kernel: rows compared, the window cover over <= 256 column
    intervals per tile) against the records by way of the window codes (pattern_direct 0): the same tiles get a record, y is the same bit
    for bit (same records, same order of the sums), the one tile without a record reads entries instead of codes; 32- and 64-bit offsets,
    both tile sizes, fp32 values.  Matrices with more than one tile in a hundred without a record take the codes either way."""
    took = 0
    for name, A0, npt, _ in (cases or pattern_code_cases()):
        for odt, vdt in ((np.int32, None), (np.int64, np.float32)):
            res = {}
            for direct in (1, 0):
                kn = {"window_codes_min_knnz": 0, "nnz_per_thread": npt, "pattern_codes": 2, "pattern_codes_min_knnz": 0, "pattern_direct": direct}
                rng = np.random.default_rng(5)
                x, y0 = rng.random(A0.ncols), rng.random(A0.nrows)
                A = dev(be, A0, odt, vdt)
                xd, yd = be.from_numpy(x), be.from_numpy(y0)
                h = kk.SPMVHandle("SPMV_DEFAULT")
                for k_, v_ in kn.items(): h.set(k_, v_)
                kk.spmv(h, "N", 1.5, A, xd, 0.5, yd)
                res[direct] = (be.to_numpy(yd).copy(), h.query("pattern_direct"), h.query("pattern_tiles"), h.query("plain_tiles"), h.query("code_tiles"), h.query("tiles"))
                check_spmv(be, A0, "N", 1.5, 0.0, "SPMV_DEFAULT", knobs=kn, max_val=32.0, nans=True, offset_dtype=odt, value_dtype=vdt)
            (y1, d1, p1, pl1, c1, n1), (y0_, d0, p0, pl0, c0, _) = res[1], res[0]
            assert d0 == 0 and np.array_equal(y1, y0_), (name, d1, p1, p0)
            if d1:
                took += 1
                assert p1 == p0 and c1 == 0 and pl1 == n1 - p1 and pl1 <= max(1, n1 // 100), (name, p1, p0, pl1, c1, n1)
            else:
                assert (p1, pl1, c1) == (p0, pl0, c0), (name, res[1][1:], res[0][1:])
    assert took >= 4, took


def pattern_code_cases():
    """(name, matrix, nnz_per_thread, row-pattern records expected) for the staged SpMV kernel's row-pattern codes."""
    out = []
    out.append(("27pt", oracle.laplace3d("FE", 210, 9, 5), 16, True))
    out.append(("27pt-2048", oracle.laplace3d("FE", 120, 11, 5), 8, True))
    out.append(("7pt", oracle.laplace3d("FD", 400, 8, 5), 8, True))
    out.append(("5pt", oracle.laplace2d("FD", 500, 37), 8, True))
    out.append(("3pt", oracle.laplace1d(12000), 16, True))
    # a stencil with a handful of rows carrying one extra explicit zero: more segments in the tiles they fall into
    A0 = oracle.laplace2d("FE", 200, 60)
    rm = A0.row_map.copy(); ent = A0.entries; val = A0.values
    for r in (1234, 1235, 5000, 9000, 9001, 9002, 9003, 9004, 9005, 9006, 9007):       # eight in a row: that tile keeps its codes
        ent = np.insert(ent, rm[r + 1], ent[rm[r + 1] - 1]); val = np.insert(val, rm[r + 1], 0.0); rm[r + 1:] += 1
    out.append(("9pt-perturbed", oracle.Crs(A0.nrows, A0.ncols, rm, ent.astype(np.int32), val), 16, True))
    # Toeplitz band of 40 diagonals (rows longer than a record's table) and a stencil with empty rows: the codes stay
    n = 3000; offs = np.arange(-20, 20)
    cols = np.arange(n)[:, None] + offs[None, :]; keep = (cols >= 0) & (cols < n)
    rm = np.zeros(n + 1, np.int64); rm[1:] = np.cumsum(keep.sum(1))
    out.append(("band40", oracle.Crs(n, n, rm, cols[keep].astype(np.int32), np.random.default_rng(1).random(int(rm[-1]))), 16, False))
    A0 = oracle.laplace2d("FD", 400, 30)
    lens = np.diff(A0.row_map).copy(); keep = np.ones(A0.nnz, bool)
    for r in range(100, A0.nrows, 3000): keep[A0.row_map[r]:A0.row_map[r + 1]] = False; lens[r] = 0
    rm = np.zeros(A0.nrows + 1, np.int64); rm[1:] = np.cumsum(lens)
    out.append(("5pt-empty-rows", oracle.Crs(A0.nrows, A0.ncols, rm, A0.entries[keep], A0.values[keep]), 16, True))
    # banded random columns: staged x, but no two rows share a pattern
    out.append(("banded-random", oracle.random_crs(6000, 6000, 9, variance=0, seed=4, bandwidth=300, sorted_rows=True), 16, False))
    return out


def mixed_tile_cases():
    """(name, matrix) where only SOME tiles of the planned kernel can use the column codes: a stencil with a few rows that
    couple to columns all over the matrix (the rest of the tiles keep codes / staged x / records), and the reverse."""
    rng = np.random.default_rng(21)
    out = []
    A0 = oracle.laplace3d("FE", 40, 30, 12)
    rm = A0.row_map.copy(); ent = A0.entries.copy(); val = A0.values.copy()
    wide = 400000                                            # x is longer than the stencil's grid: the coupling rows reach all over it
    for r in (777, 5000, 5001, 11000):                       # four rows get 40 extra far-apart columns each
        extra = (A0.ncols + np.sort(rng.choice(wide - A0.ncols, size=40, replace=False))).astype(np.int32)
        ent = np.insert(ent, rm[r + 1], extra); val = np.insert(val, rm[r + 1], rng.random(40)); rm[r + 1:] += 40
    out.append(("stencil+dense-rows", oracle.Crs(A0.nrows, wide, rm, ent, val)))
    # first half random wide (no tile coverable), second half a banded block (every tile coverable)
    n = 6000
    top = oracle.random_crs(n // 2, 2000000, 12, variance=0, seed=3)
    band = oracle.random_crs(n // 2, n // 2, 12, variance=0, seed=4, bandwidth=200, sorted_rows=True)       # columns inside [0, 3000)
    rm = np.concatenate([top.row_map, band.row_map[1:] + top.row_map[-1]])
    out.append(("random-then-banded", oracle.Crs(n, 2000000, rm, np.concatenate([top.entries, band.entries]), np.concatenate([top.values, band.values]))))
    return out


def mv4_cases():
    """(name, matrix, rows the plane-marching rank-2 kernel must leave to the gather kernel: None = any, 0 = none) -- lattice
    stencils whose patches, lines and k-chunks are ragged (nx % 32, ny % 4 != 0), boundary rows of a truncated stencil, rows
    that break the pattern"""
    out = [("27pt 40x14x14", oracle.laplace3d("FE", 40, 14, 14), None), ("7pt 35x13x14", oracle.laplace3d("FD", 35, 13, 14), None),
           ("27pt 9x33x17", oracle.laplace3d("FE", 9, 33, 17), None)]
    # clean truncated stencils (every boundary row is a subset whose missing entries point outside): no gather rows.  7 points =
    # the kernel's compiled-in pattern; 19 points (faces + edges) and 11 points = its run-time patterns with 9 and 5 groups
    def truncated(nx, ny, nz, steps, seed):
        i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        rows, cols = [], []
        for dk, dj, di in steps:
            ok = (i + di >= 0) & (i + di < nx) & (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
            rows.append((k * ny * nx + j * nx + i)[ok]); cols.append(((k + dk) * ny * nx + (j + dj) * nx + i + di)[ok])
        rows = np.concatenate(rows); cols = np.concatenate(cols)
        order = np.lexsort((cols, rows)); rows = rows[order]; cols = cols[order]
        n = nx * ny * nz
        rm = np.zeros(n + 1, dtype=np.int64); np.add.at(rm, rows + 1, 1); rm = np.cumsum(rm)
        return oracle.Crs(n, n, rm, cols.astype(np.int32), np.random.default_rng(seed).random(rows.size) + 0.5)
    seven = ((-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 1, 0), (1, 0, 0))
    nx, ny, nz = 33, 6, 21
    clean = truncated(nx, ny, nz, seven, 12)
    rm, cols, vals, n = clean.row_map, clean.entries, clean.values, clean.nrows
    out.append(("7pt clean 33x6x21", clean, 0))
    nineteen = tuple((dk, dj, di) for dk in (-1, 0, 1) for dj in (-1, 0, 1) for di in (-1, 0, 1) if abs(dk) + abs(dj) + abs(di) <= 2)
    out.append(("19pt clean 20x15x16", truncated(20, 15, 16, nineteen, 13), 0))
    eleven = seven + ((0, -1, -1), (0, 1, 1), (0, -1, 1), (0, 1, -1))
    out.append(("11pt clean 36x9x14", truncated(36, 9, 14, eleven, 14), 0))
    # periodic in i (wrap-around couplings on two faces: those rows cannot conform) and a lattice whose plane stride is a
    # multiple of its line stride only by accident of the sizes (nx = ny)
    per = seven + ((0, 0, 0),)
    def periodic_i(nx, ny, nz, seed):
        i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        rows, cols = [], []
        for dk, dj, di in seven:
            ii = (i + di) % nx
            ok = (j + dj >= 0) & (j + dj < ny) & (k + dk >= 0) & (k + dk < nz)
            rows.append((k * ny * nx + j * nx + i)[ok]); cols.append(((k + dk) * ny * nx + (j + dj) * nx + ii)[ok])
        rows = np.concatenate(rows); cols = np.concatenate(cols)
        order = np.lexsort((cols, rows)); rows = rows[order]; cols = cols[order]
        n = nx * ny * nz
        rm_ = np.zeros(n + 1, dtype=np.int64); np.add.at(rm_, rows + 1, 1); rm_ = np.cumsum(rm_)
        return oracle.Crs(n, n, rm_, cols.astype(np.int32), np.random.default_rng(seed).random(rows.size) + 0.5)
    out.append(("7pt periodic in i 34x7x20", periodic_i(34, 7, 20, 15), 2 * 7 * 20))
    out.append(("7pt clean 17x17x17", truncated(17, 17, 17, seven, 16), 0))
    # the same with rows that break the pattern: an extra coupling, a dropped interior entry, an emptied row, a wrap-around entry
    rm2 = rm.copy(); ent = cols.astype(np.int32).copy(); val = vals.copy()
    def drop(r, pos):
        nonlocal rm2, ent, val
        at = rm2[r] + pos
        ent = np.delete(ent, at); val = np.delete(val, at); rm2[r + 1:] -= 1
    def add(r, col):
        nonlocal rm2, ent, val
        seg = ent[rm2[r]:rm2[r + 1]]
        at = rm2[r] + int(np.searchsorted(seg, col))
        ent = np.insert(ent, at, col); val = np.insert(val, at, 0.25); rm2[r + 1:] += 1
    mid = (10 * ny + 3) * nx + 17
    drop(mid, 2); add(mid + 1, mid + 9); add(mid + 2 * nx, 5)
    for _ in range(int(rm2[mid + 40 + 1] - rm2[mid + 40])):
        drop(mid + 40, 0)
    out.append(("7pt + broken rows", oracle.Crs(n, n, rm2, ent, val), 4))
    return out


def mv4_duplicate_cases():
    """(name, matrix, plane-marching plan expected, rows it must leave to the gather kernel) -- lattice stencils whose rows store
    a column twice (the reference adds duplicates up: spmv_impl.hpp:240-303 walks entries, not columns).  A pattern row with a
    duplicate would put two values into one slot of the value buffer: such rows are never the pattern and never conform."""
    base = [c for c in mv4_cases() if c[0] == "7pt clean 33x6x21"][0][1]
    def dup_diag(rows_with_dup):
        rm = base.row_map.astype(np.int64); ent = base.entries; val = base.values
        r_of = np.repeat(np.arange(base.nrows), np.diff(rm))
        diag_at = np.flatnonzero(ent == r_of)
        pick = diag_at[rows_with_dup]
        ent2 = np.insert(ent, pick, ent[pick]); val2 = np.insert(val, pick, 0.375 + 0.001 * np.arange(pick.size))
        add = np.zeros(base.nrows + 1, dtype=np.int64); add[np.asarray(rows_with_dup) + 1] = 1
        return oracle.Crs(base.nrows, base.ncols, rm + np.cumsum(add), ent2.astype(np.int32), val2)
    some = np.arange(7, base.nrows, 83)
    return [("7pt, diagonal stored twice in every row", dup_diag(np.arange(base.nrows)), False, 0),
            ("7pt, diagonal stored twice in %d rows" % some.size, dup_diag(some), True, int(some.size))]


def _from_dense_mask(mask, seed, ncols=None):
    """CRS matrix with an entry wherever mask is set (rows ascending), random values in [0.5, 1.5)"""
    n = mask.shape[0]
    rows, cols = np.nonzero(mask)
    rm = np.zeros(n + 1, dtype=np.int64); np.add.at(rm, rows + 1, 1); rm = np.cumsum(rm)
    return oracle.Crs(n, ncols or mask.shape[1], rm, cols.astype(np.int32), np.random.default_rng(seed).random(rows.size) + 0.5)


def block_diagonal(nblocks, bs, seed=0):
    n = nblocks * bs
    rm = np.arange(n + 1, dtype=np.int64) * bs
    ent = (np.repeat(np.arange(nblocks), bs * bs) * bs + np.tile(np.arange(bs), n)).astype(np.int32)
    return oracle.Crs(n, n, rm, ent, np.random.default_rng(seed).random(n * bs) + 0.5)


def multi_dof(A0, ndof, seed=0):
    """every entry of A0 becomes a dense ndof x ndof block (a vector-valued finite-element matrix on A0's mesh)"""
    n = A0.nrows
    lens = np.diff(A0.row_map)
    rm = np.zeros(n * ndof + 1, dtype=np.int64)
    rm[1:] = np.cumsum(np.repeat(lens * ndof, ndof))
    ent = np.empty(int(rm[-1]), dtype=np.int32)
    for r in range(n):
        c = (A0.entries[A0.row_map[r]:A0.row_map[r + 1]].astype(np.int64)[:, None] * ndof + np.arange(ndof)[None, :]).reshape(-1)
        for d in range(ndof):
            ent[rm[r * ndof + d]:rm[r * ndof + d + 1]] = c
    return oracle.Crs(n * ndof, A0.ncols * ndof, rm, ent, np.random.default_rng(seed).random(ent.size) + 0.5)


def mv5_cases():
    """(name, matrix, 16-row tiles the matrix-core rank-2 kernel must describe: None = some, 0 = the kernel must not engage, rows
    it must leave to its gather rows: None = any) under the default thresholds"""
    out = [("block diagonal 32x32", block_diagonal(40, 32, 1), 80, 0),
           ("block diagonal 5x5, 403 rows", _crop_rows(block_diagonal(81, 5, 2), 403), 25, 3),
           ("3 dof on 27-pt 6x5x4", multi_dof(oracle.laplace3d("FE", 6, 5, 4), 3, 3), None, None)]
    n = 500                                                                      # a dense band of 21 diagonals
    ii = np.arange(n)
    out.append(("band of 21 diagonals", _from_dense_mask(np.abs(ii[:, None] - ii[None, :]) <= 10, 4), 31, 4))
    # described tiles first, then rows with columns all over a wide matrix (their tiles fall below the fill threshold)
    top = block_diagonal(20, 16, 5)
    wide = oracle.random_crs(160, 200000, 12, variance=0, seed=6, sorted_rows=True)
    rm = np.concatenate([top.row_map, wide.row_map[1:] + top.row_map[-1]])
    out.append(("blocks then scattered rows", oracle.Crs(480, 200000, rm, np.concatenate([top.entries, wide.entries]), np.concatenate([top.values, wide.values])), 20, 160))
    # tiles the plan cannot describe between tiles it can: a row that descends, a column stored twice, a tile of more than 2048
    # entries, a tile without entries; and a described tile with empty rows inside
    B = block_diagonal(12, 16, 7)
    rm = B.row_map.copy(); ent = B.entries.copy(); val = B.values.copy()
    ent[rm[17]:rm[18]] = ent[rm[17]:rm[18]][::-1]                                 # tile 1: row 17 descends
    ent[rm[35] + 3] = ent[rm[35] + 2]                                             # tile 2: row 35 stores a column twice
    keep = np.ones(ent.size, bool); keep[rm[48]:rm[64]] = False                   # tile 3: no entries at all
    keep[rm[83]:rm[85]] = False                                                   # tile 5: two empty rows inside a described tile
    lens = np.diff(rm); lens[48:64] = 0; lens[83:85] = 0
    rm2 = np.zeros(B.nrows + 1, np.int64); rm2[1:] = np.cumsum(lens)
    out.append(("tiles that cannot be described", oracle.Crs(B.nrows, B.ncols, rm2, ent[keep], val[keep]), 9, 48))
    big = _from_dense_mask(np.ones((32, 150), bool), 8)                           # 16 x 150 = 2400 entries per tile
    out.append(("tiles above 2048 entries", big, 0, None))
    out.append(("uniform random", oracle.random_crs(800, 800, 9, variance=3, seed=5), 0, None))
    return out


def _crop_rows(A0, n):
    e = int(A0.row_map[n])
    return oracle.Crs(n, A0.ncols, A0.row_map[:n + 1].copy(), A0.entries[:e].copy(), A0.values[:e].copy())


def mv3_cases():
    """(name, matrix, staged tiles expected) for the LDS-staged rank-2 kernel"""
    out = [("27pt", oracle.laplace3d("FE", 37, 11, 9), True), ("7pt", oracle.laplace3d("FD", 50, 12, 7), True),
           ("9pt", oracle.laplace2d("FE", 130, 41), True),
           ("banded-random", oracle.random_crs(3000, 3000, 9, variance=4, seed=4, bandwidth=60, sorted_rows=True), True),
           ("random-wide", oracle.random_crs(2000, 50000, 9, variance=3, seed=5), False)]
    name, A0 = mixed_tile_cases()[0]
    out.append((name, A0, True))
    # rows longer than a staged tile may hold, between stencil rows
    A0 = oracle.laplace2d("FD", 90, 40)
    rm = A0.row_map.copy(); ent = A0.entries.copy(); val = A0.values.copy()
    rng = np.random.default_rng(8)
    for r in (500, 2000):
        extra = np.sort(rng.choice(A0.ncols, size=2500, replace=False)).astype(np.int32)
        ent = np.insert(ent, rm[r + 1], extra); val = np.insert(val, rm[r + 1], rng.random(2500)); rm[r + 1:] += 2500
    out.append(("5pt+long-rows", oracle.Crs(A0.nrows, A0.ncols, rm, ent, val), True))
    return out
