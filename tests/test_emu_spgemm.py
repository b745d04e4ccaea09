"""Kernel-LOGIC tests of the SpGEMM / scan / sort / generator kernels under the SIMT emulator (no GPU)."""
import glob
import os

import numpy as np
import pytest

import oracle
import parity_cases as pc
from emu import emu_backend

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def be():
    return emu_backend.backend()


def test_exclusive_scan(be):
    rng = np.random.default_rng(0)
    for n in (1, 2, 255, 256, 257, 2048, 2049, 5000, 2048 * 3 + 17):
        for dt, code in ((np.int32, 0), (np.int64, 1)):
            a = rng.integers(0, 50, size=n).astype(dt)
            ref = np.concatenate([[0], np.cumsum(a)[:-1]]).astype(dt)
            d = a.copy()
            pc.kk._capi.check(be.lib, be.lib.kkamd_exclusive_scan(d.ctypes.data, n, code, None))
            assert np.array_equal(d, ref)
    big = np.ones(2048 * 2048 + 5, dtype=np.int64)     # three-level scan
    pc.kk._capi.check(be.lib, be.lib.kkamd_exclusive_scan(big.ctypes.data, len(big), 1, None))
    assert np.array_equal(big, np.arange(len(big)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "structured_*bc1.npz"))))
def test_device_generators_match_reference(be, path):
    g = np.load(path)
    kind, stencil, dims, _ = os.path.basename(path)[len("structured_"):-4].split("_")
    dims = [int(v) for v in dims.split("x")]
    for odt in (np.int32, np.int64):
        A = pc.kk.laplace_matrix(stencil.upper(), *dims, offset_dtype=odt, backend=be)
        rm, ent, val = A.to_host()
        assert np.array_equal(rm, g["row_map"]) and np.array_equal(ent, g["entries"]) and np.array_equal(val, g["values"])


def test_sort_crs(be):
    A0 = oracle.random_crs(300, 5000, 40, variance=39, seed=5)           # unsorted, with duplicate columns
    long_row = oracle.random_crs(3, 100000, 3000, variance=2500, seed=6)
    for M in (A0, long_row):
        A = pc.dev(be, M)
        pc.kk.sort_crs_matrix(A)
        gold = oracle.Crs(M.nrows, M.ncols, M.row_map, M.entries.copy(), M.values.copy())
        oracle.sort_crs(gold)
        rm, ent, val = A.to_host()
        assert np.array_equal(ent, gold.entries) and np.array_equal(val, gold.values)     # stable like the oracle


def test_sort_long_rows_merge_transpose(be):
    # rows longer than one 8192-entry LDS segment: chunk sort + merge passes (1, 2 and 3 passes), with duplicate columns
    rng = np.random.default_rng(8)
    lens = [20000, 3, 8193, 0, 70000, 8192, 16384]
    rm = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = rng.integers(0, 30000, size=rm[-1]).astype(np.int32)
    M = oracle.Crs(len(lens), 30000, rm, ent, 1 + 49 * rng.random(rm[-1]))
    for odt in (np.int32, np.int64):
        A = pc.dev(be, M, odt)
        pc.kk.sort_crs_matrix(A)
        gold = oracle.Crs(M.nrows, M.ncols, M.row_map, M.entries.copy(), M.values.copy()); oracle.sort_crs(gold)
        _, e, v = A.to_host()
        assert np.array_equal(e, gold.entries) and np.array_equal(v, gold.values)          # stable like the oracle
    # sort_and_merge_matrix: duplicates summed left to right
    A = pc.dev(be, M)
    Cm = pc.kk.sort_and_merge_matrix(A)
    gm = oracle.sort_and_merge(oracle.Crs(M.nrows, M.ncols, M.row_map, M.entries.copy(), M.values.copy()))
    r, e, v = Cm.to_host()
    assert np.array_equal(r, gm.row_map) and np.array_equal(e, gm.entries) and np.allclose(v, gm.values, rtol=1e-13, atol=0)
    small = pc.randomized(oracle.random_crs(50, 40, 6, seed=3))                              # short rows with duplicates, graph-only too
    gs = oracle.sort_and_merge(oracle.Crs(small.nrows, small.ncols, small.row_map, small.entries.copy(), small.values.copy()))
    r, e, v = pc.kk.sort_and_merge_matrix(pc.dev(be, small, np.int64)).to_host()
    assert np.array_equal(r, gs.row_map) and np.array_equal(e, gs.entries) and np.allclose(v, gs.values, rtol=1e-13, atol=0)
    # transpose_matrix against the oracle's transpose (no duplicates -> deterministic), including a hub column
    T0 = pc.hub_matrix(300, 200, 5, {7: 150}, seed=4)
    At = pc.kk.transpose_matrix(pc.dev(be, T0))
    gt = oracle.transpose(T0)
    r, e, v = At.to_host()
    assert np.array_equal(r, gt.row_map) and np.array_equal(e, gt.entries) and np.array_equal(v, gt.values)
    big = oracle.Crs(1, 30000, np.array([0, 9000]), np.sort(rng.choice(30000, 9000, replace=False)).astype(np.int32), rng.random(9000))
    Bt = pc.kk.transpose_matrix(pc.kk.transpose_matrix(pc.dev(be, big)))                       # (A^T)^T == A, long row on the way back
    r, e, v = Bt.to_host()
    assert np.array_equal(e, big.entries) and np.array_equal(v, big.values)


@pytest.mark.parametrize("m,n,k,nnzA,nnzB", [
    (0, 0, 0, 0, 0), (0, 12, 5, 0, 20), (10, 10, 0, 20, 0), (10, 0, 10, 0, 0),
    (10, 10, 10, 0, 0), (10, 10, 10, 20, 0), (10, 10, 10, 0, 20)])
def test_spgemm_degenerate(be, m, n, k, nnzA, nnzB):
    # the seven empties of sparse/unit_test/Test_Sparse_spgemm.hpp:491-504
    A0 = pc.randomized(oracle.random_crs(m, n, nnzA // m if m else 0, seed=1, sorted_rows=True))
    B0 = pc.randomized(oracle.random_crs(n, k, nnzB // n if n else 0, seed=2, sorted_rows=True))
    got = pc.check_spgemm(be, A0, B0)
    assert got.nnz == 0


@pytest.mark.parametrize("odt", [np.int32, np.int64])
def test_spgemm_random_shapes(be, odt):
    # scaled-down versions of Test_Sparse_spgemm.hpp:485-490 (10000x8000x6000 @160k, 1000x500x1600)
    A0 = pc.randomized(oracle.random_crs(400, 320, 16, variance=10, seed=3, sorted_rows=True))
    B0 = pc.randomized(oracle.random_crs(320, 240, 20, variance=12, seed=4, sorted_rows=True))
    pc.check_spgemm(be, A0, B0, offset_dtype=odt)
    A1 = pc.randomized(oracle.random_crs(100, 50, 8, seed=5))            # unsorted inputs with duplicates are legal
    B1 = pc.randomized(oracle.random_crs(50, 160, 30, seed=6))
    pc.check_spgemm(be, A1, B1, offset_dtype=odt)


def test_spgemm_float(be):
    A0 = pc.randomized(oracle.random_crs(120, 100, 9, variance=4, seed=13, sorted_rows=True))
    pc.check_spgemm(be, A0, pc.randomized(oracle.random_crs(100, 90, 7, seed=14, sorted_rows=True)), value_dtype=np.float32)


def test_spgemm_all_bins(be):
    # rows landing in every launch shape of both phases: flops 0 / <=1365 / <=2048 / <=16384 / dense,
    # and nnz(C row) <=256 / <=2048 / <=5461 / dense
    B0 = pc.hub_matrix(64, 30000, 40, {0: 9000, 1: 3000, 2: 600, 3: 120, 5: 20000}, seed=1)
    lens = {0: 1, 1: 1, 2: 1, 3: 2, 4: 0, 5: 1, 6: 3, 7: 30}
    rng = np.random.default_rng(2)
    rows = []
    cols_for = {0: [0], 1: [1], 2: [2], 3: [3, 7], 4: [], 5: [5], 6: [0, 1, 5], 7: list(range(6, 36))}
    rm = [0]; ent = []
    for i in range(8):
        ent += cols_for[i]; rm.append(len(ent))
    A0 = oracle.Crs(8, 64, np.array(rm), np.array(ent, dtype=np.int32), 1 + 49 * rng.random(len(ent)))
    got = pc.check_spgemm(be, A0, B0)
    sizes = np.diff(got.row_map)
    assert sizes[4] == 0 and sizes[3] <= 256 and 256 < sizes[2] <= 2048 and 2048 < sizes[1] <= 5461 and sizes[0] > 5461
    assert sizes[6] > 20000        # union of three hub rows: dense path with overlapping columns


def _set(be, key, value):
    pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(key.encode(), int(value)))


def test_spgemm_dense_row_windows(be):
    # dense rows with (a) the column bitmap cut into several LDS windows, (b) small value windows so the per-entry
    # cursors are exercised across many passes, (c) the HBM-accumulator fallback taken when B is not sorted
    B0 = pc.hub_matrix(48, 26000, 25, {0: 9000, 1: 7000, 2: 12000, 3: 300}, seed=11)
    rm = [0, 3, 4, 8, 8, 11]
    ent = np.array([0, 1, 2,   2,   0, 3, 5, 9,   1, 2, 30], dtype=np.int32)
    rng = np.random.default_rng(3)
    A0 = oracle.Crs(5, 48, np.array(rm), ent, 1 + 49 * rng.random(len(ent)))
    # a row of A longer than the LDS cursor cache (512 entries): the HBM cursors carry the rest across windows
    B1 = pc.randomized(oracle.random_crs(700, 20000, 12, variance=6, seed=21, sorted_rows=True))
    cols = np.sort(rng.choice(700, size=650, replace=False)).astype(np.int32)
    A1 = oracle.Crs(2, 700, np.array([0, 650, 653]), np.concatenate([cols, [1, 5, 9]]).astype(np.int32), 1 + 49 * rng.random(653))
    # a row of A above the row-flops pass's workgroup-per-row threshold (2048 entries), next to short ones
    B2 = pc.randomized(oracle.random_crs(3000, 5000, 3, variance=2, seed=22, sorted_rows=True))
    cols2 = np.sort(rng.choice(3000, size=2500, replace=False)).astype(np.int32)
    A2 = oracle.Crs(3, 3000, np.array([0, 2, 2502, 2505]), np.concatenate([[4, 7], cols2, [1, 5, 9]]).astype(np.int32), 1 + 49 * rng.random(2505))
    pc.check_spgemm(be, A2, B2)
    pc.check_spgemm(be, A2, B2, offset_dtype=np.int64)
    try:
        _set(be, "spgemm_emit_win_bits", 4096)               # narrower windows for the numeric emission of rows whose bitmap was not kept
        pc.check_spgemm(be, A0, B0)
        _set(be, "spgemm_emit_win_bits", 0)
        _set(be, "spgemm_win_bits", 4096); _set(be, "spgemm_val_cap", 192)
        got = pc.check_spgemm(be, A0, B0)
        assert np.diff(got.row_map).max() > 12000
        got = pc.check_spgemm(be, A1, B1)
        assert np.diff(got.row_map)[0] > 2048
        _set(be, "spgemm_win_bits", 1 << 20); _set(be, "spgemm_val_cap", 2048)
        pc.check_spgemm(be, A0, B0, offset_dtype=np.int64, value_dtype=np.float32)
        _set(be, "spgemm_force_unsorted", 1)
        pc.check_spgemm(be, A0, B0)
        _set(be, "spgemm_force_unsorted", 0); _set(be, "spgemm_emit_chunked", 1)      # contiguous-run bitmap walk for every row
        pc.check_spgemm(be, A0, B0)
        _set(be, "spgemm_win_bits", 4096)
        pc.check_spgemm(be, A1, B1)
    finally:
        _set(be, "spgemm_win_bits", 1 << 20); _set(be, "spgemm_val_cap", 2048); _set(be, "spgemm_force_unsorted", 0)
        _set(be, "spgemm_emit_chunked", 0); _set(be, "spgemm_emit_win_bits", 0)
    ent, val = B0.entries.copy(), B0.values.copy()          # unsorted B: detected by the symbolic phase, dense rows fall back
    for i in range(B0.nrows):
        lo, hi = B0.row_map[i], B0.row_map[i + 1]
        q = rng.permutation(hi - lo)
        ent[lo:hi], val[lo:hi] = ent[lo:hi][q], val[lo:hi][q]
    Bu = oracle.Crs(B0.nrows, B0.ncols, B0.row_map, ent, val)
    pc.check_spgemm(be, A0, Bu)
    with pytest.raises(pc.kk.KkamdError):
        _set(be, "spgemm_win_bits", 100)


def test_spgemm_structure_kept_by_the_symbolic_phase(be):
    pc.check_spgemm_kept_structure(be)


def test_spgemm_symbolic_by_units(be):
    pc.check_spgemm_units(be, light=True)


def test_spgemm_galerkin_products(be):
    pc.check_spgemm_galerkin(be)


def test_spgemm_column_block_value_kernel(be):
    pc.check_spgemm_block_kernel(be)


def test_spgemm_value_walk_steps_in_flight(be):
    pc.check_spgemm_val_steps(be)


def test_spgemm_entries_sorted_in_lds(be):
    pc.check_spgemm_sorted_emission(be)


def test_spgemm_four_rows_per_wave(be):
    pc.check_spgemm_quad_rows(be)


def test_spgemm_issue402(be):
    g = np.load(os.path.join(GOLD, "matrix_issue402.npz"))
    A0 = oracle.Crs(1813, 1813, g["row_map"], g["entries"], g["values"])
    At = oracle.transpose(A0)
    oracle.sort_crs(A0); oracle.sort_crs(At)
    sub = 300    # leading principal block keeps the emulator run short; the GPU test uses the full matrix
    def head(M, r):
        return oracle.Crs(r, M.ncols, M.row_map[:r + 1], M.entries[:M.row_map[r]], M.values[:M.row_map[r]])
    pc.check_spgemm(be, head(A0, sub), At, reuse=False)


def test_spgemm_handle_contract(be):
    A0 = pc.randomized(oracle.random_crs(40, 30, 5, seed=7, sorted_rows=True))
    B0 = pc.randomized(oracle.random_crs(30, 20, 4, seed=8, sorted_rows=True))
    A, B = pc.dev(be, A0), pc.dev(be, B0)
    kh = pc.kk.KokkosKernelsHandle(be)
    with pytest.raises(ValueError, match="does not have an SpGEMM handle"):
        pc.kk.spgemm_symbolic(kh, A, False, B, False)
    kh.create_spgemm_handle()
    with pytest.raises(RuntimeError, match="transposing"):
        pc.kk.spgemm_symbolic(kh, A, True, B, False)
    # numeric before symbolic (Test_Sparse_spgemm.hpp:444-481)
    Cfake = pc.kk.CrsMatrix(40, 20, be.empty(41, np.int32), be.empty(1, np.int32), be.empty(1, np.float64), backend=be)
    with pytest.raises(ValueError, match="must first call spgemm_symbolic"):
        pc.kk.spgemm_numeric(kh, A, False, B, False, Cfake)
    C1 = pc.kk.spgemm_symbolic(kh, A, False, B, False)
    C2 = pc.kk.spgemm_symbolic(kh, A, False, B, False)          # repeated symbolic is idempotent (:315-370)
    assert np.array_equal(be.to_numpy(C1.graph.row_map), oracle.spgemm_symbolic(A0, B0)[0])
    assert C2.nnz() == C1.nnz()
    assert kh.get_spgemm_handle().is_symbolic_called() and not kh.get_spgemm_handle().is_numeric_called()
    pc.kk.spgemm_numeric(kh, A, False, B, False, C1)
    assert kh.get_spgemm_handle().is_numeric_called()
    kh.destroy_spgemm_handle()
    Cn = pc.kk.spgemm(A, False, B, False)                        # no-reuse interface
    ok, msg = oracle.is_same_matrix(oracle.Crs(40, 20, *[np.asarray(v) for v in Cn.to_host()]), oracle.spgemm(A0, B0))
    assert ok, msg
    with pytest.raises(RuntimeError, match="numCols"):
        pc.kk.spgemm(A, False, A, False)


def test_spgemm_compression(be):
    """a18: B compressed into 32-column sets + masks for the symbolic phase (impl_compression.hpp): forced on every row bin
    (wave, block-small, block-large, bitmap), kept by the 0.85 rule on stencils and dropped on matrices without column runs"""
    L = oracle.laplace3d("FE", 9, 8, 7)
    pc.check_spgemm(be, L, L, expect_compressed=False)                                  # off unless asked for
    pc.check_spgemm(be, L, L, options={"compression": 1}, expect_compressed=True)       # runs of three neighbours: pays
    pc.check_spgemm(be, L, L, options={"compression": 0}, expect_compressed=False)
    R = pc.randomized(oracle.random_crs(400, 30000, 9, variance=4, seed=2, sorted_rows=True))
    Rt = pc.randomized(oracle.random_crs(30000, 40000, 7, variance=3, seed=3, sorted_rows=True))
    pc.check_spgemm(be, R, Rt, options={"compression": 1}, expect_compressed=False)     # scattered columns: dropped
    pc.check_spgemm(be, R, Rt, options={"compression": 2}, expect_compressed=True)      # ... unless forced
    pc.check_spgemm(be, L, L, options={"compression": 1, "compression_cut_off": 0.1}, expect_compressed=False)
    # every symbolic bin with compressed input: long rows of A against a banded B, hub rows (bitmap kernel, several windows)
    band = pc.randomized(oracle.random_crs(6000, 6000, 40, variance=10, seed=5, bandwidth=150, sorted_rows=True))
    lens = [3, 40, 200, 900, 2500, 0, 60]
    rm = np.concatenate([[0], np.cumsum(lens)])
    rng = np.random.default_rng(9)
    ent = np.concatenate([np.sort(rng.choice(6000, size=l, replace=False)) for l in lens]).astype(np.int32)
    A0 = oracle.Crs(len(lens), 6000, rm, ent, 1 + 49 * rng.random(int(rm[-1])))
    for odt in (np.int32, np.int64):
        pc.check_spgemm(be, A0, band, offset_dtype=odt, options={"compression": 2}, expect_compressed=True)
    try:
        _set(be, "spgemm_win_bits", 4096)
        pc.check_spgemm(be, A0, band, options={"compression": 2}, expect_compressed=True)
    finally:
        _set(be, "spgemm_win_bits", 1 << 20)
    # unsorted B cannot be compressed (a set is a run of a sorted row)
    U = pc.randomized(oracle.random_crs(300, 300, 8, variance=3, seed=8))
    pc.check_spgemm(be, U, U, options={"compression": 2}, expect_compressed=False)


def test_spgemm_dense_accumulator_algorithm(be):
    """a21: SPGEMM_KK_DENSE / SPGEMM_ACC_DENSE -- every row through a k-wide dense accumulator (impl_speed.hpp:28-150)"""
    for A0, B0 in ((pc.randomized(oracle.laplace3d("FE", 7, 6, 5)),) * 2,
                   (pc.randomized(oracle.random_crs(120, 900, 11, variance=6, seed=4, sorted_rows=True)),
                    pc.randomized(oracle.random_crs(900, 700, 9, variance=5, seed=6, sorted_rows=True))),
                   (pc.hub_matrix(30, 2000, 6, {3: 700}, seed=2), pc.randomized(oracle.random_crs(2000, 1500, 10, variance=4, seed=7, sorted_rows=True)))):
        got = pc.check_spgemm(be, A0, B0, algo="SPGEMM_KK_DENSE")
        pc.check_spgemm(be, A0, B0, options={"accumulator": 1}, offset_dtype=np.int64, value_dtype=np.float32)
    kh = pc.kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK_DENSE")
    assert kh.get_spgemm_handle().get(8) == 1
    for alias in ("SPGEMM_KK_MEMORY", "SPGEMM_KK_SPEED", "SPGEMM_KK_MEMSPEED", "SPGEMM_KK_LP", "SPGEMM_DEFAULT"):
        kh.create_spgemm_handle(alias)
        assert kh.get_spgemm_handle().get(8) == 0


def test_spgemm_numeric_reuse_keeps_entries(be):
    """numeric reuse (sparse/tpls/KokkosSparse_spgemm_numeric_tpl_spec_decl.hpp:288-329, Test_Sparse_spgemm.hpp:243-252): a repeated
    numeric call on the same handle and the same C arrays keeps entries(C) of the dense rows (no second structure pass); new C
    arrays, or the caller saying the entries are gone, write them again"""
    B0 = pc.hub_matrix(60, 30000, 20, {0: 9000, 1: 7000, 5: 12000}, seed=11)
    A0 = pc.randomized(oracle.random_crs(40, 60, 12, variance=4, seed=3, sorted_rows=True))
    A, B = pc.dev(be, A0), pc.dev(be, B0)
    gold = oracle.spgemm(A0, B0)
    kh = pc.kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
    sh = kh.get_spgemm_handle()
    Cm = pc.kk.spgemm_symbolic(kh, A, False, B, False)

    def check(tag, Ah=A0, Ad=A, Cd=None):
        Cd = Cd or Cm
        pc.kk.spgemm_numeric(kh, Ad, False, B, False, Cd)
        rm, ent, val = Cd.to_host()
        g = gold if Ah is A0 else oracle.spgemm(Ah, B0)
        ok, msg = oracle.is_same_matrix(oracle.Crs(A0.nrows, B0.ncols, rm.astype(np.int64), ent, val), g)
        assert ok, tag + ": " + msg
    kept = sh.get(13)                                        # rows of the symbolic bitmap bin (more than 2048 products) with at least k / 32 entries
    lenB = np.diff(B0.row_map)
    flops = np.array([lenB[A0.entries[A0.row_map[i]:A0.row_map[i + 1]]].sum() for i in range(A0.nrows)])
    assert kept == int(((np.diff(gold.row_map) >= B0.ncols // 32) & (flops > 2048)).sum()) and kept > 0, kept
    check("first numeric"); assert sh.get(11) == 0
    assert sh.get(12) == kept and sh.get(13) == 0, (sh.get(12), sh.get(13))     # entries(C) of those rows came from the bitmaps, which are gone now
    A2 = pc.randomized(A0, seed=77)
    Ad2 = pc.kk.CrsMatrix(A0.nrows, A0.ncols, A.graph.row_map, A.graph.entries, be.from_numpy(A2.values), backend=be)
    check("reuse, new values", A2, Ad2); assert sh.get(11) == 1, "the structure pass of the dense rows ran again"
    Cm.graph.entries[:] = -1                               # the caller clobbered entries(C) and says so
    sh.set("entries_computed", 0)
    check("entries_computed = 0", A2, Ad2); assert sh.get(11) == 0
    check("reuse again", A0, A); assert sh.get(11) == 1
    C2 = pc.kk.CrsMatrix(A0.nrows, B0.ncols, Cm.graph.row_map, be.empty(Cm.nnz(), np.int32), be.empty(Cm.nnz(), np.float64), backend=be)
    C2.graph.entries[:] = -7
    check("other C arrays", A0, A, C2); assert sh.get(11) == 0
    sh.set("algorithm", pc.kk.sparse._SPGEMM_ALGOS["SPGEMM_KK_DENSE"])     # other row bins: nothing may be kept
    check("algorithm changed", A0, A, C2); assert sh.get(11) == 0


def test_spgemm_options_act_or_raise(be, capfd):
    """handle options: what this implementation has acts, the reference's tuning hints are accepted and recorded (as in the
    reference's own TPL paths; its driver and unit tests set them before every spgemm), unknown keys raise"""
    kh = pc.kk.KokkosKernelsHandle(be)
    L = pc.randomized(oracle.laplace3d("FE", 6, 5, 4))
    A = pc.dev(be, L)
    for algo in ("SPGEMM_DEBUG", "SPGEMM_SERIAL"):        # host-sequential in the reference: same C (numeric_spec.hpp:138-140 sorts every algorithm's rows)
        kh.create_spgemm_handle(algo)
        assert kh.get_spgemm_handle().get(8) == 0 and kh.get_spgemm_handle().get(9) == pc.kk.sparse._SPGEMM_ALGOS[algo]
        pc.check_spgemm(be, L, L, algo=algo)
    with pytest.raises(RuntimeError):
        kh.create_spgemm_handle("SPGEMM_CUSPARSE")
    kh.create_spgemm_handle()
    sh = kh.get_spgemm_handle()
    hints = ("team_work_size", "shmem_size", "suggested_team_size", "suggested_vector_size", "dynamic_scheduling", "min_hash_size_scale",
             "first_level_hash_cut_off", "mkl_sort_option", "read_write_cost_calc", "compression_steps", "max_col_dense_acc")
    for i, key in enumerate(hints):
        sh.set(key, 16 + i)
        assert sh.get_hint(key) == 16 + i
    assert sh.get(10) == len(hints)
    with pytest.raises(pc.kk.KkamdError):
        sh.get_hint("multi_color_scale")                  # never set
    with pytest.raises(pc.kk.KkamdError):
        sh.set("no_such_option", 1)
    sh.set("sort_option", 0)                              # rows of C leave sorted whatever is asked
    sh.set("sort_option", 1); sh.set("verbose", 1); sh.set("compression", 1)
    sh.set("team_work_size", 256)
    Cm = pc.kk.spgemm_symbolic(kh, A, False, A, False)
    pc.kk.spgemm_numeric(kh, A, False, A, False, Cm)
    out = capfd.readouterr().out
    assert "kkamd spgemm symbolic" in out and "compression kept" in out and "kkamd spgemm numeric (SPGEMM_KK)" in out, out
    assert "hint team_work_size = 256 recorded" in out, out
