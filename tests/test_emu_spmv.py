"""Kernel-LOGIC tests of the SpMV kernels under the SIMT emulator (tests/emu): the same .hip sources as
the product, compiled for the CPU against kk_emu.h.  These run without a GPU; the GPU parity tests
(test_gpu_*.py) exercise the real gfx950 build through the C ABI."""
import numpy as np
import pytest

import oracle
import parity_cases as pc
from emu import emu_backend


@pytest.fixture(scope="module")
def be():
    return emu_backend.backend()


ALGOS = [None, "SPMV_DEFAULT", "SPMV_FAST_SETUP", "SPMV_NATIVE", "SPMV_MERGE_PATH", "SPMV_NATIVE_MERGE_PATH"]


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("mode", ["N", "C", "T", "H"])
def test_spmv_algorithms_modes_alpha_beta(be, algo, mode):
    # Test_Sparse_spmv.hpp:415-451 (heavy sweep) on a 1000-row matrix with 3..20 nnz/row
    A0 = oracle.random_crs(700, 650, 11, variance=8, seed=1)
    for alpha in (0.0, 1.0, -1.0, 2.5):
        for beta in (0.0, 1.0, -1.0, 2.5):
            pc.check_spmv(be, A0, mode, alpha, beta, algo)
            if beta == 0.0:
                pc.check_spmv(be, A0, mode, alpha, beta, algo, nans=True)


@pytest.mark.parametrize("npt", [4, 8, 16])
def test_stream_kernel_tilings(be, npt):
    # rows short and long relative to the tile, nnz not a multiple of the tile, rows straddling tiles
    for nnz_row, var, n in ((27, 0, 600), (3, 2, 4000), (300, 250, 90), (1, 0, 5000)):
        A0 = oracle.random_crs(n, n + 13, nnz_row, variance=var, seed=npt + nnz_row)
        pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs={"nnz_per_thread": npt})
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs={"nnz_per_thread": npt, "xcd_remap": 0})


@pytest.mark.parametrize("variant", [1, 6])
def test_stream_variants(be, variant):
    # the planned kernel with the default analysis (1) and with the column codes attempted whatever the size (6)
    mats = [oracle.laplace3d("FE", 12, 11, 10), oracle.random_crs(900, 880, 13, variance=9, seed=2), oracle.random_crs(2000, 2000, 25, variance=5, seed=3, bandwidth=40)]
    for A0 in mats:
        for npt in (4, 8, 16):
            pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs={"nnz_per_thread": npt, "stream_variant": variant}, max_val=32.0)
            pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs={"nnz_per_thread": npt, "stream_variant": variant}, max_val=32.0)


def test_window_codes(be):
    # stream_variant 6: columns from 16-bit window codes where every tile fits 16 windows, plain entries otherwise
    for ci, (name, A0, ok) in enumerate(pc.window_code_cases()):
        for npt in ((4, 8, 16) if ci < 2 else (8,)):        # every tile size on the first two matrices (the GPU suite runs them all)
            kn = {"nnz_per_thread": npt, "stream_variant": 6}
            pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs=kn, max_val=32.0, expect={"window_codes": ok})
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs={"stream_variant": 6}, max_val=32.0, expect={"window_codes": ok})
        # the default kernel tries the codes by itself from window_codes_min_knnz on, and never with window_codes = 0
        pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"window_codes_min_knnz": 0}, max_val=32.0, expect={"window_codes": ok, "tile": 2048})
        if ci < 2:                                              # (the GPU suite runs every case through all of these)
            pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", max_val=32.0, expect={"window_codes": 0})
            pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"window_codes_min_knnz": 0, "window_codes": 0}, max_val=32.0, expect={"window_codes": 0})
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs={"stream_variant": 6}, max_val=32.0, offset_dtype=np.int64,
                      value_dtype=np.float32, expect={"window_codes": ok})
        # window_codes 2: the codes without the staged x window
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs={"window_codes_min_knnz": 0, "window_codes": 2}, max_val=32.0,
                      expect={"window_codes": ok, "window_staged_x": 0})
    # contiguous column runs: the staged x window is used; a tile whose runs do not fit it keeps the gather
    for A0, npt, staged in ((oracle.laplace3d("FE", 64, 40, 9), 8, 1), (oracle.laplace3d("FD", 70, 30, 12), 8, 1), (oracle.laplace3d("FE", 64, 40, 9), 4, 1),
                            (oracle.laplace3d("FE", 64, 40, 9), 16, 1), (oracle.laplace2d("FD", 500, 37), 8, 1), (oracle.laplace3d("FE", 700, 5, 4), 16, 1),
                            (pc.window_code_cases()[0][1], 8, None)):
        # (the last case: some tiles' runs do not fit the LDS window -- those keep the gather, tile by tile)
        h = pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", nans=False, knobs={"stream_variant": 6, "nnz_per_thread": npt}, max_val=32.0,
                          expect={"window_codes": 1} if staged is None else {"window_codes": 1, "window_staged_x": staged})
        if staged is None: assert 0 < h.query("staged_tiles") < h.query("tiles")
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs={"stream_variant": 6, "nnz_per_thread": npt}, max_val=32.0)
        if npt != 8 or staged is None:                          # the other tile sizes and the mixed case also with 64-bit offsets / fp32 values
            pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs={"stream_variant": 6, "nnz_per_thread": npt}, max_val=32.0, offset_dtype=np.int64,
                          value_dtype=np.float32, expect={"window_codes": 1})


def test_pattern_codes(be):
    # row-pattern records instead of per-nonzero codes (staged-x kernel): forced on for every tile that has one
    for name, A0, npt, expect_pat in pc.pattern_code_cases():
        kn = {"window_codes_min_knnz": 0, "nnz_per_thread": npt, "pattern_codes": 2}
        for odt, vdt, beta in ((np.int32, None, 0.5), (np.int64, np.float32, 0.0)):
            h = pc.check_spmv(be, A0, "N", 1.5, beta, "SPMV_DEFAULT", knobs=kn, max_val=32.0, nans=(beta == 0.0), offset_dtype=odt, value_dtype=vdt,
                              expect={"window_staged_x": 1})
            assert (h.query("pattern_tiles") > 0) == expect_pat, (name, h.query("pattern_tiles"), h.query("tiles"))
        # auto (90 % of the tiles, from pattern_codes_min_knnz on) and off give the same y
        pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"window_codes_min_knnz": 0, "nnz_per_thread": npt, "pattern_codes_min_knnz": 0}, max_val=32.0)
        pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"window_codes_min_knnz": 0, "nnz_per_thread": npt}, max_val=32.0,
                      expect={"pattern_tiles": 0})                 # below the default size threshold
        pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"window_codes_min_knnz": 0, "nnz_per_thread": npt, "pattern_codes": 0,
                                                                       "pattern_codes_min_knnz": 0}, max_val=32.0, expect={"pattern_tiles": 0})


def test_pattern_records_straight_from_the_matrix(be):
    pc.check_pattern_direct(be)


def test_mixed_tiles(be):
    # the column analysis is per tile: tiles the windows cannot cover read entries, the others keep codes / staged x / records
    for name, A0 in pc.mixed_tile_cases():
        for npt, pat in ((8, 0), (8, 2), (4, 2), (16, 0)):         # (the GPU suite runs all six combinations)
            kn = {"window_codes_min_knnz": 0, "nnz_per_thread": npt, "pattern_codes": pat, "window_codes_min_pct": 10}
            h = pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs=kn, max_val=32.0, expect={"window_codes": 1})
            assert 0 < h.query("plain_tiles") < h.query("tiles"), (name, npt, h.query("plain_tiles"), h.query("tiles"))
            assert h.query("code_tiles") + h.query("pattern_tiles") + h.query("plain_tiles") == h.query("tiles")
            pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs=kn, max_val=32.0, nans=True, offset_dtype=np.int64, value_dtype=np.float32)
        # too few coverable tiles for the threshold: plain kernel
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs={"window_codes_min_knnz": 0, "window_codes_min_pct": 100}, max_val=32.0,
                      expect={"window_codes": 0})


def test_knob_validation(be):
    # knobs that would break the tile permutation or select a kernel that does not exist are rejected (not silently accepted)
    import kk_loader
    kk = kk_loader.load()
    A0 = oracle.random_crs(3000, 3000, 9, seed=1)
    for key, val in (("xcd_remap", 3), ("xcd_remap", 6), ("mv_remap", 12), ("nnz_per_thread", 5), ("stream_variant", 2), ("ablate", 1),
                     ("lds_pad_kb", 8), ("nontemporal", 1), ("kernel", 7), ("window_codes", 9), ("colslab", 5), ("colslab_shift", 1), ("colslab_shift", 31),
                     ("colslab_const", 2), ("colslab_min_knnz", -1), ("mv4_xcol", 2), ("mv4_2d", -1)):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set(key, val)
        with pytest.raises(kk.KkamdError):
            pc.check_spmv(be, A0, "N", 1.0, 0.0, None) if False else kk.spmv(h, "N", 1.0, pc.dev(be, A0), be.from_numpy(np.ones(3000)), 0.0, be.from_numpy(np.zeros(3000)))
    # an fp32-valued matrix re-analysed with 1024-nnz tiles asked for: the plan keeps a tile size its kernels exist for
    h = pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", value_dtype=np.float32, vec_dtype=np.float32)
    h.set("nnz_per_thread", 4)
    assert h.query("tile") in (0, 2048)


def test_mv3_lds_staged(be):
    # rank-2 kernel over LDS-staged X tiles (analysed handles): staged and gather tiles, 8 / 16 / 24 / 32 right-hand sides,
    # both layouts, beta = 0 over NaNs, every tile order
    combos = ((16, "C", "C", 1.5, 0.5), (8, "C", "C", 1.0, 0.0), (32, "C", "C", -1.0, 0.0), (24, "C", "F", 1.0, 1.0), (16, "F", "F", 2.0, 0.0), (16, "F", "C", 1.0, -1.0))
    for ci, (name, A0, staged) in enumerate(pc.mv3_cases()):
        # (an opt-in kernel: every combination on the first two matrices, two on the others -- the GPU suite runs them all)
        for nvec, xo, yo, alpha, beta in (combos if ci < 2 else (combos[0], combos[3])):
            h = pc.check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", knobs={"mv_kernel": 3}, max_val=32.0,
                                 nans=(beta == 0.0))
            assert (h.query("mv_staged_tiles") > 0) == staged, (name, nvec, h.query("mv_staged_tiles"), h.query("mv_tiles"))
    A0 = oracle.laplace3d("FE", 37, 11, 9)
    for order in (0, 1, 2):
        pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", knobs={"mv_kernel": 3, "mv_order": order}, max_val=32.0, nans=True)
    # strip order (every XCD walks strips of the grid plane after plane): thresholds lowered so that a small grid engages it
    A0 = oracle.laplace3d("FE", 130, 10, 6)
    h = pc.check_spmv_mv(be, A0, 16, "N", 1.5, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, nans=True,
                         knobs={"mv_kernel": 3, "mv_order": 2, "mv_strip_min_kb": 100, "mv_strip_l2_kb": 64}, expect={"mv_order": 2})
    # the staged kernel runs on request only (knob mv_kernel = 3); widths that are not multiples of 8 never take it
    h = pc.check_spmv_mv(be, oracle.laplace3d("FE", 300, 10, 5), 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, nans=True,
                         knobs={"mv_kernel": 2, "mv_strip_min_kb": 100, "mv_strip_l2_kb": 160}, expect={"mv_tiles": 0, "mv_order": 2, "mv_period": 3000})
    pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, expect={"mv_tiles": 0})
    pc.check_spmv_mv(be, A0, 12, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", knobs={"mv_kernel": 3}, max_val=32.0, expect={"mv_tiles": 0})


def test_mv4_plane_marching(be):
    # rank-2 plane-marching kernel (analysed handles, fp64, right-hand sides in blocks of 16, lattice stencils): every lattice
    # row it takes (interior and truncated boundary rows) and the rows it leaves to the gather kernel; both layouts (X packed
    # per call, Y strided), alpha / beta, beta = 0 over NaNs, 64-bit offsets, fp32 values, k-chunks from 1 to nz / 4
    combos = ((16, "C", "C", 1.5, 0.5, np.int32), (16, "C", "C", 1.0, 0.0, np.int64), (32, "C", "C", -1.0, 0.0, np.int32),
              (16, "F", "F", 2.0, 0.0, np.int32), (16, "F", "C", 1.0, -1.0, np.int32), (48, "C", "F", 1.0, 1.0, np.int32),
              (21, "C", "C", 0.5, 0.0, np.int32), (37, "F", "F", 1.0, 2.0, np.int32))     # 16 + 5, 32 + 5: the remainder is a partial block
    for ci, (name, A0, left) in enumerate(pc.mv4_cases()):
        for nvec, xo, yo, alpha, beta, off in (combos if ci < 3 else (combos[0], combos[3], combos[6])):    # all of them on the first three matrices
            h = pc.check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0), offset_dtype=off)
            assert h.query("mv4_workgroups") > 0, (name, nvec)
            if left is not None:
                assert h.query("mv4_other_rows") == left, (name, h.query("mv4_other_rows"))
    name, A0, _ = pc.mv4_cases()[0]
    for wg in (1, 64):
        pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", knobs={"mv_kernel": 4, "mv4_wg_per_cu": wg}, max_val=32.0, nans=True)
    pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, value_dtype=np.float32)
    # column-major X takes the column-wise piece order with swizzled slab rows by default; mv4_xcol 0 keeps the general-stride order
    for xcol in (0, 1):
        h = pc.check_spmv_mv(be, A0, 32, "N", 1.5, 0.0, "F", "F", algo="SPMV_DEFAULT", knobs={"mv4_xcol": xcol}, max_val=32.0, nans=True)
        assert h.query("mv4_workgroups") > 0
    # Inf and NaN in X reach exactly the rows the reference lets them reach (no 0 * Inf from halo or pad entries): a corner,
    # a face, an interior point
    name, A0, _ = pc.mv4_cases()[3]
    h = pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, nans=True,
                         x_special={0: np.inf, 33 * 6 * 10 + 5: -np.inf, A0.nrows - 1: np.nan, 33 * 6 * 7 + 33 * 2 + 16: np.inf})
    assert h.query("mv4_workgroups") > 0
    # narrow multivectors and remainders run the partial-block form (columns past the block clamped on the X side, masked on the Y side):
    # every width from 4 to 15, both layouts, beta != 0 over the masked columns' neighbours
    for nvec in (4, 5, 8, 11, 12, 15):
        for xo, yo, beta in ((("C", "C", 0.0), ("F", "F", 0.5), ("C", "F", -1.0)) if nvec in (5, 12) else (("C", "C", 0.0), ("F", "F", 0.5))[nvec % 2:][:1]):
            h = pc.check_spmv_mv(be, A0, nvec, "N", 1.5, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0))
            assert h.query("mv4_workgroups") > 0, (nvec, xo, yo)
    # 2-D lattices: the lines are grouped m at a time into "planes" (m: a divisor of the line count in 32..128); the first and last
    # line of every group go to the gather rows, the rest marches
    for st, nxl, nyl, m_ in (("FE", 70, 128, 32), ("FD", 40, 256, 64)):
        A2 = oracle.laplace2d(st, nxl, nyl)
        for nvec, xo, yo, beta in ((16, "C", "C", 0.0), (32, "F", "F", 0.5), (5, "C", "F", 0.0)):
            h = pc.check_spmv_mv(be, A2, nvec, "N", 1.5, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0))
            assert h.query("mv4_workgroups") > 0 and h.query("mv4_other_rows") == 2 * (nyl // m_ - 1) * nxl, (st, h.query("mv4_workgroups"), h.query("mv4_other_rows"))
        h = pc.check_spmv_mv(be, A2, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", knobs={"mv4_2d": 0}, max_val=32.0)
        assert h.query("mv4_workgroups") == 0
    # not its matrices / widths: no far stride and no usable line count (2-D, 41 lines), too few lattice rows, 3 right-hand sides (below mv4_min_nvec), no analysis, the gather kernel asked for
    for A1, nvec, algo, knobs in ((oracle.laplace2d("FE", 130, 41), 16, "SPMV_DEFAULT", None), (oracle.laplace3d("FE", 12, 12, 12), 16, "SPMV_DEFAULT", None),
                                  (A0, 3, "SPMV_DEFAULT", None), (A0, 8, "SPMV_DEFAULT", {"mv4_min_nvec": 16}), (A0, 16, "SPMV_FAST_SETUP", None), (A0, 16, "SPMV_DEFAULT", {"mv_kernel": 2}),
                                  (oracle.random_crs(5000, 5000, 9, variance=3, seed=5), 16, "SPMV_DEFAULT", None)):
        h = pc.check_spmv_mv(be, A1, nvec, "N", 1.0, 0.0, "C", "C", algo=algo, knobs=knobs, max_val=32.0)
        assert h.query("mv4_workgroups") == 0


def test_mv5_matrix_core(be):
    # rank-2 matrix-core kernel (kk_spmv_mvblk.hip: 16-row tiles, union of columns in blocks of four, one v_mfma_f64_16x16x4 per block):
    # described tiles and the rows left to its gather rows, every width, the four layout pairs (column-major Y swaps the operands),
    # beta = 0 over NaNs, 64-bit offsets, fp32 values, Inf / NaN in X (a tile that sees one recomputes entry by entry)
    pc.check_mv5(be, light=True)


def test_check_entries_knob(be):
    pc.check_entries_guard(be)


def test_column_slab_deterministic_form(be):
    pc.check_colslab_deterministic(be)


def test_transposed_plan_inherits_the_handles_knobs(be):
    pc.check_transpose_plan_inherits_knobs(be)


def test_handle_that_begins_with_rank2_defers_the_rank1_analysis(be):
    pc.check_rank2_first_handle_defers_rank1(be)


def test_values_tracking_policies(be):
    # exact (default) / notify / fingerprints for the cached transpose and the column-slab copy; kkamd_spmv_plan_values_changed
    pc.check_values_tracking(be)


def test_mv_transposed_modes_through_cached_transpose(be):
    # rank 2, modes T / H of an analysed handle: the mode-N dispatch on the cached transpose
    pc.check_mv_transpose_cached(be, light=True)


def test_mv6_nonzero_split(be):
    # rank-2 nonzero-split kernel (kk_spmv_mvnnz.hip): chunks of 128 entries per 16-lane group, cut rows finished from carries, empty
    # rows from the plan's list; every width / layout pair, beta = 0 over NaNs, 64-bit offsets, fp32 values, Inf / NaN in X
    pc.check_mv6(be, light=True)


def test_mv_long_rows(be):
    # rank 2 on a matrix with a few very long rows (R-MAT-like hubs): the wave-private gather kernel leaves rows above 32 x the average
    # length (at least 1024 entries) to spmv_mv_long_kernel (a workgroup per row); every width, both layouts, beta 0 over NaNs and != 0
    A0 = pc.hub_matrix(3000, 9000, 6, {5: 7000, 17: 1500, 1234: 1025, 2999: 4000, 40: 1024}, seed=3)
    for nvec, xo, yo, alpha, beta in ((16, "C", "C", 1.5, 0.0), (16, "F", "F", 1.0, 0.5), (5, "C", "C", 2.0, -1.0), (33, "C", "F", 1.0, 0.0), (2, "C", "C", 1.0, 1.0)):
        h = pc.check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", max_val=50.0, nans=(beta == 0.0), knobs={"mv_long_T": 1024, "mv6": 0})
        assert h.query("mv_long_rows") == 4, h.query("mv_long_rows")            # 7000, 1500, 1025, 4000 (1024 itself stays)
        h = pc.check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", max_val=50.0, nans=(beta == 0.0), knobs={"mv6": 0})
        assert h.query("mv_long_rows") == 5, h.query("mv_long_rows")            # automatic threshold: 4 x the average row, at least 64
    h = pc.check_spmv_mv(be, pc.randomized(oracle.random_crs(2000, 2000, 9, variance=3, seed=5)), 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT")
    assert h.query("mv_long_rows") == 0


def test_mv4_widths_beyond_one_block(be):
    # widths 24, 40 and 47 (the emulator takes a minute per case; the GPU suite runs every width from 17 to 47) on a lattice matrix: full passes of 16 columns + one partial pass, row-major and column-major multivectors in turn,
    # beta = 0 over NaNs and beta != 0 (round-4 review item 8 asks for the parity of these widths whatever their speed)
    name, A0, _ = pc.mv4_cases()[0]
    for nvec in (24, 40, 47):
        xo, yo = (("C", "C"), ("F", "F"), ("C", "F"))[nvec % 3]
        beta = 0.0 if nvec % 2 else 0.5
        h = pc.check_spmv_mv(be, A0, nvec, "N", 1.5, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0), seed=nvec)
        assert h.query("mv4_workgroups") > 0, nvec


def test_mv4_duplicate_entries(be):
    # a lattice row that stores one column twice must never become the plane-marching pattern, and a row
    # with a duplicate must go to the gather rows (the reference sums duplicates); rank 2 and the rank-1 marching kernel
    for name, A0, planned, left in pc.mv4_duplicate_cases():
        for knobs in (None, {"mv_kernel": 4}):
            h = pc.check_spmv_mv(be, A0, 16, "N", 1.5, 0.0, "C", "C", algo="SPMV_DEFAULT", knobs=knobs, max_val=32.0, nans=True)
            assert (h.query("mv4_workgroups") > 0) == planned, (name, knobs)
            if planned:
                assert h.query("mv4_other_rows") == left, (name, h.query("mv4_other_rows"))
        h = pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, max_val=32.0, knobs={"march": 1})
        assert (h.query("march_workgroups") > 0) == planned, name


def test_march_rank1(be):
    # rank 1 on the plane-marching analysis (knob march): every lattice matrix of the rank-2 cases, alpha / beta, beta = 0 over
    # NaNs, 64-bit offsets, fp32 values, k-chunks of 1 plane, a few planes, the whole lattice; mode T through the cached transpose
    for ci, (name, A0, left) in enumerate(pc.mv4_cases()):
        for alpha, beta, off, planes in ((1.0, 0.0, np.int32, 20), (1.5, -0.5, np.int64, 3), (2.0, 1.0, np.int32, 1000))[:3 if ci < 3 else 1]:
            h = pc.check_spmv(be, A0, "N", alpha, beta, "SPMV_DEFAULT", nans=(beta == 0.0), offset_dtype=off, max_val=32.0,
                              knobs={"march": 1, "march_planes": planes})
            assert h.query("march_workgroups") > 0, name
            if left is not None:
                assert h.query("mv4_other_rows") == left, (name, h.query("mv4_other_rows"))
    name, A0, _ = pc.mv4_cases()[0]
    pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=32.0, knobs={"march": 1, "march_planes": 1})
    pc.check_spmv(be, A0, "N", 1.0, 0.5, "SPMV_DEFAULT", max_val=32.0, knobs={"march": 1}, value_dtype=np.float32)
    pc.check_spmv(be, A0, "T", 1.0, 0.0, "SPMV_DEFAULT", max_val=32.0, knobs={"march": 1})
    # off by default; matrices that are not lattices keep the planned stream kernel
    h = pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=32.0)
    assert h.query("march_workgroups") == 0
    h = pc.check_spmv(be, oracle.random_crs(5000, 5000, 9, variance=3, seed=5), "N", 1.0, 0.0, "SPMV_DEFAULT", knobs={"march": 1})
    assert h.query("march_workgroups") == 0


def test_xcd_group_orders(be):
    # grouped tile orders (xcd_remap / mv_remap = G): whole blocks of 8G tiles are permuted, the incomplete last block is not
    for nrows in (64 * 15 + 5, 64 * 16, 64 * 17 + 1, 64 * 130):
        A0 = oracle.random_crs(nrows, nrows + 7, 16, variance=0, seed=nrows)       # 1024-nnz tiles: nrows / 64 of them
        for g in (2, 16, 32, 0):
            pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs={"nnz_per_thread": 4, "xcd_remap": g}, expect={"tile": 1024})
        for g in (4, 16, 0):
            pc.check_spmv_mv(be, A0, 4, "N", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", knobs={"mv_remap": g})


def _custom(lens, ncols, seed=0):
    rng = np.random.default_rng(seed)
    lens = np.asarray(lens)
    rm = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = rng.integers(0, ncols, size=rm[-1]).astype(np.int32)
    return oracle.Crs(len(lens), ncols, rm, ent, rng.random(rm[-1]))


@pytest.mark.parametrize("algo", ["SPMV_DEFAULT", "SPMV_FAST_SETUP"])
def test_spmv_row_shapes(be, algo):
    cases = {
        "one_row_many_tiles": [9000],
        "long_rows_span_tiles": [5000, 1, 0, 4100, 2048, 2048, 3],
        "empty_rows_everywhere": [0, 0, 5, 0, 0, 0, 7, 0, 2047, 1, 0, 0],
        "all_empty_but_one": [0] * 300 + [4] + [0] * 300,
        "exact_tile_multiple": [1024, 1024, 2048, 0, 0],
        "leading_trailing_empty": [0] * 50 + [30] * 200 + [0] * 70,
        "single_entry": [1],
    }
    for name, lens in cases.items():
        A0 = _custom(lens, 977, seed=len(lens))
        for beta in (0.0, 2.0):
            pc.check_spmv(be, A0, "N", -1.5, beta, algo, nans=(beta == 0.0))


def test_spmv_int64_offsets_and_float(be):
    A0 = oracle.random_crs(500, 480, 9, variance=5, seed=3)
    pc.check_spmv(be, A0, "N", 1.0, 1.0, "SPMV_DEFAULT", offset_dtype=np.int64)
    pc.check_spmv(be, A0, "T", 1.0, 1.0, None, offset_dtype=np.int64)
    pc.check_spmv(be, A0, "N", 2.0, 0.5, "SPMV_DEFAULT", value_dtype=np.float32, vec_dtype=np.float32)
    pc.check_spmv(be, A0, "N", 2.0, 0.5, None, value_dtype=np.float32, vec_dtype=np.float32)
    pc.check_spmv(be, A0, "N", 2.0, 0.5, "SPMV_DEFAULT", value_dtype=np.float32)       # float matrix, double vectors


def test_github_issue_101(be):
    # Test_Sparse_spmv.hpp:823-961 -- exact known answer, rank-1 and 1..22 right-hand sides, double and mixed
    expected = 1.0 + pc.EPS_F / 2.0
    for vdt in (np.float64, np.float32):
        A = pc.kk.CrsMatrix.from_host(1, 2, [0, 2], [0, 1], np.array([1.0, pc.EPS_F / 2.0], dtype=vdt), backend=be)
        for h in (None, pc.kk.SPMVHandle("SPMV_DEFAULT")):
            y = be.from_numpy(np.zeros(1))
            args = ("N", 1.0, A, be.from_numpy(np.ones(2)), 0.0, y)
            pc.kk.spmv(*args) if h is None else pc.kk.spmv(h, *args)
            assert be.to_numpy(y)[0] == expected
        for nv in range(1, 23):
            X = np.ones((2, nv), order="F"); Y = np.zeros((1, nv), order="F")
            pc.kk.spmv("N", 1.0, A, X, 0.0, Y)
            assert (Y == expected).all()


def test_wiki_example_and_structured(be):
    A0 = oracle.laplace2d("FD", 10, 10, bc=(0, 0, 0, 0))
    A = pc.dev(be, A0)
    y = be.from_numpy(np.full(100, 2.0))
    pc.kk.spmv("N", 1.0, A, be.from_numpy(np.ones(100)), 1.0, y)
    assert (be.to_numpy(y) == 2.0).all()      # example/wiki/sparse/KokkosSparse_wiki_spmv.cpp:66-95
    for A0 in (oracle.laplace2d("FD", 40, 30), oracle.laplace2d("FE", 25, 31), oracle.laplace3d("FD", 9, 8, 10),
               oracle.laplace3d("FE", 11, 9, 8)):
        for algo in ("SPMV_DEFAULT", "SPMV_FAST_SETUP"):
            pc.check_spmv(be, A0, "N", 1.0, 1.0, algo, max_val=32.0)


@pytest.mark.parametrize("orders", ["FF", "CC", "FC", "CF"])
def test_spmv_mv_layouts(be, orders):
    # Test_Sparse_spmv.hpp:1075-1092: layouts Left/Right/mixed, 1..30 vectors, incl. the 2x3 matrix
    A0 = oracle.random_crs(260, 240, 8, variance=5, seed=7)
    for nv in (1, 2, 3, 5, 8, 10, 16, 17, 30):
        pc.check_spmv_mv(be, A0, nv, "N", 2.5, -1.0, orders[0], orders[1])
    pc.check_spmv_mv(be, A0, 5, "T", 2.5, 0.0, orders[0], orders[1])
    pc.check_spmv_mv(be, A0, 4, "N", 0.0, 2.0, orders[0], orders[1])
    A23 = oracle.random_crs(2, 3, 2, seed=1)
    for nv in (1, 4, 16):
        pc.check_spmv_mv(be, A23, nv, "N", 1.0, 0.0, orders[0], orders[1], algo="SPMV_DEFAULT")


def test_mv_row_major_fast_path_and_packing(be):
    # row-major X with an even leading dimension -> 16-byte-per-lane kernel; column-major X with a handle -> packed copy
    A0 = oracle.random_crs(300, 280, 9, variance=6, seed=21)
    for nv in (2, 3, 4, 6, 7, 8, 12, 16, 17, 24):
        for orders in ("CC", "CF", "FC", "FF"):
            pc.check_spmv_mv(be, A0, nv, "N", 1.5, 0.5, orders[0], orders[1], algo="SPMV_DEFAULT")
            pc.check_spmv_mv(be, A0, nv, "N", 1.0, 0.0, orders[0], orders[1])
    rows = _custom([700, 0, 3, 300, 1, 1, 0, 40, 260, 255, 257, 5], 200, seed=9)   # rows longer than a 256-nnz staging window
    for orders in ("CC", "FF"):
        pc.check_spmv_mv(be, rows, 16, "N", 1.0, 1.0, orders[0], orders[1], algo="SPMV_DEFAULT")
        pc.check_spmv_mv(be, rows, 5, "N", 2.0, 0.0, orders[0], orders[1], algo="SPMV_DEFAULT")


def test_mv_long_rows_chunked_staging(be):
    A0 = _custom([5000, 0, 3, 2500, 1, 1, 0, 40], 300, seed=4)     # rows longer than the 2048-entry LDS chunk
    pc.check_spmv_mv(be, A0, 16, "N", 1.0, 1.0, "C", "C")
    pc.check_spmv_mv(be, A0, 3, "N", 1.0, 0.0, "F", "F")


@pytest.mark.parametrize("dims,st", [(d, 1) for d in pc.STRUCT_CASES_1D + pc.STRUCT_CASES_2D + pc.STRUCT_CASES_3D[:3]] +
                         [(d, 2) for d in pc.STRUCT_CASES_2D + pc.STRUCT_CASES_3D[2:]])
def test_spmv_struct_reference_cases(be, dims, st):
    pc.check_spmv_struct(be, dims, st)


def test_spmv_struct_strip_order(be):
    # strip order of the interior workgroups (knob struct_strip = lines per XCD strip): padded last block of lines, 2-D and 3-D
    def setk(v): pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(b"struct_strip", v))
    try:
        for strip in (1, 2, 8, 0):
            setk(strip)
            pc.check_spmv_struct(be, (12, 21, 5), 2)
            pc.check_spmv_struct(be, (12, 21, 5), 1, offset_dtype=np.int64)
            pc.check_spmv_struct(be, (40, 19), 2)
            pc.check_spmv_struct(be, (140, 5, 4), 2)                   # fewer lines than one block: the order is not applied
    finally:
        setk(0)


def test_spmv_struct_variants(be):
    pc.check_spmv_struct(be, (140, 5, 4), 2)                       # more than one 128-row chunk per grid line
    pc.check_spmv_struct(be, (131, 6), 1, offset_dtype=np.int64)
    pc.check_spmv_struct(be, (3, 3, 3), 2)                         # a single interior point
    pc.check_spmv_struct(be, (2, 7), 1)                            # no interior at all
    pc.check_spmv_struct(be, (12, 9, 7), 2, value_dtype=np.float32, vec_dtype=np.float32)
    pc.check_spmv_struct(be, (12, 9, 7), 1, value_dtype=np.float32)             # float matrix, double vectors
    pc.check_spmv_struct(be, (9, 8, 7), 2, mode="T"); pc.check_spmv_struct(be, (9, 8), 2, mode="H")
    pc.check_spmv_struct(be, (9, 8, 7), 2, mode="C", rank2=True)
    # interior rows that do not have exactly S entries: the per-row path through row_map must take over
    A0 = oracle.laplace2d("FE", 40, 6)
    rm = A0.row_map.copy(); ent = A0.entries; val = A0.values
    r = 1 * 40 + 7                                                  # an interior row: give it one extra explicit zero at the end
    ent2 = np.insert(ent, rm[r + 1], ent[rm[r + 1] - 1]); val2 = np.insert(val, rm[r + 1], 0.0); rm[r + 1:] += 1
    pc.check_spmv_struct(be, (40, 6), 2, A0=oracle.Crs(A0.nrows, A0.ncols, rm, ent2.astype(np.int32), val2))
    A = pc.dev(be, oracle.laplace2d("FD", 8, 8))
    x = be.from_numpy(np.ones(64)); y = be.from_numpy(np.zeros(64))
    with pytest.raises(pc.kk.KkamdError, match="does not match"):
        pc.kk.spmv_struct("N", 1, (8, 9), 1.0, A, x, 0.0, y)
    with pytest.raises(pc.kk.KkamdError, match="stencil_type"):
        pc.kk.spmv_struct("N", 3, (8, 8), 1.0, A, x, 0.0, y)
    with pytest.raises(RuntimeError, match="Dimensions do not match"):
        pc.kk.spmv_struct("N", 1, (8, 8), 1.0, A, be.from_numpy(np.ones(60)), 0.0, y)


def test_handle_less_calls_analyse_on_the_fly(be):
    """large matrices without an analysed handle (no handle, SPMV_FAST_SETUP) take the nnz-split kernel through a
    per-thread scratch plan; matrices of different sizes alternate to exercise the scratch growth"""
    pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(b"transient_min_knnz", 1))
    try:
        for A0 in (oracle.laplace2d("FE", 40, 30), oracle.laplace3d("FE", 14, 13, 12), pc._custom([0, 5000, 1, 0, 3, 900] * 3, 6000, seed=2) if hasattr(pc, "_custom") else oracle.laplace2d("FD", 50, 50)):
            for algo in (None, "SPMV_FAST_SETUP"):
                for alpha, beta in ((1.0, 0.0), (-2.0, 0.5)):
                    pc.check_spmv(be, A0, "N", alpha, beta, algo=algo, max_val=32.0)
        pc.check_spmv(be, oracle.laplace2d("FE", 40, 30), "N", 1.0, 1.0, algo=None, offset_dtype=np.int64, value_dtype=np.float32, vec_dtype=np.float32, max_val=32.0)
        pc.check_spmv(be, oracle.laplace2d("FE", 40, 30), "T", 1.0, 0.0, algo=None, max_val=32.0)
    finally:
        pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(b"transient_min_knnz", 10000))


def test_transposed_modes_through_cached_explicit_transpose(be):
    """modes T / H with an analysed handle: the plan caches A^T once and refreshes its values on every call"""
    pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(b"explicit_transpose_min_knnz", 0))
    try:
        for A0 in (oracle.laplace3d("FE", 9, 8, 7), oracle.random_crs(120, 75, 9, variance=6, seed=4), oracle.random_crs(40, 300, 30, variance=25, seed=5)):
            for mode in "TH":
                for alpha, beta in ((1.0, 0.0), (-1.5, 0.5)):
                    pc.check_spmv(be, A0, mode, alpha, beta, algo="SPMV_DEFAULT", knobs={"explicit_transpose": 1 + (alpha < 0)}, max_val=32.0)
        pc.check_spmv(be, oracle.laplace2d("FD", 30, 20), "T", 2.0, 1.0, algo="SPMV_DEFAULT", offset_dtype=np.int64, value_dtype=np.float32,
                      vec_dtype=np.float32, knobs={"explicit_transpose": 1}, max_val=8.0)
        # values change between calls on the same handle; the structure does not
        A0 = oracle.random_crs(60, 50, 7, variance=3, seed=6, sorted_rows=True)
        A = pc.dev(be, A0)
        h = pc.kk.SPMVHandle("SPMV_DEFAULT"); h.set("explicit_transpose", 1)
        rng = np.random.default_rng(2)
        x = rng.random(A0.nrows)
        for rep in range(3):
            y = be.from_numpy(np.zeros(A0.ncols))
            pc.kk.spmv(h, "T", 1.0, A, be.from_numpy(x), 0.0, y)
            exp = oracle.spmv_sequential("T", oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, be.to_numpy(A.values).copy()), 1.0, x, 0.0, np.zeros(A0.ncols))
            assert np.allclose(be.to_numpy(y), exp, rtol=1e-13, atol=1e-13)
            if rep == 0: A.values[:] = rng.random(A0.nnz) + rep             # in place: same device array, new numbers
            else: A.values[::3] = rng.random(len(A.values[::3])) - rep      # ... and only some of them
        # several fingerprint tiles (4096 values each); one value changes, then two swap places
        A0 = oracle.random_crs(1500, 1200, 9, variance=3, seed=7, sorted_rows=True)
        A = pc.dev(be, A0)
        h = pc.kk.SPMVHandle("SPMV_DEFAULT"); h.set("explicit_transpose", 1)
        x = rng.random(A0.nrows)
        v = A0.values.copy()
        for rep in range(4):
            y = be.from_numpy(np.zeros(A0.ncols))
            pc.kk.spmv(h, "T", 1.0, A, be.from_numpy(x), 0.0, y)
            exp = oracle.spmv_sequential("T", oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, v), 1.0, x, 0.0, np.zeros(A0.ncols))
            assert np.allclose(be.to_numpy(y), exp, rtol=1e-13, atol=1e-13), rep
            if rep == 0: v[9000] = -3.5
            elif rep == 1: v[[10, len(v) - 1]] = v[[len(v) - 1, 10]]
            A.values[:] = be.from_numpy(v)
        # the atomic kernel stays reachable
        pc.check_spmv(be, oracle.laplace3d("FE", 9, 8, 7), "T", 1.0, 0.0, algo="SPMV_DEFAULT", knobs={"explicit_transpose": 0}, max_val=32.0)
    finally:
        pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(b"explicit_transpose_min_knnz", 1000))


def test_error_behaviour(be):
    A0 = oracle.random_crs(20, 30, 3, seed=2)
    A = pc.dev(be, A0)
    with pytest.raises(RuntimeError, match="Dimensions do not match"):
        pc.kk.spmv("N", 1.0, A, be.from_numpy(np.ones(29)), 0.0, be.from_numpy(np.ones(20)))
    with pytest.raises(RuntimeError, match="Dimensions do not match \\(transpose\\)"):
        pc.kk.spmv("T", 1.0, A, be.from_numpy(np.ones(30)), 0.0, be.from_numpy(np.ones(30)))
    with pytest.raises(RuntimeError, match="Invalid transpose mode"):
        pc.kk.spmv("X", 1.0, A, be.from_numpy(np.ones(30)), 0.0, be.from_numpy(np.ones(20)))
    with pytest.raises(ValueError):
        pc.kk.SPMVHandle("SPMV_BSR_TC")
    # a handle is bound to one matrix for life (spmv_handle.hpp:273-277)
    h = pc.kk.SPMVHandle("SPMV_DEFAULT")
    pc.kk.spmv(h, "N", 1.0, A, be.from_numpy(np.ones(30)), 0.0, be.from_numpy(np.ones(20)))
    B = pc.dev(be, oracle.random_crs(20, 30, 4, seed=3))
    with pytest.raises(pc.kk.KkamdError) as ei:
        pc.kk.spmv(h, "N", 1.0, B, be.from_numpy(np.ones(30)), 0.0, be.from_numpy(np.ones(20)))
    assert ei.value.status == pc.kk._capi.ERR_STATE


def test_degenerate_dimensions(be):
    for nrows, ncols, per in ((0, 0, 0), (0, 7, 0), (7, 0, 0), (9, 9, 0)):
        A0 = oracle.random_crs(nrows, ncols, per, seed=1)
        pc.check_spmv(be, A0, "N", 1.0, 0.0, None, nans=True)
        pc.check_spmv(be, A0, "N", 1.0, 2.0, "SPMV_DEFAULT")
        pc.check_spmv(be, A0, "T", 1.0, 0.0, None, nans=True)


def test_tile_descriptors_against_merge_path(be):
    """The plan's tile -> row search (spmv_plan_kernel) against the closed form pinned by the merge-matrix vectors
    (tests/test_oracle.py::test_merge_path_split_equals_nnz_split_descriptors): tile b starts at nonzero p = b * tile; its
    descriptor is the first row starting at or after p, with bit 31 set when p falls inside a row -- i.e. (rows that end at or
    before p) + (1 if p is inside a row), which is what the reference's diagonal_search yields on the diagonal through p."""
    for lens in ([5000, 1, 0, 4100, 2048, 2048, 3], [0, 0, 5, 0, 0, 0, 7, 0, 2047, 1, 0, 0] * 40, [27] * 700, [0] * 30 + [1024] * 9 + [0] * 5):
        A0 = _custom(lens, 5000, seed=len(lens))
        h = pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs={"nnz_per_thread": 4})
        tile, tiles = h.query("tile"), h.query("tiles")
        assert tile == 1024 and tiles == -(-A0.nnz // tile)
        got = h.export("tile_first_row", tiles + 1).astype(np.int64)
        rm = A0.row_map
        for b in range(tiles + 1):
            if b == tiles:
                assert got[b] == A0.nrows
                continue
            p = b * tile
            first = int(np.searchsorted(rm, p, side="left"))            # first row whose start is >= p
            inside = rm[first] > p                                       # p cuts a row: that row started earlier
            ended = int(np.searchsorted(rm[1:], p, side="right"))       # merge path: rows consumed on the diagonal through p
            while ended < A0.nrows and rm[ended + 1] == rm[ended] and rm[ended] == p: ended += 1   # empty rows at p belong to neither side
            assert (got[b] & 0x7fffffff) == first and bool(got[b] >> 31 & 1) == bool(inside), (b, got[b], first, inside)
            assert first == ended + (1 if inside else 0) or rm[first] == p, (b, first, ended, inside)


def test_column_slab_copy(be):
    # kk_spmv_colslab.hip: forced (colslab = 2) with narrow slabs so that small matrices have many of them; the stable sort by slab,
    # the wave fold of equal rows, the tail tile, empty rows, duplicate entries and the int64 / float variants
    cases = [oracle.random_crs(3000, 3000, 9, variance=4, seed=3),                       # ~27 K entries: 7 sort tiles
             _custom([0, 0, 700, 1, 0, 5000, 3, 0, 0, 64, 65, 0], 4000, seed=4),        # rows longer than a tile, empty rows
             oracle.random_crs(500, 9000, 40, variance=10, seed=6)]                      # wide: more columns than rows
    r, e = cases[0].row_map, cases[0].entries.copy()
    e[r[10]:r[10] + 3] = e[r[10]]                                                        # duplicate columns inside a row
    cases.append(oracle.Crs(3000, 3000, r, e, cases[0].values))
    for A0 in cases:
        for shift in (4, 7, 12):
            kn = {"colslab": 2, "colslab_shift": shift}
            h = pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs=kn, max_val=50.0, expect={"colslab": 1, "colslab_shift": max(shift, int(np.ceil(np.log2(A0.ncols / 256))))})
            assert h.query("colslab_slabs") == -(-A0.ncols // (1 << h.query("colslab_shift")))
            pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs=kn, max_val=50.0, nans=True)
        pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"colslab": 2, "colslab_shift": 6}, max_val=50.0, offset_dtype=np.int64, value_dtype=np.float32)
        pc.check_spmv(be, A0, "N", 1.0, 1.0, "SPMV_DEFAULT", knobs={"colslab": 2, "colslab_shift": 6, "colslab_const": 1}, max_val=50.0, value_dtype=np.float32, vec_dtype=np.float32)
        pc.check_spmv(be, A0, "T", 1.0, 0.0, "SPMV_DEFAULT", knobs={"colslab": 2, "colslab_shift": 6}, max_val=50.0)          # mode T never takes the copy of A
    # the automatic mode does nothing on a small matrix, and nothing at all under the emulator (there is nothing to time)
    h = pc.check_spmv(be, cases[0], "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=50.0, knobs={"colslab": 1}, expect={"colslab": 0, "colslab_tried": 1})
    h = pc.check_spmv(be, cases[0], "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=50.0, expect={"colslab": 0, "colslab_tried": 1})       # the default handle applies its rule (deterministic form): a small matrix says no
    h = pc.check_spmv(be, cases[0], "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=50.0, knobs={"colslab": 0}, expect={"colslab": 0, "colslab_tried": 0})


def test_column_slab_follows_value_changes(be):
    # the copy holds its own values: a call fingerprints A.values tile by tile and moves the tiles that changed
    A0 = oracle.random_crs(2500, 2500, 10, variance=3, seed=8)
    rng = np.random.default_rng(1)
    x = rng.random(A0.ncols)
    A = pc.dev(be, A0)
    h = pc.kk.SPMVHandle("SPMV_DEFAULT"); h.set("colslab", 2); h.set("colslab_shift", 5)
    xd, yd = be.from_numpy(x), be.from_numpy(np.zeros(A0.nrows))
    def run_and_check(vals):
        pc.kk.spmv(h, "N", 1.0, A, xd, 0.0, yd)
        exp = oracle.spmv_serial("N", oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, vals), 1.0, x, 0.0, np.zeros(A0.nrows))
        np.testing.assert_allclose(be.to_numpy(yd), exp, rtol=1e-12, atol=1e-12)
    run_and_check(A0.values)
    assert h.query("colslab") == 1
    v = A0.values.copy()
    v[5000] = -7.25                                       # one value in one tile
    A.values[:] = be.from_numpy(v)
    run_and_check(v)
    v = v * 3.0 + 1.0                                     # every value
    A.values[:] = be.from_numpy(v)
    run_and_check(v)
    v[[0, len(v) - 1]] = v[[len(v) - 1, 0]]               # two values swapped (the plain sum of the bits does not move)
    A.values[:] = be.from_numpy(v)
    run_and_check(v)
