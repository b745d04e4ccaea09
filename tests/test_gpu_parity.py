"""GPU parity tests (`-m gpu`): the gfx950 build of libkkamd.so, called through its C ABI, against the
CPU oracle on the same seeded inputs; reference comparators and tolerances (see parity_cases.py)."""
import glob
import os

import numpy as np
import pytest

import oracle
import parity_cases as pc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def be():
    import ctypes as C
    b = pc.kk.torch_backend()
    name = C.create_string_buffer(256); g = C.c_int(); cus = C.c_int()
    pc.kk._capi.check(b.lib, b.lib.kkamd_device_info(name, 256, C.byref(g), C.byref(cus)))
    print("device:", name.value.decode(), "CUs:", cus.value)
    assert g.value == 1, "libkkamd.so is built for gfx950 only; found %s" % name.value.decode()
    return b


ALGOS = [None, "SPMV_DEFAULT", "SPMV_FAST_SETUP", "SPMV_NATIVE", "SPMV_MERGE_PATH", "SPMV_NATIVE_MERGE_PATH"]


@pytest.mark.parametrize("algo", ALGOS)
def test_spmv_heavy(be, algo):
    # Test_Sparse_spmv.hpp:1060-1068 "heavy": 1000 rows, 3..20 nnz/row, all modes, alpha/beta in {0,1,-1,2.5}
    for nnz_row, var in ((3, 2), (10, 5), (20, 15)):
        A0 = oracle.random_crs(1000, 1000, nnz_row, variance=var, seed=nnz_row)
        for mode in ("N", "C", "T", "H"):
            for alpha in (0.0, 1.0, -1.0, 2.5):
                for beta in (0.0, 1.0, -1.0, 2.5):
                    pc.check_spmv(be, A0, mode, alpha, beta, algo)
                    if beta == 0.0:
                        pc.check_spmv(be, A0, mode, alpha, beta, algo, nans=True)


@pytest.mark.parametrize("algo", ["SPMV_DEFAULT", "SPMV_FAST_SETUP"])
@pytest.mark.parametrize("n,nnz_row,var,bw", [(10000, 10, 5, 100), (50000, 10, 10, 500), (50000, 27, 0, None), (20000, 150, 140, None)])
def test_spmv_light(be, algo, n, nnz_row, var, bw):
    A0 = oracle.random_crs(n, n, nnz_row, variance=var, seed=n % 97, bandwidth=bw)
    for mode in ("N", "T"):
        for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (0.0, 1.0)):
            pc.check_spmv(be, A0, mode, alpha, beta, algo, nans=(beta == 0.0))


@pytest.mark.parametrize("npt", [4, 8, 16])
@pytest.mark.parametrize("remap", [1, 0])
def test_stream_kernel_variants(be, npt, remap):
    for nnz_row, var, n in ((27, 0, 40000), (3, 2, 100000), (700, 650, 2000), (1, 0, 50000)):
        A0 = oracle.random_crs(n, n + 13, nnz_row, variance=var, seed=npt + nnz_row)
        knobs = {"nnz_per_thread": npt, "xcd_remap": remap}
        pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs=knobs)
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs=knobs)


@pytest.mark.parametrize("lpr", [1, 2, 4, 8, 16, 32, 64])
def test_vector_kernel_variants(be, lpr):
    A0 = oracle.random_crs(30000, 29000, 27, variance=20, seed=lpr)
    pc.check_spmv(be, A0, "N", 1.0, 1.0, "SPMV_FAST_SETUP", knobs={"lanes_per_row": lpr})


@pytest.mark.parametrize("variant", [1, 6])
def test_stream_variants(be, variant):
    # the planned kernel with the default analysis (1) and with the column codes attempted whatever the size (6)
    mats = [oracle.laplace3d("FE", 60, 50, 40), oracle.random_crs(40000, 39000, 13, variance=9, seed=2), oracle.random_crs(30000, 30000, 25, variance=5, seed=3, bandwidth=40)]
    for A0 in mats:
        for npt in (4, 8, 16):
            pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs={"nnz_per_thread": npt, "stream_variant": variant}, max_val=32.0)
            pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs={"nnz_per_thread": npt, "stream_variant": variant}, max_val=32.0)


def test_window_codes(be):
    # 16-bit window codes for the columns: forced (stream_variant 6), picked by the default analysis, and the fall-back
    for name, A0, ok in pc.window_code_cases():
        for npt in (4, 8, 16):
            pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs={"nnz_per_thread": npt, "stream_variant": 6}, max_val=32.0,
                          expect={"window_codes": ok})
        pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, knobs={"window_codes_min_knnz": 0}, max_val=32.0,
                      offset_dtype=np.int64, value_dtype=np.float32, expect={"window_codes": ok})
    A0 = oracle.laplace3d("FE", 60, 50, 40)            # 3.1e6 nnz: above the default threshold
    pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", nans=True, max_val=32.0, expect={"window_codes": 1, "tile": 2048})
    pc.check_spmv(be, A0, "N", 1.0, 1.0, "SPMV_DEFAULT", knobs={"window_codes": 0}, max_val=32.0, expect={"window_codes": 0})
    pc.check_spmv(be, A0, "T", 1.0, 1.0, "SPMV_DEFAULT", knobs={"explicit_transpose": 1}, max_val=32.0)
    A0 = oracle.random_crs(200000, 3000000, 11, variance=4, seed=9)   # 2.2e6 nnz, no structure: plain entries
    pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=1.0, expect={"window_codes": 0})


def test_pattern_codes(be):
    # row-pattern records instead of per-nonzero codes (staged-x kernel): forced on for every tile that has one
    for name, A0, npt, expect_pat in pc.pattern_code_cases():
        kn = {"window_codes_min_knnz": 0, "nnz_per_thread": npt, "pattern_codes": 2}
        for odt, vdt, beta in ((np.int32, None, 0.5), (np.int64, None, 0.0), (np.int32, np.float32, 0.0)):
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
    # ... and at a size where a plan is worth timing: 27-pt 120^3 (46.7e6 nonzeros: 2048-nnz tiles by the automatic rule)
    kk = pc.kk
    A = kk.laplace_matrix("FE", 120, 120, 120)
    ys = []
    for direct in (1, 0):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("pattern_direct", direct)
        x = be.from_numpy(np.random.default_rng(3).random(A.numCols())); y = be.from_numpy(np.zeros(A.numRows()))
        kk.spmv(h, "N", 1.0, A, x, 0.0, y)
        ys.append(be.to_numpy(y).copy())
        assert h.query("pattern_direct") == direct and h.query("pattern_tiles") >= 0.99 * h.query("tiles"), (direct, h.query("pattern_tiles"), h.query("tiles"))
    assert np.array_equal(ys[0], ys[1])


def test_mixed_tiles(be):
    # the column analysis is per tile: tiles the windows cannot cover read entries, the others keep codes / staged x / records
    for name, A0 in pc.mixed_tile_cases():
        for npt in (4, 8, 16):
            for pat in (0, 2):
                kn = {"window_codes_min_knnz": 0, "nnz_per_thread": npt, "pattern_codes": pat, "window_codes_min_pct": 10}
                h = pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs=kn, max_val=32.0, expect={"window_codes": 1})
                assert 0 < h.query("plain_tiles") < h.query("tiles"), (name, npt, h.query("plain_tiles"), h.query("tiles"))
                pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs=kn, max_val=32.0, nans=True, offset_dtype=np.int64, value_dtype=np.float32)


def test_knob_validation(be):
    A0 = oracle.random_crs(3000, 3000, 9, seed=1)
    A = pc.dev(be, A0)
    x, y = be.from_numpy(np.ones(3000)), be.from_numpy(np.zeros(3000))
    for key, val in (("xcd_remap", 3), ("xcd_remap", 6), ("mv_remap", 12), ("nnz_per_thread", 5), ("stream_variant", 2), ("ablate", 1),
                     ("lds_pad_kb", 8), ("nontemporal", 1), ("kernel", 7)):
        h = pc.kk.SPMVHandle("SPMV_DEFAULT"); h.set(key, val)
        with pytest.raises(pc.kk.KkamdError):
            pc.kk.spmv(h, "N", 1.0, A, x, 0.0, y)
    with pytest.raises(pc.kk.KkamdError):
        from kokkos_kernels_amd._capi import check
        check(be.lib, be.lib.kkamd_set_default(b"spgemm_debug", 1))     # measurement-build knob: not in libkkamd.so


def test_handle_stream_change(be):
    # a handle used on a second stream: the plan's scratch (carries) is fenced on the old stream first
    # (TPL_SpMV_Data::set_exec_space, sparse/src/KokkosSparse_spmv_handle.hpp:95-104)
    import torch
    A0 = oracle.laplace3d("FE", 50, 40, 30)
    A = pc.dev(be, A0)
    rng = np.random.default_rng(2)
    xs = [rng.random(A0.ncols) for _ in range(4)]
    h = pc.kk.SPMVHandle("SPMV_DEFAULT")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for i, x in enumerate(xs):
        with torch.cuda.stream(streams[i % 2]):
            xd = torch.from_numpy(x).cuda(); yd = torch.full((A0.nrows,), float("nan"), dtype=torch.float64, device="cuda")
            pc.kk.spmv(h, "N", 1.0, A, xd, 0.0, yd)
            outs.append((xd, yd))
    torch.cuda.synchronize()
    tol = oracle.spmv_max_error(A0, 1.0, 0.0, max_val=32.0)
    for x, (xd, yd) in zip(xs, outs):
        assert np.abs(yd.cpu().numpy() - oracle.spmv_serial("N", A0, 1.0, x, 0.0, np.zeros(A0.nrows))).max() <= tol


def test_mv3_lds_staged(be):
    # rank-2 kernel over LDS-staged X tiles: staged and gather tiles, 8 / 16 / 24 / 32 right-hand sides, both layouts, every tile order
    for name, A0, staged in pc.mv3_cases():
        for nvec, xo, yo, alpha, beta in ((16, "C", "C", 1.5, 0.5), (8, "C", "C", 1.0, 0.0), (32, "C", "C", -1.0, 0.0), (24, "C", "F", 1.0, 1.0),
                                          (16, "F", "F", 2.0, 0.0), (16, "F", "C", 1.0, -1.0)):
            h = pc.check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", knobs={"mv_kernel": 3}, max_val=32.0,
                                 nans=(beta == 0.0))
            assert (h.query("mv_staged_tiles") > 0) == staged, (name, nvec, h.query("mv_staged_tiles"), h.query("mv_tiles"))
    # a grid large enough for the strip order to engage (far stride 160 * 120 rows * 128 B * 3 > 3 MB)
    A0 = oracle.laplace3d("FE", 160, 120, 12)
    for order in (0, 1, 2):
        h = pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", knobs={"mv_kernel": 3, "mv_order": order}, max_val=32.0, nans=True)
        assert h.query("mv_tiles") > 0 and h.query("mv_order") == order, (order, h.query("mv_order"))
    # the wave-private gather kernel takes its row blocks in strip order on the same grid, for every width
    for nvec, xo, yo in ((16, "C", "C"), (8, "C", "C"), (4, "C", "C"), (16, "F", "F"), (3, "C", "C")):
        h = pc.check_spmv_mv(be, A0, nvec, "N", 1.5, 0.0, xo, yo, algo="SPMV_DEFAULT", knobs={"mv_kernel": 2}, max_val=32.0, nans=True)
        assert h.query("mv_tiles") == 0 and h.query("mv_period") == 160 * 120, (nvec, h.query("mv_period"))
        assert h.query("mv_order") == (2 if nvec >= 8 else h.query("mv_order")), (nvec, h.query("mv_order"))
    for order in (0, 1):
        pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", knobs={"mv_kernel": 2, "mv_order": order}, max_val=32.0, expect={"mv_order": 0})


def test_mv4_plane_marching(be):
    # rank-2 plane-marching kernel (default for analysed handles, fp64, right-hand sides in blocks of 16, lattice stencils):
    # interior and truncated boundary rows, rows left to the gather kernel, both layouts, alpha / beta, 64-bit offsets, fp32 values
    for name, A0, left in pc.mv4_cases():
        for nvec, xo, yo, alpha, beta, off in ((16, "C", "C", 1.5, 0.5, np.int32), (16, "C", "C", 1.0, 0.0, np.int64), (32, "C", "C", -1.0, 0.0, np.int32),
                                               (16, "F", "F", 2.0, 0.0, np.int32), (16, "F", "C", 1.0, -1.0, np.int32), (48, "C", "F", 1.0, 1.0, np.int32),
                                               (21, "C", "C", 0.5, 0.0, np.int32), (37, "F", "F", 1.0, 2.0, np.int32)):      # 16 + 5, 32 + 5: the remainder is a partial block
            h = pc.check_spmv_mv(be, A0, nvec, "N", alpha, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0), offset_dtype=off)
            assert h.query("mv4_workgroups") > 0, (name, nvec)
            if left is not None:
                assert h.query("mv4_other_rows") == left, (name, h.query("mv4_other_rows"))
    # grids with many patches and k-chunks; the chunking knob from one chunk to nz / 4
    for A0 in (oracle.laplace3d("FE", 160, 120, 12), oracle.laplace3d("FD", 70, 50, 90)):
        for wg in (8, 1, 64):
            for alpha, beta in ((1.0, 0.0), (0.5, 2.0)):
                h = pc.check_spmv_mv(be, A0, 16, "N", alpha, beta, "C", "C", algo="SPMV_DEFAULT", knobs={"mv4_wg_per_cu": wg}, max_val=32.0, nans=(beta == 0.0))
                assert h.query("mv4_workgroups") > 0 and h.query("mv4_other_rows") < 0.02 * A0.nrows
    name, A0, _ = pc.mv4_cases()[0]
    pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.5, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, value_dtype=np.float32)
    # column-major X takes the column-wise piece order with swizzled slab rows by default; mv4_xcol 0 keeps the general-stride order
    for xcol in (0, 1):
        for name_, A1, _ in pc.mv4_cases()[:4]:
            h = pc.check_spmv_mv(be, A1, 32, "N", 1.5, 0.0, "F", "F", algo="SPMV_DEFAULT", knobs={"mv4_xcol": xcol}, max_val=32.0, nans=True)
            assert h.query("mv4_workgroups") > 0
            pc.check_spmv_mv(be, A1, 16, "N", -1.0, 0.5, "F", "C", algo="SPMV_DEFAULT", knobs={"mv4_xcol": xcol}, max_val=32.0)
    # 2-D lattices: the lines are grouped m at a time into "planes"; the first and last line of every group go to the gather rows
    for st, nxl, nyl, m_ in (("FE", 70, 128, 32), ("FD", 40, 256, 64), ("FE", 500, 400, 100)):
        A2 = oracle.laplace2d(st, nxl, nyl)
        for nvec, xo, yo, beta in ((16, "C", "C", 0.0), (32, "F", "F", 0.5), (5, "C", "F", 0.0), (12, "F", "F", -1.0)):
            h = pc.check_spmv_mv(be, A2, nvec, "N", 1.5, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0))
            assert h.query("mv4_workgroups") > 0 and h.query("mv4_other_rows") == 2 * (nyl // m_ - 1) * nxl, (st, h.query("mv4_workgroups"), h.query("mv4_other_rows"))
        h = pc.check_spmv_mv(be, A2, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", knobs={"mv4_2d": 0}, max_val=32.0)
        assert h.query("mv4_workgroups") == 0
    # Inf and NaN in X reach exactly the rows the reference lets them reach (no 0 * Inf from halo or pad entries)
    name, A0, _ = pc.mv4_cases()[3]
    h = pc.check_spmv_mv(be, A0, 16, "N", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, nans=True,
                         x_special={0: np.inf, 33 * 6 * 10 + 5: -np.inf, A0.nrows - 1: np.nan, 33 * 6 * 7 + 33 * 2 + 16: np.inf})
    assert h.query("mv4_workgroups") > 0
    # narrow multivectors and remainders run the partial-block form (columns past the block clamped on the X side, masked on the Y side):
    # every width from 4 to 15, both layouts, beta != 0 over the masked columns' neighbours
    for nvec in (4, 5, 8, 11, 12, 15):
        for xo, yo, beta in (("C", "C", 0.0), ("F", "F", 0.5), ("C", "F", -1.0)):
            h = pc.check_spmv_mv(be, A0, nvec, "N", 1.5, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0))
            assert h.query("mv4_workgroups") > 0, (nvec, xo, yo)
    # not its matrices / widths: no far stride (2-D), too few lattice rows, 3 right-hand sides (below mv4_min_nvec), no analysis, the gather kernel asked for
    for A1, nvec, algo, knobs in ((oracle.laplace2d("FE", 130, 41), 16, "SPMV_DEFAULT", None), (oracle.laplace3d("FE", 12, 12, 12), 16, "SPMV_DEFAULT", None),
                                  (A0, 3, "SPMV_DEFAULT", None), (A0, 8, "SPMV_DEFAULT", {"mv4_min_nvec": 16}), (A0, 16, "SPMV_FAST_SETUP", None), (A0, 16, "SPMV_DEFAULT", {"mv_kernel": 2}),
                                  (oracle.random_crs(5000, 5000, 9, variance=3, seed=5), 16, "SPMV_DEFAULT", None)):
        h = pc.check_spmv_mv(be, A1, nvec, "N", 1.0, 0.0, "C", "C", algo=algo, knobs=knobs, max_val=32.0)
        assert h.query("mv4_workgroups") == 0


def test_mv5_matrix_core(be):
    # rank-2 matrix-core kernel (v_mfma_f64_16x16x4 over 16-row tiles whose rows share columns, kk_spmv_mvblk.hip): every case of
    # pc.mv5_cases() x widths 16 / 5 / 37 / 32 x the four layout pairs, described tiles and gather rows as expected, Inf / NaN in X,
    # fp32 values, 64-bit offsets, forced on sparse matrices, off
    pc.check_mv5(be, light=False)


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
    pc.check_mv_transpose_cached(be)


def test_mv6_nonzero_split(be):
    # rank-2 nonzero-split kernel (kk_spmv_mvnnz.hip): chunks of 128 entries per 16-lane group, cut rows finished from carries, empty
    # rows from the plan's list; every width / layout pair, beta = 0 over NaNs, 64-bit offsets, fp32 values, Inf / NaN in X
    pc.check_mv6(be, light=False)


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
    # every width from 17 to 47 on a lattice matrix: full passes of 16 columns + one partial pass, row-major and column-major multivectors in turn,
    # beta = 0 over NaNs and beta != 0 (round-4 review item 8 asks for the parity of these widths whatever their speed)
    name, A0, _ = pc.mv4_cases()[0]
    for nvec in range(17, 48):
        xo, yo = (("C", "C"), ("F", "F"), ("C", "F"))[nvec % 3]
        beta = 0.0 if nvec % 2 else 0.5
        h = pc.check_spmv_mv(be, A0, nvec, "N", 1.5, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0), seed=nvec)
        assert h.query("mv4_workgroups") > 0, nvec
    # the other lattice matrices (truncated stencils, rows left to the gather rows, a 2-D lattice), wider multivectors: up to eight blocks
    # share a launch (128 columns), beyond that the blocks go eight at a time; 64-bit offsets; mode T through the cached transpose
    cases = [c[:2] for c in pc.mv4_cases()[1:4]] + [("9-pt 70 x 128", oracle.laplace2d("FE", 70, 128))]
    for ci, (name, A1) in enumerate(cases):
        for nvec in (18, 24, 33, 40, 64, 100, 128, 129, 136, 150):
            xo, yo = (("C", "C"), ("F", "F"), ("F", "C"))[(nvec + ci) % 3]
            beta = 0.5 if (nvec + ci) % 2 else 0.0
            h = pc.check_spmv_mv(be, A1, nvec, "N", -1.0, beta, xo, yo, algo="SPMV_DEFAULT", max_val=32.0, nans=(beta == 0.0), seed=nvec + 7 * ci,
                                 offset_dtype=np.int64 if nvec % 3 == 0 else np.int32)
            assert h.query("mv4_workgroups") > 0, (name, nvec)
    pc.check_spmv_mv(be, A0, 40, "T", 1.0, 0.0, "C", "C", algo="SPMV_DEFAULT", max_val=32.0, knobs={"explicit_transpose_min_knnz": 0})
    pc.check_spmv_mv(be, A0, 24, "T", 2.0, 0.5, "F", "F", algo="SPMV_DEFAULT", max_val=32.0, knobs={"explicit_transpose_min_knnz": 0})


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
    # rank 1 on the plane-marching analysis (knob march, off by default): lattice matrices incl. boundary and broken rows
    for name, A0, left in pc.mv4_cases() + [("27pt 160x120x12", oracle.laplace3d("FE", 160, 120, 12), None), ("7pt 70x50x90", oracle.laplace3d("FD", 70, 50, 90), None)]:
        for alpha, beta, off, planes in ((1.0, 0.0, np.int32, 20), (1.5, -0.5, np.int64, 3), (2.0, 1.0, np.int32, 1000)):
            h = pc.check_spmv(be, A0, "N", alpha, beta, "SPMV_DEFAULT", nans=(beta == 0.0), offset_dtype=off, max_val=32.0,
                              knobs={"march": 1, "march_planes": planes})
            assert h.query("march_workgroups") > 0, name
    name, A0, _ = pc.mv4_cases()[0]
    pc.check_spmv(be, A0, "N", 1.0, 0.5, "SPMV_DEFAULT", max_val=32.0, knobs={"march": 1}, value_dtype=np.float32)
    h = pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=32.0)
    assert h.query("march_workgroups") == 0


def test_xcd_group_orders(be):
    # grouped tile orders (xcd_remap / mv_remap = G): whole blocks of 8G tiles are permuted, the incomplete last block is not
    for nrows in (64 * 255 + 5, 64 * 256, 64 * 257 + 1, 64 * 1030):
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


@pytest.mark.parametrize("algo", ["SPMV_DEFAULT", "SPMV_FAST_SETUP", None])
def test_spmv_row_shapes(be, algo):
    cases = {
        "one_row_many_tiles": [200000],
        "long_rows_span_tiles": [50000, 1, 0, 4100, 2048, 2048, 3, 100000],
        "empty_rows_everywhere": [0, 0, 5, 0, 0, 0, 7, 0, 2047, 1, 0, 0] * 500,
        "all_empty_but_one": [0] * 30000 + [4] + [0] * 30000,
        "exact_tile_multiple": [1024, 1024, 2048, 0, 0] * 64,
        "leading_trailing_empty": [0] * 5000 + [30] * 20000 + [0] * 7000,
        "single_entry": [1],
    }
    for name, lens in cases.items():
        A0 = _custom(lens, 9770, seed=len(lens))
        for beta in (0.0, 2.0):
            pc.check_spmv(be, A0, "N", -1.5, beta, algo, nans=(beta == 0.0))


def test_spmv_types(be):
    A0 = oracle.random_crs(20000, 19000, 12, variance=7, seed=3)
    pc.check_spmv(be, A0, "N", 1.0, 1.0, "SPMV_DEFAULT", offset_dtype=np.int64)
    pc.check_spmv(be, A0, "N", 1.0, 1.0, None, offset_dtype=np.int64)
    pc.check_spmv(be, A0, "T", 1.0, 1.0, None, offset_dtype=np.int64)
    pc.check_spmv(be, A0, "N", 2.0, 0.5, "SPMV_DEFAULT", value_dtype=np.float32, vec_dtype=np.float32)
    pc.check_spmv(be, A0, "N", 2.0, 0.5, None, value_dtype=np.float32, vec_dtype=np.float32)
    pc.check_spmv(be, A0, "N", 2.0, 0.5, "SPMV_DEFAULT", value_dtype=np.float32)


def test_github_issue_101(be):
    import torch
    expected = 1.0 + pc.EPS_F / 2.0
    for vdt in (np.float64, np.float32):
        A = pc.kk.CrsMatrix.from_host(1, 2, [0, 2], [0, 1], np.array([1.0, pc.EPS_F / 2.0], dtype=vdt), backend=be)
        for h in (None, pc.kk.SPMVHandle("SPMV_DEFAULT")):
            y = torch.zeros(1, dtype=torch.float64, device="cuda")
            args = ("N", 1.0, A, torch.ones(2, dtype=torch.float64, device="cuda"), 0.0, y)
            pc.kk.spmv(*args) if h is None else pc.kk.spmv(h, *args)
            assert y.item() == expected
        for nv in range(1, 23):
            for left in (True, False):
                X = torch.ones((nv, 2) if left else (2, nv), dtype=torch.float64, device="cuda")
                Y = torch.zeros((nv, 1) if left else (1, nv), dtype=torch.float64, device="cuda")
                if left:
                    X, Y = X.t(), Y.t()
                pc.kk.spmv("N", 1.0, A, X, 0.0, Y)
                assert (Y == expected).all().item()


def test_wiki_example_and_structured(be):
    import torch
    A = pc.dev(be, oracle.laplace2d("FD", 10, 10, bc=(0, 0, 0, 0)))
    y = torch.full((100,), 2.0, dtype=torch.float64, device="cuda")
    pc.kk.spmv("N", 1.0, A, torch.ones(100, dtype=torch.float64, device="cuda"), 1.0, y)
    assert (y == 2.0).all().item()
    for A0 in (oracle.laplace2d("FD", 300, 200), oracle.laplace2d("FE", 250, 310), oracle.laplace3d("FD", 40, 30, 50),
               oracle.laplace3d("FE", 41, 39, 38)):
        for algo in ("SPMV_DEFAULT", "SPMV_FAST_SETUP"):
            pc.check_spmv(be, A0, "N", 1.0, 1.0, algo, max_val=32.0)


@pytest.mark.parametrize("orders", ["FF", "CC", "FC", "CF"])
def test_spmv_mv_layouts(be, orders):
    A0 = oracle.random_crs(6000, 5500, 10, variance=8, seed=7)
    for nv in list(range(1, 31)):
        pc.check_spmv_mv(be, A0, nv, "N", 2.5, -1.0, orders[0], orders[1])
    for nv in (1, 5, 10):
        pc.check_spmv_mv(be, A0, nv, "T", 2.5, 0.0, orders[0], orders[1])
        pc.check_spmv_mv(be, A0, nv, "N", 0.0, 2.0, orders[0], orders[1])
        pc.check_spmv_mv(be, A0, nv, "N", 1.0, 0.0, orders[0], orders[1], algo="SPMV_DEFAULT")
    A23 = oracle.random_crs(2, 3, 2, seed=1)
    for nv in (1, 4, 16):
        pc.check_spmv_mv(be, A23, nv, "N", 1.0, 0.0, orders[0], orders[1])
    long_rows = _custom([50000, 0, 3, 2500, 1, 1, 0, 40] * 3, 3000, seed=4)
    pc.check_spmv_mv(be, long_rows, 16, "N", 1.0, 1.0, orders[0], orders[1])


def test_mv_row_major_fast_path_and_packing(be):
    A0 = oracle.random_crs(8000, 7000, 12, variance=9, seed=21)
    for nv in (2, 3, 4, 6, 7, 8, 12, 16, 17, 24):
        for orders in ("CC", "CF", "FC", "FF"):
            pc.check_spmv_mv(be, A0, nv, "N", 1.5, 0.5, orders[0], orders[1], algo="SPMV_DEFAULT")
    rows = _custom([7000, 0, 3, 300, 1, 1, 0, 40, 260, 255, 257, 5] * 20, 2000, seed=9)
    for orders in ("CC", "FF"):
        pc.check_spmv_mv(be, rows, 16, "N", 1.0, 1.0, orders[0], orders[1], algo="SPMV_DEFAULT")
    L = oracle.laplace3d("FE", 30, 31, 29)
    for orders in ("CC", "FF"):
        pc.check_spmv_mv(be, L, 16, "N", 1.0, 0.0, orders[0], orders[1], algo="SPMV_DEFAULT")


@pytest.mark.parametrize("dims,st", [(d, 1) for d in pc.STRUCT_CASES_1D + pc.STRUCT_CASES_2D + pc.STRUCT_CASES_3D] +
                         [(d, 2) for d in pc.STRUCT_CASES_2D + pc.STRUCT_CASES_3D])
def test_spmv_struct_reference_cases(be, dims, st):
    pc.check_spmv_struct(be, dims, st)


def test_spmv_struct_strip_order(be):
    # strip order of the interior workgroups (knob struct_strip = lines per XCD strip): padded last block of lines, 2-D and 3-D
    def setk(v): pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(b"struct_strip", v))
    try:
        for strip in (1, 2, 8, 0):
            setk(strip)
            pc.check_spmv_struct(be, (70, 37, 21), 2)
            pc.check_spmv_struct(be, (70, 37, 21), 1, offset_dtype=np.int64)
            pc.check_spmv_struct(be, (300, 90), 2)
            pc.check_spmv_struct(be, (140, 5, 4), 2)                   # fewer lines than one block: the order is not applied
    finally:
        setk(0)


def test_spmv_struct_variants(be):
    pc.check_spmv_struct(be, (300, 40, 7), 2)                      # three 128-row chunks per grid line
    pc.check_spmv_struct(be, (515, 33), 1, offset_dtype=np.int64)
    pc.check_spmv_struct(be, (1000,), 1)
    pc.check_spmv_struct(be, (3, 3, 3), 2); pc.check_spmv_struct(be, (2, 7), 1)
    pc.check_spmv_struct(be, (40, 30, 20), 2, value_dtype=np.float32, vec_dtype=np.float32)
    pc.check_spmv_struct(be, (40, 30, 20), 1, value_dtype=np.float32)
    pc.check_spmv_struct(be, (30, 20, 10), 2, mode="T"); pc.check_spmv_struct(be, (30, 20), 2, mode="H")
    pc.check_spmv_struct(be, (30, 20, 10), 2, mode="C", rank2=True)
    A0 = oracle.laplace2d("FE", 200, 6)                            # an interior row with one extra entry: row_map path
    rm = A0.row_map.copy(); r = 1 * 200 + 77
    ent2 = np.insert(A0.entries, rm[r + 1], A0.entries[rm[r + 1] - 1]); val2 = np.insert(A0.values, rm[r + 1], 0.0); rm[r + 1:] += 1
    pc.check_spmv_struct(be, (200, 6), 2, A0=oracle.Crs(A0.nrows, A0.ncols, rm, ent2.astype(np.int32), val2))


def test_row_range_views_with_unaligned_offsets(be):
    """The multi-GPU overlap path (dist.py) runs planned SpMVs on zero-copy row-range views of a slab: rebased row_map,
    entries / values pointers offset by an arbitrary (odd) number of elements, y offset too."""
    import torch
    n = 48
    A = pc.kk.laplace_matrix("FE", n, n, n)
    rm = A.graph.row_map
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda", generator=g)
    y_full = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
    pc.kk.spmv(pc.kk.SPMVHandle("SPMV_DEFAULT"), "N", 1.0, A, x, 0.0, y_full)
    tol = 10 * np.finfo(np.float64).eps * 27 * 26
    tried_odd = False
    odd = torch.nonzero(rm[n * n:2 * n * n] % 2 == 1).flatten()            # a row whose first entry sits at an odd offset
    a_odd = n * n + int(odd[0].item())
    for a, b in ((a_odd, n * n * (n - 1) - 3), (1, 7 * n * n + 5), (n * n * (n - 1) + 2, n ** 3)):
        p0, p1 = int(rm[a].item()), int(rm[b].item())
        tried_odd |= (p0 % 2 == 1)
        sub = pc.kk.CrsMatrix(b - a, A.numCols(), (rm[a:b + 1] - rm[a]).contiguous(), A.graph.entries[p0:p1], A.values[p0:p1], backend=be)
        y = torch.full((A.numRows(),), 7.0, dtype=torch.float64, device="cuda")
        h = pc.kk.SPMVHandle("SPMV_DEFAULT")
        pc.kk.spmv(h, "N", 1.0, sub, x, 0.0, y[a:b])
        pc.kk.spmv(h, "N", 2.0, sub, x, -1.0, y[a:b])          # y := 2Ax - Ax
        assert (y[a:b] - y_full[a:b]).abs().max().item() <= 3 * tol
        assert (y[:a] == 7.0).all().item() and (y[b:] == 7.0).all().item()
    assert tried_odd


def test_dist_operator_on_device_with_loopback_transport(be):
    """The C implementation of the row-partitioned SpMV (kkamd_dist_spmv_*) on the GPU, one process playing rank 1 of 3 of a
    20 x 20 x 30 grid: a loop-back kkamd_transport_t answers the collectives from the test's own copy of the global x, so the
    halo lists, the interior / boundary split, the communication stream + events and the zero-copy x window all run on real
    device pointers (the world-2 gloo test covers the protocol between processes; RCCL itself needs a second GPU)."""
    import ctypes as C
    import torch
    from kokkos_kernels_amd import _capi
    from kokkos_kernels_amd.dist import DistSpmv, _DeviceView
    nx, ny, nz, world, rank = 20, 20, 30, 3, 1
    rows = nx * ny * (nz // world); n = nx * ny * nz; plane = nx * ny
    offsets = [r * rows for r in range(world + 1)]
    A = pc.kk.laplace_matrix("FE", nx, ny, nz, rows=(rank * rows, rows))
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    ranges = [(max(0, offsets[p] - plane), min(n, offsets[p + 1] + plane) - 1) for p in range(world)]     # column range of every slab

    from dist_loopback import Loopback
    tr = Loopback(x, offsets, rank, ranges)
    op = DistSpmv(A, offsets, rank, transport=tr)
    assert op.exchange_mode == "halo" and op.exchange_bytes == 2 * plane * 8 and op.query("parts") == 3
    assert rows - 2 * plane - 16 <= op.interior_rows <= rows - 2 * plane
    p_full = C.c_void_p(); pc.kk._capi.check(be.lib, be.lib.kkamd_dist_spmv_x_local(op._op, None, C.byref(p_full)))
    tr.base = p_full.value
    y_ref = torch.zeros(rows, dtype=torch.float64, device="cuda")
    pc.kk.spmv(pc.kk.SPMVHandle("SPMV_DEFAULT"), "N", 1.0, A, x, 0.0, y_ref)
    tol = 10 * np.finfo(np.float64).eps * 27 * 32
    xl = op.x_local(); xl.copy_(x[offsets[rank]:offsets[rank + 1]])          # x kept in the operator's window: no copy in apply()
    for rep in range(2):
        y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        op.apply(1.0, xl, 0.0, y)
        assert (y - y_ref).abs().max().item() <= tol
    y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    op.apply(1.0, x[offsets[rank]:offsets[rank + 1]].clone(), 0.0, y)           # a foreign x shard is copied in
    assert (y - y_ref).abs().max().item() <= tol
    # one plane from each neighbour, one plane to each neighbour
    assert sorted(set(tr.log)) == sorted([("recv", 0, offsets[1] - plane, plane), ("recv", 2, offsets[2], plane),
                                          ("send", 0, offsets[1], plane), ("send", 2, offsets[2] - plane, plane)]), tr.log[:8]
    del op


def test_builtin_rccl_transport_on_one_rank(be):
    """The library's OWN RCCL transport (kk::Rccl in kk_dist.hip: nine nccl* entry points bound with dlsym and called through
    hand-declared signatures) executed on the one GPU of the box -- no kkamd_transport_t anywhere:
      * kkamd_dist_transport_selftest: ncclGetUniqueId, ncclCommInitRank(nranks = 1), ncclAllGather in place and out of place, a group
        of ncclSend / ncclRecv to self, an empty group, ncclCommDestroy -- payload compared byte for byte;
      * kkamd_dist_spmv_* with world = 1 and every exchange forced through the one-rank communicator: the all-gather is the in-place
        ncclAllGather of the one shard, the halos are groups without peers, the peer-to-peer all-gather runs its two barrier
        collectives -- y against oracle.spmv_serial, x changing between the calls.
    SURVEY 8(e): the reference has no distributed layer to compare with; north_star's RCCL all-gather of x is the contract."""
    import torch
    import oracle
    from kokkos_kernels_amd.dist import DistSpmv, transport_selftest
    for nbytes in (8, 4097, 1 << 22):
        transport_selftest(be, nbytes)
    A0 = oracle.laplace3d("FE", 24, 20, 16)
    n = A0.nrows
    A = pc.kk.CrsMatrix.from_host(n, n, A0.row_map, A0.entries, A0.values, backend=be)
    rng = np.random.default_rng(11)
    x = rng.random(n); y0 = rng.random(n)
    tol = 10 * np.finfo(np.float64).eps * 27 * 32
    for mode in ("allgather", "halo", "halo_set", "allgather_p2p"):
        op = DistSpmv(A, [0, n], 0, exchange=mode, transport="rccl")
        assert op.exchange_mode == mode, (mode, op.exchange_mode)
        for scale, alpha, beta in ((1.0, 2.0, 0.5), (3.0, 1.0, 0.0), (-1.0, 1.0, 1.0)):
            xs = torch.from_numpy(scale * x).cuda(); ys = torch.from_numpy(y0.copy()).cuda()
            op.apply(alpha, xs, beta, ys)
            torch.cuda.synchronize()
            exp = oracle.spmv_serial("N", A0, alpha, scale * x, beta, y0.copy())
            assert float(np.abs(ys.cpu().numpy() - exp).max()) <= tol * 3, mode
        xl = op.x_local(); xl.copy_(torch.from_numpy(x).cuda())       # x kept in the operator's window: ncclAllGather strictly in place
        ys = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
        op.apply(1.0, xl, 0.0, ys); torch.cuda.synchronize()
        assert float(np.abs(ys.cpu().numpy() - oracle.spmv_serial("N", A0, 1.0, x, 0.0, y0.copy())).max()) <= tol, mode
        del op
    # auto at world = 1 stays local (no communicator is formed)
    op = DistSpmv(A, [0, n], 0, exchange="auto", transport="rccl")
    assert op.exchange_mode == "local"
    del op


def test_spgemm_structure_kept_by_the_symbolic_phase(be):
    pc.check_spgemm_kept_structure(be)


def test_spgemm_symbolic_by_units(be):
    pc.check_spgemm_units(be)


def test_spgemm_pool_returns_with_the_last_handle(be):
    pc.check_spgemm_pool_release(be)


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


def test_row_partitioned_spgemm_through_the_c_abi(be):
    """kkamd_dist_spgemm_* on the GPU: work-balanced partition from the device, the four slabs of C computed one after the other
    concatenate to the oracle's product (R-MAT scale 12), numeric reuse on every slab, a slab of the wrong height refused"""
    from test_dist_gloo import _check_spgemm_slabs
    _check_spgemm_slabs(be, pc.kk, oracle, oracle.rmat(12, 8), 4)


def test_transposed_modes_through_cached_explicit_transpose(be):
    import torch
    for A0 in (oracle.laplace3d("FE", 60, 50, 40), oracle.rmat(14, 16)):                  # >= 1e6 nnz: the cached-transpose path
        for mode in "TH":
            pc.check_spmv(be, A0, mode, -1.5, 0.5, algo="SPMV_DEFAULT", knobs={"explicit_transpose": 1 if mode == "T" else 2}, max_val=50.0)
    pc.check_spmv(be, oracle.laplace3d("FE", 60, 50, 40), "T", 1.0, 0.0, algo="SPMV_DEFAULT", knobs={"explicit_transpose": 0}, max_val=50.0)
    # values change between calls on the same handle
    A = pc.kk.laplace_matrix("FE", 64, 64, 64)
    h = pc.kk.SPMVHandle("SPMV_DEFAULT"); h.set("explicit_transpose", 1); h0 = pc.kk.SPMVHandle("SPMV_DEFAULT")
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    x = torch.rand(A.numRows(), dtype=torch.float64, device="cuda", generator=g)
    for rep in range(3):
        y = torch.zeros(A.numCols(), dtype=torch.float64, device="cuda"); y0 = torch.zeros_like(y)
        pc.kk.spmv(h, "T", 1.0, A, x, 0.0, y); pc.kk.spmv(h0, "T", 1.0, A, x, 0.0, y0)     # cached transpose vs atomics
        assert (y - y0).abs().max().item() <= 10 * np.finfo(np.float64).eps * 27 * 30 * (rep + 1)
        A.values.mul_(1.0 + rep).add_(0.25)


def test_error_behaviour(be):
    import torch
    A = pc.dev(be, oracle.random_crs(20, 30, 3, seed=2))
    ones = lambda n: torch.ones(n, dtype=torch.float64, device="cuda")
    with pytest.raises(RuntimeError, match="Dimensions do not match"):
        pc.kk.spmv("N", 1.0, A, ones(29), 0.0, ones(20))
    with pytest.raises(RuntimeError, match="Invalid transpose mode"):
        pc.kk.spmv("X", 1.0, A, ones(30), 0.0, ones(20))
    h = pc.kk.SPMVHandle("SPMV_DEFAULT")
    pc.kk.spmv(h, "N", 1.0, A, ones(30), 0.0, ones(20))
    B = pc.dev(be, oracle.random_crs(20, 30, 4, seed=3))
    with pytest.raises(pc.kk.KkamdError) as ei:
        pc.kk.spmv(h, "N", 1.0, B, ones(30), 0.0, ones(20))
    assert ei.value.status == pc.kk._capi.ERR_STATE


def test_degenerate_dimensions(be):
    for nrows, ncols, per in ((0, 0, 0), (0, 7, 0), (7, 0, 0), (9, 9, 0)):
        A0 = oracle.random_crs(nrows, ncols, per, seed=1)
        pc.check_spmv(be, A0, "N", 1.0, 0.0, None, nans=True)
        pc.check_spmv(be, A0, "N", 1.0, 2.0, "SPMV_DEFAULT")
        pc.check_spmv(be, A0, "T", 1.0, 0.0, None, nans=True)


# ------------------------------------------------------------------------------------------- utilities
def test_exclusive_scan(be):
    import torch
    for n in (1, 255, 2049, 5_000_001):
        for dt, code in ((torch.int32, 0), (torch.int64, 1)):
            a = torch.randint(0, 50, (n,), dtype=dt, device="cuda")
            ref = torch.cumsum(a, 0) - a
            pc.kk._capi.check(be.lib, be.lib.kkamd_exclusive_scan(a.data_ptr(), n, code, be.stream()))
            assert torch.equal(a, ref.to(dt))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "structured_*bc1.npz"))))
def test_device_generators_match_reference(be, path):
    g = np.load(path)
    kind, stencil, dims, _ = os.path.basename(path)[len("structured_"):-4].split("_")
    dims = [int(v) for v in dims.split("x")]
    for odt in (np.int32, np.int64):
        rm, ent, val = pc.kk.laplace_matrix(stencil.upper(), *dims, offset_dtype=odt).to_host()
        assert np.array_equal(rm, g["row_map"]) and np.array_equal(ent, g["entries"]) and np.array_equal(val, g["values"])


def test_device_generator_vs_oracle_medium(be):
    for st, dims in (("FE", (37, 41, 29)), ("FD", (50, 20, 33)), ("FD", (301, 77)), ("FE", (64, 65))):
        A0 = oracle.laplace3d(st, *dims) if len(dims) == 3 else oracle.laplace2d(st, *dims)
        rm, ent, val = pc.kk.laplace_matrix(st, *dims).to_host()
        assert np.array_equal(rm, A0.row_map) and np.array_equal(ent, A0.entries) and np.array_equal(val, A0.values)


def test_sort_crs(be):
    for M in (oracle.random_crs(3000, 50000, 40, variance=39, seed=5), oracle.random_crs(30, 1000000, 6000, variance=2000, seed=6)):
        A = pc.dev(be, M)
        pc.kk.sort_crs_matrix(A)
        gold = oracle.Crs(M.nrows, M.ncols, M.row_map, M.entries.copy(), M.values.copy())
        oracle.sort_crs(gold)
        rm, ent, val = A.to_host()
        assert np.array_equal(ent, gold.entries) and np.array_equal(val, gold.values)


# ------------------------------------------------------------------------------------------- SpGEMM
def test_sort_long_rows_merge_transpose(be):
    rng = np.random.default_rng(8)
    lens = [20000, 3, 8193, 0, 600000, 8192, 16384, 131072]
    rm = np.zeros(len(lens) + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = rng.integers(0, 250000, size=rm[-1]).astype(np.int32)
    M = oracle.Crs(len(lens), 250000, rm, ent, 1 + 49 * rng.random(rm[-1]))
    for odt in (np.int32, np.int64):
        A = pc.dev(be, M, odt)
        pc.kk.sort_crs_matrix(A)
        gold = oracle.Crs(M.nrows, M.ncols, M.row_map, M.entries.copy(), M.values.copy()); oracle.sort_crs(gold)
        _, e, v = A.to_host()
        assert np.array_equal(e, gold.entries) and np.array_equal(v, gold.values)
    Cm = pc.kk.sort_and_merge_matrix(pc.dev(be, M))
    gm = oracle.sort_and_merge(oracle.Crs(M.nrows, M.ncols, M.row_map, M.entries.copy(), M.values.copy()))
    r, e, v = Cm.to_host()
    assert np.array_equal(r, gm.row_map) and np.array_equal(e, gm.entries) and np.allclose(v, gm.values, rtol=1e-13, atol=0)
    # transpose of an R-MAT matrix (hub columns become rows far longer than one LDS segment) against the oracle
    R = oracle.rmat(15, 16)
    At = pc.kk.transpose_matrix(pc.dev(be, R, np.int64))
    gt = oracle.transpose(R)
    r, e, v = At.to_host()
    assert np.array_equal(r, gt.row_map) and np.array_equal(e, gt.entries) and np.array_equal(v, gt.values)
    # SpGEMM with an explicitly transposed operand: A^T * A is symmetric
    S = pc.kk.spgemm(At, False, pc.dev(be, R, np.int64), False)
    St = pc.kk.transpose_matrix(S)
    r1, e1, v1 = S.to_host(); r2, e2, v2 = St.to_host()
    assert np.array_equal(r1, r2) and np.array_equal(e1, e2) and np.allclose(v1, v2, rtol=1e-12)


@pytest.mark.parametrize("m,n,k,nnzA,nnzB", [
    (0, 0, 0, 0, 0), (0, 12, 5, 0, 20), (10, 10, 0, 20, 0), (10, 0, 10, 0, 0),
    (10, 10, 10, 0, 0), (10, 10, 10, 20, 0), (10, 10, 10, 0, 20)])
def test_spgemm_degenerate(be, m, n, k, nnzA, nnzB):
    A0 = pc.randomized(oracle.random_crs(m, n, nnzA // m if m else 0, seed=1, sorted_rows=True))
    B0 = pc.randomized(oracle.random_crs(n, k, nnzB // n if n else 0, seed=2, sorted_rows=True))
    assert pc.check_spgemm(be, A0, B0).nnz == 0


@pytest.mark.parametrize("odt", [np.int32, np.int64])
def test_spgemm_reference_shapes(be, odt):
    # Test_Sparse_spgemm.hpp:485-490: 10000 x 8000 x 6000 with 160k nnz each, and 1000 x 500 x 1600
    A0 = pc.randomized(oracle.random_crs(10000, 8000, 16, variance=10, seed=3, sorted_rows=True))
    B0 = pc.randomized(oracle.random_crs(8000, 6000, 20, variance=12, seed=4, sorted_rows=True))
    pc.check_spgemm(be, A0, B0, offset_dtype=odt)
    A1 = pc.randomized(oracle.random_crs(1000, 500, 16, seed=5))
    B1 = pc.randomized(oracle.random_crs(500, 1600, 64, seed=6))
    pc.check_spgemm(be, A1, B1, offset_dtype=odt)


def test_spgemm_float_and_laplacian(be):
    A0 = pc.randomized(oracle.random_crs(2000, 1800, 9, variance=4, seed=13, sorted_rows=True))
    pc.check_spgemm(be, A0, pc.randomized(oracle.random_crs(1800, 900, 7, seed=14, sorted_rows=True)), value_dtype=np.float32)
    L = pc.randomized(oracle.laplace3d("FE", 20, 19, 18))
    pc.check_spgemm(be, L, L)                     # 27-pt squared: 125-pt rows
    F = oracle.laplace2d("FD", 10, 10); E = oracle.laplace2d("FE", 10, 10)
    pc.check_spgemm(be, F, E, reuse=False)        # example/wiki/sparse/KokkosSparse_wiki_spgemm.cpp:45-56


def test_spgemm_all_bins(be):
    B0 = pc.hub_matrix(64, 30000, 40, {0: 9000, 1: 3000, 2: 600, 3: 120, 5: 20000}, seed=1)
    rng = np.random.default_rng(2)
    cols_for = {0: [0], 1: [1], 2: [2], 3: [3, 7], 4: [], 5: [5], 6: [0, 1, 5], 7: list(range(6, 36))}
    rm = [0]; ent = []
    for i in range(8):
        ent += cols_for[i]; rm.append(len(ent))
    A0 = oracle.Crs(8, 64, np.array(rm), np.array(ent, dtype=np.int32), 1 + 49 * rng.random(len(ent)))
    got = pc.check_spgemm(be, A0, B0)
    sizes = np.diff(got.row_map)
    assert sizes[4] == 0 and sizes[3] <= 256 and 256 < sizes[2] <= 2048 and 2048 < sizes[1] <= 5461 and sizes[0] > 5461


def test_spgemm_more_than_2p20_columns(be):
    # dense rows of C wider than one LDS bitmap window (2^20 columns): the column kernel takes several passes
    B0 = pc.hub_matrix(40, 2_600_000, 30, {0: 300_000, 1: 120_000, 2: 9_000}, seed=21)
    rm = [0, 2, 3, 6, 6, 8]
    ent = np.array([0, 1,   2,   0, 2, 9,   1, 30], dtype=np.int32)
    rng = np.random.default_rng(5)
    A0 = oracle.Crs(5, 40, np.array(rm), ent, 1 + 49 * rng.random(len(ent)))
    got = pc.check_spgemm(be, A0, B0, offset_dtype=np.int64)
    assert np.diff(got.row_map).max() > 300_000 and got.ncols == 2_600_000


def test_spgemm_rmat_square(be):
    A0 = oracle.rmat(13, 8, seed=7)               # skewed: hub rows exercise the dense path on real structure
    pc.check_spgemm(be, A0, A0)


def test_spgemm_issue402(be):
    g = np.load(os.path.join(GOLD, "matrix_issue402.npz"))
    A0 = oracle.Crs(1813, 1813, g["row_map"], g["entries"], g["values"])
    At = oracle.transpose(A0)
    oracle.sort_crs(A0); oracle.sort_crs(At)
    pc.check_spgemm(be, A0, At, reuse=False)      # Test_Sparse_spgemm.hpp:372-442


def test_spgemm_handle_contract(be):
    A0 = pc.randomized(oracle.random_crs(40, 30, 5, seed=7, sorted_rows=True))
    B0 = pc.randomized(oracle.random_crs(30, 20, 4, seed=8, sorted_rows=True))
    A, B = pc.dev(be, A0), pc.dev(be, B0)
    kh = pc.kk.KokkosKernelsHandle(be)
    with pytest.raises(ValueError, match="does not have an SpGEMM handle"):
        pc.kk.spgemm_symbolic(kh, A, False, B, False)
    kh.create_spgemm_handle()
    Cfake = pc.kk.CrsMatrix(40, 20, be.empty(41, np.int32), be.empty(1, np.int32), be.empty(1, np.float64), backend=be)
    with pytest.raises(ValueError, match="must first call spgemm_symbolic"):
        pc.kk.spgemm_numeric(kh, A, False, B, False, Cfake)
    C1 = pc.kk.spgemm_symbolic(kh, A, False, B, False)
    C2 = pc.kk.spgemm_symbolic(kh, A, False, B, False)
    assert np.array_equal(be.to_numpy(C1.graph.row_map), oracle.spgemm_symbolic(A0, B0)[0]) and C2.nnz() == C1.nnz()
    pc.kk.spgemm_numeric(kh, A, False, B, False, C1)
    kh.destroy_spgemm_handle()
    Cn = pc.kk.spgemm(A, False, B, False)
    ok, msg = oracle.is_same_matrix(oracle.Crs(40, 20, *[np.asarray(v) for v in Cn.to_host()]), oracle.spgemm(A0, B0))
    assert ok, msg


def _set_default(be, key, value):
    pc.kk._capi.check(be.lib, be.lib.kkamd_set_default(key.encode(), int(value)))


def test_spgemm_compression(be):
    """a18 on the HIP build: B compressed into 32-column sets + masks for the symbolic phase (impl_compression.hpp), kept by the
    0.85 rule on stencils, dropped on matrices without column runs, forced through every symbolic row bin (wave, block-small,
    block-large, bitmap with one and several windows), both offset types; numeric reuse after a compressed symbolic
    (sparse/unit_test/Test_Sparse_spgemm.hpp:243-252,485-504)"""
    L = pc.randomized(oracle.laplace3d("FE", 60, 60, 60))
    pc.check_spgemm(be, L, L, expect_compressed=False)                                  # off unless asked for
    pc.check_spgemm(be, L, L, options={"compression": 1}, expect_compressed=True)       # runs of three neighbours: pays
    pc.check_spgemm(be, L, L, options={"compression": 2}, expect_compressed=True, offset_dtype=np.int64)
    pc.check_spgemm(be, L, L, options={"compression": 1, "compression_cut_off": 0.1}, expect_compressed=False)
    S = pc.randomized(oracle.random_crs(4000, 300000, 9, variance=4, seed=2, sorted_rows=True))
    St = pc.randomized(oracle.random_crs(300000, 400000, 7, variance=3, seed=3, sorted_rows=True))
    pc.check_spgemm(be, S, St, options={"compression": 1}, expect_compressed=False)     # scattered columns: dropped by the rule
    pc.check_spgemm(be, S, St, options={"compression": 2}, expect_compressed=True)      # ... unless forced
    R = oracle.rmat(13, 8, seed=7)
    pc.check_spgemm(be, R, R, options={"compression": 1}, expect_compressed=True)       # 8192 columns, hub rows: 0.43 of the insertions left
    pc.check_spgemm(be, R, R, options={"compression": 2}, expect_compressed=True)       # hub rows: bitmap kernel on masks
    pc.check_spgemm(be, R, R, options={"compression": 2}, expect_compressed=True, offset_dtype=np.int64, value_dtype=np.float32)
    band = pc.randomized(oracle.random_crs(6000, 6000, 40, variance=10, seed=5, bandwidth=150, sorted_rows=True))
    lens = [3, 40, 200, 900, 2500, 0, 60]
    rm = np.concatenate([[0], np.cumsum(lens)])
    rng = np.random.default_rng(9)
    ent = np.concatenate([np.sort(rng.choice(6000, size=l, replace=False)) for l in lens]).astype(np.int32)
    A0 = oracle.Crs(len(lens), 6000, rm, ent, 1 + 49 * rng.random(int(rm[-1])))
    for odt in (np.int32, np.int64):
        pc.check_spgemm(be, A0, band, offset_dtype=odt, options={"compression": 2}, expect_compressed=True)
    try:
        _set_default(be, "spgemm_win_bits", 4096)
        pc.check_spgemm(be, A0, band, options={"compression": 2}, expect_compressed=True)
    finally:
        _set_default(be, "spgemm_win_bits", 1 << 20)
    U = pc.randomized(oracle.random_crs(300, 300, 8, variance=3, seed=8))               # unsorted B cannot be compressed
    pc.check_spgemm(be, U, U, options={"compression": 2}, expect_compressed=False)


def test_spgemm_dense_accumulator_algorithm(be):
    """a21 on the HIP build: SPGEMM_KK_DENSE / SPGEMM_ACC_DENSE -- every row through a k-wide dense accumulator
    (impl_speed.hpp:28-150), int32 / int64 offsets, fp64 / fp32 values, numeric reuse with new values"""
    for A0, B0 in ((pc.randomized(oracle.laplace3d("FE", 30, 25, 20)),) * 2,
                   (oracle.rmat(12, 8, seed=3),) * 2,
                   (pc.randomized(oracle.random_crs(120, 900, 11, variance=6, seed=4, sorted_rows=True)),
                    pc.randomized(oracle.random_crs(900, 700, 9, variance=5, seed=6, sorted_rows=True))),
                   (pc.hub_matrix(30, 2000, 6, {3: 700}, seed=2), pc.randomized(oracle.random_crs(2000, 1500, 10, variance=4, seed=7, sorted_rows=True)))):
        pc.check_spgemm(be, A0, B0, algo="SPGEMM_KK_DENSE")
        pc.check_spgemm(be, A0, B0, options={"accumulator": 1}, offset_dtype=np.int64, value_dtype=np.float32)
    kh = pc.kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK_DENSE")
    assert kh.get_spgemm_handle().get(8) == 1
    kh.get_spgemm_handle().set("accumulator", 2)                 # back to the hash accumulators
    assert kh.get_spgemm_handle().get(8) == 0
    for alias in ("SPGEMM_KK_MEMORY", "SPGEMM_KK_SPEED", "SPGEMM_KK_MEMSPEED", "SPGEMM_KK_LP", "SPGEMM_DEFAULT", "SPGEMM_DEBUG", "SPGEMM_SERIAL"):
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


def test_spgemm_options_act_or_are_recorded(be, capfd):
    """kkamd_spgemm_set on the HIP build: options act, the reference's hints are accepted and recorded, unknown keys fail; the
    host-sequential algorithms of the reference give the same C"""
    kh = pc.kk.KokkosKernelsHandle(be)
    L = pc.randomized(oracle.laplace3d("FE", 12, 10, 8))
    for algo in ("SPGEMM_DEBUG", "SPGEMM_SERIAL"):
        kh.create_spgemm_handle(algo)
        assert kh.get_spgemm_handle().get(9) == pc.kk.sparse._SPGEMM_ALGOS[algo]
        pc.check_spgemm(be, L, L, algo=algo)
    with pytest.raises(RuntimeError):
        kh.create_spgemm_handle("SPGEMM_CUSPARSE")
    kh.create_spgemm_handle()
    sh = kh.get_spgemm_handle()
    hints = ("team_work_size", "shmem_size", "suggested_team_size", "suggested_vector_size", "dynamic_scheduling", "min_hash_size_scale",
             "first_level_hash_cut_off", "mkl_sort_option", "read_write_cost_calc", "compression_steps", "max_col_dense_acc", "sort_option")
    for i, key in enumerate(hints):
        sh.set(key, 16 + i)
        assert sh.get_hint(key) == 16 + i
    assert sh.get(10) == len(hints)
    for bad_key, bad_val in (("no_such_option", 1), ("compression", 3), ("compression_cut_off", 0.0), ("algorithm", 9), ("accumulator", 5)):
        with pytest.raises(pc.kk.KkamdError) as e:
            sh.set(bad_key, bad_val)
        assert e.value.status == pc.kk._capi.ERR_INVALID_ARG
    sh.set("verbose", 1); sh.set("compression", 1); sh.set("team_work_size", 256)
    A = pc.dev(be, L)
    Cm = pc.kk.spgemm_symbolic(kh, A, False, A, False)
    pc.kk.spgemm_numeric(kh, A, False, A, False, Cm)
    out = capfd.readouterr().out
    assert "kkamd spgemm symbolic" in out and "compression kept" in out and "kkamd spgemm numeric (SPGEMM_KK)" in out, out
    assert "hint team_work_size = 256 recorded" in out, out


def test_fuzz_slice(be):
    """a 60-second slice of the randomised sweep (tests/fuzz_cases.py; tools/fuzz_gpu.py runs it for minutes): every kind of case at
    least once -- SpGEMM skewed / R-MAT with compression and dense-accumulator options / unsorted / long rows, SpMV on irregular
    rows and through every plan mode and rank-2 kernel, sort / merge / transpose, spmv_struct"""
    import fuzz_cases
    n_ok, per_kind, last = fuzz_cases.run(be, 60.0, seed0=3_000_000)
    print("fuzz slice: %d cases, per kind %s, seeds 3000000..%d" % (n_ok, per_kind, last))
    assert n_ok >= 16 and min(per_kind) >= 1, (n_ok, per_kind)


# ------------------------------------------------------------------------------------------- full-size properties
def test_full_size_27pt_properties(be):
    """BASELINE config C2 (27-pt 300^3): size-independent checks -- nnz closed form, A*1 = row-sum vector
    (interior rows 0, boundary rows 1), linearity, and agreement of the two kernels."""
    import torch
    n = 300
    A = pc.kk.laplace_matrix("FE", n, n, n)
    assert A.nnz() == 724_150_792 and A.numRows() == 27_000_000
    ones = torch.ones(A.numCols(), dtype=torch.float64, device="cuda")
    y = torch.full((A.numRows(),), float("nan"), dtype=torch.float64, device="cuda")
    h = pc.kk.SPMVHandle("SPMV_DEFAULT")
    pc.kk.spmv(h, "N", 1.0, A, ones, 0.0, y)
    lens = (A.graph.row_map[1:] - A.graph.row_map[:-1])
    assert (y[lens == 27] == 0.0).all().item() and (y[lens < 27] == 1.0).all().item()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x1 = torch.rand(A.numCols(), dtype=torch.float64, device="cuda", generator=g)
    x2 = torch.rand(A.numCols(), dtype=torch.float64, device="cuda", generator=g)
    y1 = torch.empty_like(y); y2 = torch.empty_like(y); y12 = torch.empty_like(y); yv = torch.empty_like(y)
    pc.kk.spmv(h, "N", 1.0, A, x1, 0.0, y1)
    pc.kk.spmv(h, "N", 1.0, A, x2, 0.0, y2)
    pc.kk.spmv(h, "N", 1.0, A, x1 + 2.0 * x2, 0.0, y12)
    tol = 10 * np.finfo(np.float64).eps * 27 * 32 * 3
    assert (y12 - (y1 + 2.0 * y2)).abs().max().item() <= tol
    pc.kk.spmv("N", 1.0, A, x1, 0.0, yv)                      # handle-less (vector kernel) vs planned (stream kernel)
    assert (yv - y1).abs().max().item() <= tol
    # beta path: y := 1*A*x + 1*y  ==  previous + y
    yb = y2.clone()
    pc.kk.spmv(h, "N", 1.0, A, x1, 1.0, yb)
    assert (yb - (y1 + y2)).abs().max().item() <= tol
    # structured path on the same matrix == CRS path
    ys = torch.zeros_like(y)
    pc.kk.spmv_struct("N", 2, (n, n, n), 1.0, A, x1, 0.0, ys)
    assert (ys - y1).abs().max().item() <= tol


def test_column_slab_copy(be):
    # rank 1 on gather-bound matrices (kk_spmv_colslab.hip): the column-slab copy forced with narrow slabs (many slabs on small matrices),
    # every type pair, beta 0 over NaNs, duplicates, rows longer than a sort tile, empty rows
    base = oracle.random_crs(30000, 30000, 12, variance=5, seed=3)
    r, e = base.row_map, base.entries.copy()
    e[r[10]:r[10] + 3] = e[r[10]]
    cases = [base, oracle.Crs(30000, 30000, r, e, base.values), pc.hub_matrix(3000, 9000, 6, {5: 7000, 17: 1500, 2999: 4200}, seed=3),
             oracle.random_crs(500, 90000, 40, variance=10, seed=6)]
    for A0 in cases:
        for shift in (4, 9, 14):
            kn = {"colslab": 2, "colslab_shift": shift}
            h = pc.check_spmv(be, A0, "N", 1.5, 0.5, "SPMV_DEFAULT", knobs=kn, max_val=50.0, expect={"colslab": 1})
            assert h.query("colslab_slabs") == -(-A0.ncols // (1 << h.query("colslab_shift"))) <= 256
            pc.check_spmv(be, A0, "N", 1.0, 0.0, "SPMV_DEFAULT", knobs=kn, max_val=50.0, nans=True)
        pc.check_spmv(be, A0, "N", -1.0, 2.0, "SPMV_DEFAULT", knobs={"colslab": 2, "colslab_shift": 6}, max_val=50.0, offset_dtype=np.int64, value_dtype=np.float32)
        pc.check_spmv(be, A0, "N", 1.0, 1.0, "SPMV_DEFAULT", knobs={"colslab": 2, "colslab_shift": 6, "colslab_const": 1}, max_val=50.0, value_dtype=np.float32, vec_dtype=np.float32)
        pc.check_spmv(be, A0, "T", 1.0, 0.0, "SPMV_DEFAULT", knobs={"colslab": 2, "colslab_shift": 6}, max_val=50.0)
    pc.check_spmv(be, base, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=50.0, knobs={"colslab": 1}, expect={"colslab": 0, "colslab_tried": 1})       # small matrix: the gates say no
    pc.check_spmv(be, base, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=50.0, expect={"colslab": 0, "colslab_tried": 1})       # the default handle applies its rule (deterministic form): a small matrix says no
    pc.check_spmv(be, base, "N", 1.0, 0.0, "SPMV_DEFAULT", max_val=50.0, knobs={"colslab": 0}, expect={"colslab": 0, "colslab_tried": 0})


def test_column_slab_follows_value_changes(be):
    # the copy holds its own values; a call fingerprints A.values tile by tile and moves the tiles that changed
    A0 = oracle.random_crs(25000, 25000, 10, variance=3, seed=8)
    rng = np.random.default_rng(1)
    x = rng.random(A0.ncols)
    A = pc.dev(be, A0)
    h = pc.kk.SPMVHandle("SPMV_DEFAULT"); h.set("colslab", 2); h.set("colslab_shift", 8)
    xd, yd = be.from_numpy(x), be.from_numpy(np.zeros(A0.nrows))
    def run_and_check(vals):
        pc.kk.spmv(h, "N", 1.0, A, xd, 0.0, yd)
        exp = oracle.spmv_serial("N", oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries, vals), 1.0, x, 0.0, np.zeros(A0.nrows))
        np.testing.assert_allclose(be.to_numpy(yd), exp, rtol=1e-12, atol=1e-12)
    run_and_check(A0.values)
    assert h.query("colslab") == 1
    v = A0.values.copy()
    v[50000] = -7.25
    A.values[:] = be.from_numpy(v); run_and_check(v)
    v = v * 3.0 + 1.0
    A.values[:] = be.from_numpy(v); run_and_check(v)
    v[[0, len(v) - 1]] = v[[len(v) - 1, 0]]
    A.values[:] = be.from_numpy(v); run_and_check(v)


def test_column_slab_selection_on_a_gather_bound_matrix(be):
    # the automatic mode: uniform random columns over 3e6 columns (x = 24 MB, six L2s): the analysis calls it gather-bound, both
    # kernels are timed on the first call and whichever is kept, the result is the oracle's
    n, k = 3_000_000, 8
    rng = np.random.default_rng(12)
    ent = np.sort(rng.integers(0, n, size=(n, k), dtype=np.int64), axis=1).astype(np.int32).reshape(-1)
    A0 = oracle.Crs(n, n, np.arange(0, n * k + 1, k, dtype=np.int64), ent, rng.random(n * k) + 0.5)
    h = pc.check_spmv(be, A0, "N", 1.5, 0.0, "SPMV_DEFAULT", max_val=2.0, nans=True, knobs={"colslab": 1, "colslab_min_knnz": 20000}, expect={"colslab_tried": 1})
    assert h.query("colslab_crs_us") > 0 and h.query("colslab_us") > 0, (h.query("colslab_crs_us"), h.query("colslab_us"))
    print("column-slab selection: CRS %d us, copy %d us, kept %d" % (h.query("colslab_crs_us"), h.query("colslab_us"), h.query("colslab")))
    h2 = pc.check_spmv(be, A0, "N", 1.0, 0.5, "SPMV_DEFAULT", max_val=2.0, knobs={"colslab": 2}, expect={"colslab": 1, "colslab_shift": 18, "colslab_slabs": 12})
