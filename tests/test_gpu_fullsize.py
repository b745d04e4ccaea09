"""Parity at the sizes that are BENCHMARKED (BASELINE.json configs 2-4 and the unstructured matrices of the plan table):
the default-knob paths -- exactly what bench.py / tools/run_configs.py time -- compared directly with the CPU oracle on the
same inputs, with the reference's own comparators (sparse/unit_test/Test_Sparse_spmv.hpp:67-104,181;
sparse/unit_test/Test_Sparse_Utils.hpp:39-127).  The matrices are generated on the device (generator parity with the
oracle / the reference's generator is pinned at smaller sizes in test_gpu_parity.py and test_oracle.py) and copied to the
host for the oracle, so both sides see the same arrays.  KK_SKIP_HEAVY=1 skips this file (iteration runs only)."""
import os

import numpy as np
import pytest

import oracle
import parity_cases as pc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("KK_SKIP_HEAVY") == "1", reason="KK_SKIP_HEAVY=1")]

EPS = float(np.finfo(np.float64).eps)


@pytest.fixture(scope="module")
def be():
    import torch
    assert torch.cuda.is_available(), "GPU parity tests need the MI355X"
    oracle.set_omp_threads(oracle.usable_cpus())
    return pc.kk.torch_backend()


@pytest.fixture(scope="module")
def c2(be):
    """27-pt FE Laplacian 300^3 on the device + its host copy for the oracle"""
    n = 300
    A = pc.kk.laplace_matrix("FE", n, n, n)
    assert A.nnz() == 724_150_792 and A.numRows() == 27_000_000
    rm, ent, val = A.to_host()
    return A, oracle.Crs(A.numRows(), A.numCols(), rm.astype(np.int64), ent, val), rm


def _fspmv(exp, got, tol, what):
    nanmis = np.isnan(exp) ^ np.isnan(got)
    assert not nanmis.any(), "%s: NaN mismatch in %d places" % (what, int(nanmis.sum()))
    err = float(np.nanmax(np.abs(exp - got)))
    assert err <= tol, "%s: max |expected - y| = %g > %g" % (what, err, tol)
    return err


def test_c2_default_plan_against_oracle(be, c2):
    """BASELINE config 2, the path bench.py times: SPMVHandle(SPMV_DEFAULT), default knobs -> row-pattern records on >= 90 % of
    the tiles; y compared with the Kokkos::Serial restatement (fSPMV bound), beta = 0 over a NaN-seeded y and beta = 1."""
    import torch
    A, A0, rm = c2
    rng = np.random.default_rng(17312837)
    x = rng.integers(-20, 20, size=A0.ncols).astype(np.float64)          # the perf driver's x (KokkosSparse_kk_spmv.cpp:95-99)
    y0 = rng.integers(-20, 20, size=A0.nrows).astype(np.float64)
    xd = torch.from_numpy(x).cuda()
    h = pc.kk.SPMVHandle("SPMV_DEFAULT")
    for beta, seed_nan in ((0.0, True), (1.0, False)):
        yh = y0.copy()
        if seed_nan:
            yh[::19] = np.nan
        yd = torch.from_numpy(yh).cuda()
        pc.kk.spmv(h, "N", 1.0, A, xd, beta, yd)
        assert h.query("tile") == 4096 and h.query("window_staged_x") == 1
        assert h.query("pattern_tiles") >= 0.9 * h.query("tiles"), (h.query("pattern_tiles"), h.query("tiles"))
        exp = oracle.spmv_serial("N", A0, 1.0, x, beta, np.where(np.isnan(yh), 0.0, yh) if beta == 0.0 else yh.copy())
        tol = 10 * EPS * (abs(beta) * 20.0 + 27 * 32.0 * 20.0)           # 10 eps (beta max_y + alpha max_nnz_row max_val max_x)
        _fspmv(exp, yd.cpu().numpy(), tol, "C2 default plan beta=%g" % beta)
    # the plan keeps less than a quarter of a byte per nonzero (tile descriptors, window meta, pattern records)
    assert h.query("plan_bytes") <= 0.25 * A0.nnz, h.query("plan_bytes")
    # handle-less route (no codes, plain nnz-split kernel) on the same inputs
    yd = torch.full((A0.nrows,), float("nan"), dtype=torch.float64, device="cuda")
    pc.kk.spmv("N", 1.0, A, xd, 0.0, yd)
    _fspmv(oracle.spmv_serial("N", A0, 1.0, x, 0.0, np.zeros(A0.nrows)), yd.cpu().numpy(), 10 * EPS * 27 * 32.0 * 20.0, "C2 handle-less")


@pytest.mark.parametrize("layout", ["right", "left"])
def test_c3_all_columns_against_oracle(be, c2, layout):
    """BASELINE config 3 (C2 x 16 right-hand sides), default knobs, LayoutRight and LayoutLeft: all 16 columns against the
    OpenMP port of the host SpMV_MV functor (sparse/impl/KokkosSparse_spmv_impl.hpp:745-792)."""
    import torch
    A, A0, rm = c2
    nv = 16
    rng = np.random.default_rng(5)
    X = rng.integers(-20, 20, size=(A0.ncols, nv)).astype(np.float64)
    Y0 = np.full((A0.nrows, nv), np.nan)
    if layout == "left":
        Xd = torch.from_numpy(np.ascontiguousarray(X.T)).cuda().t()
        Yd = torch.from_numpy(np.ascontiguousarray(Y0.T)).cuda().t()
    else:
        Xd = torch.from_numpy(X).cuda(); Yd = torch.from_numpy(Y0).cuda()
    h = pc.kk.SPMVHandle("SPMV_DEFAULT")
    pc.kk.spmv(h, "N", 1.0, A, Xd, 0.0, Yd)
    # the plane-marching kernel engaged; the gather kernel is left the few rows the generator couples across a lattice edge
    assert h.query("mv_period") == 300 * 300 and h.query("mv4_workgroups") > 0 and h.query("mv4_stencil") == 27
    assert h.query("mv4_other_rows") < 0.005 * A0.nrows, h.query("mv4_other_rows")
    exp = np.zeros((A0.nrows, nv))
    oracle.spmv_mv_omp(rm.astype(np.int32), A0.entries, A0.values, 1.0, X, 0.0, exp)
    tol = 10 * EPS * 27 * 32.0 * 20.0
    _fspmv(exp, Yd.cpu().numpy(), tol, "C3 %s" % layout)
    # beta = -1 on the row-major pair
    if layout == "right":
        Y1 = rng.integers(-20, 20, size=(A0.nrows, nv)).astype(np.float64)
        Yd = torch.from_numpy(Y1).cuda()
        pc.kk.spmv(h, "N", 2.0, A, Xd, -1.0, Yd)
        exp = Y1.copy()
        oracle.spmv_mv_omp(rm.astype(np.int32), A0.entries, A0.values, 2.0, X, -1.0, exp)
        _fspmv(exp, Yd.cpu().numpy(), 10 * EPS * (20.0 + 2 * 27 * 32.0 * 20.0), "C3 right beta=-1")


def _avail_gb():
    try:
        import psutil
        return psutil.virtual_memory().available / 2**30
    except Exception:
        return 0.0


def test_spgemm_rmat_against_oracle(be):
    """BASELINE config 4 family: C = A*A on R-MAT (scale 18 when the host has the memory for the oracle's copy of C, else 17),
    row_map and entries IDENTICAL to SPGEMM_DEBUG + sort, values to 1e-7 relative (Test_Sparse_Utils.hpp:86-119)."""
    scale = 18 if _avail_gb() >= 90 else 17
    R = oracle.rmat(scale, 16)
    A = pc.dev(be, R, offset_dtype=np.int64)
    Cd = pc.kk.spgemm(A, False, A, False)
    rm, ent, val = Cd.to_host()
    G = oracle.spgemm(R, R)
    assert Cd.nnz() == G.nnz, (scale, Cd.nnz(), G.nnz)
    assert np.array_equal(rm.astype(np.int64), G.row_map), "row_map differs"
    assert np.array_equal(ent, G.entries), "entries differ"
    worst = 0.0
    for s in range(0, G.nnz, 1 << 26):                                     # chunked: no 10-GB temporaries
        a = val[s:s + (1 << 26)]; b = G.values[s:s + (1 << 26)]
        den = np.abs(a) + np.abs(b)
        worst = max(worst, float((np.abs(a - b) / np.where(den > 0, den, 1.0)).max()))
    assert worst <= 1e-7, "values differ: max rel %g" % worst
    print("spgemm R-MAT scale %d: nnz(C) %d identical structure, max rel value error %.2e" % (scale, G.nnz, worst))


def test_spgemm_rmat_s20_without_a_host_copy_of_c(be):
    """The largest single-GPU member of BASELINE config 4 (R-MAT scale 20, edge factor 16: nnz(C) = 9.69e9, 116 GB -- no host
    copy of C is possible).  Checked through what does not need one: (1) row_map of C bit-exact against the OpenMP KKMEM
    symbolic kernel of the oracle (counts only); (2) every row of C strictly ascending in its columns, all inside [0, k) -- with
    (1), each row holds exactly as many DISTINCT columns as the reference's; (3) C x == A (A x) for a positive x, C x on the
    device, A (A x) by the host oracle SpMV -- ties every value to its column; (4) numeric again on the same handle with new
    values of A (the reference's reuse case, Test_Sparse_spgemm.hpp:243-252): entries(C) untouched, C' x == A' (A x)."""
    import torch
    if torch.cuda.mem_get_info()[0] < 200 * 2**30:
        pytest.skip("needs 200 GB of free HBM")
    R = oracle.rmat(20, 16)
    A = pc.dev(be, R, offset_dtype=np.int64)
    kh = pc.kk.KokkosKernelsHandle(be); kh.create_spgemm_handle("SPGEMM_KK")
    Cd = pc.kk.spgemm_symbolic(kh, A, False, A, False)
    rm_gold = oracle.spgemm_symbolic_kkmem_omp(R, R)
    assert Cd.nnz() == int(rm_gold[-1]), (Cd.nnz(), int(rm_gold[-1]))
    assert torch.equal(Cd.graph.row_map.cpu(), torch.from_numpy(rm_gold)), "row_map differs from the KKMEM symbolic kernel's"
    pc.kk.spgemm_numeric(kh, A, False, A, False, Cd)
    ent, rm = Cd.graph.entries, Cd.graph.row_map
    nnz = Cd.nnz()
    starts = torch.zeros(nnz + 1, dtype=torch.bool, device="cuda")
    starts[rm] = True                                                        # entry j starts a row (empty rows mark the same place twice)
    lo = int(ent.min().item()); hi = int(ent.max().item())
    assert lo >= 0 and hi < R.ncols, (lo, hi)
    step = 1 << 28
    for s in range(0, nnz - 1, step):                                        # chunked: no 40-GB temporaries
        e = min(nnz - 1, s + step)
        asc = (ent[s + 1:e + 1] > ent[s:e]) | starts[s + 1:e + 1]
        assert bool(asc.all().item()), "a row of C is not strictly ascending near entry %d" % s
    del starts
    rng = np.random.default_rng(20)
    x = 0.5 + rng.random(R.ncols)
    xd = torch.from_numpy(x).cuda()
    # a second probe with mixed signs: sums cancel, so the bound is relative to the row's norm -- |C| |x| row by row, which is
    # A (A |x|) for the non-negative A -- not to the result
    xs = (0.5 + rng.random(R.ncols)) * np.where(rng.random(R.ncols) < 0.5, -1.0, 1.0)
    xsd = torch.from_numpy(xs).cuda()

    def check(Rh, tag):
        ax = oracle.spmv_omp(R.row_map, R.entries, R.values, 1.0, x, 0.0, np.zeros(R.nrows))          # B = A keeps its values
        gold = oracle.spmv_omp(Rh.row_map, Rh.entries, Rh.values, 1.0, ax, 0.0, np.zeros(R.nrows))
        yd = torch.full((R.nrows,), float("nan"), dtype=torch.float64, device="cuda")
        pc.kk.spmv("N", 1.0, Cd, xd, 0.0, yd)
        got = yd.cpu().numpy()
        den = np.abs(gold) + np.abs(got)
        rel = float((np.abs(got - gold) / np.where(den > 0, den, 1.0)).max())
        assert rel <= 1e-12, "%s: C x differs from A (A x): max rel %g" % (tag, rel)             # all terms positive: no cancellation
        axs = oracle.spmv_omp(R.row_map, R.entries, R.values, 1.0, xs, 0.0, np.zeros(R.nrows))
        golds = oracle.spmv_omp(Rh.row_map, Rh.entries, Rh.values, 1.0, axs, 0.0, np.zeros(R.nrows))
        axa = oracle.spmv_omp(R.row_map, R.entries, R.values, 1.0, np.abs(xs), 0.0, np.zeros(R.nrows))
        norm = oracle.spmv_omp(Rh.row_map, Rh.entries, Rh.values, 1.0, axa, 0.0, np.zeros(R.nrows))      # sum_j |C_ij| |x_j|
        pc.kk.spmv("N", 1.0, Cd, xsd, 0.0, yd)
        gots = yd.cpu().numpy()
        rels = float((np.abs(gots - golds) / np.where(norm > 0, norm, 1.0)).max())
        assert rels <= 1e-12, "%s: C x differs from A (A x) for the mixed-sign probe: max error relative to the row norm %g" % (tag, rels)
        return max(rel, rels)
    rel1 = check(R, "numeric")
    ent_sum = int(ent[::1009].to(torch.int64).sum().item())
    R2 = oracle.Crs(R.nrows, R.ncols, R.row_map, R.entries, 1.0 + 49.0 * rng.random(R.nnz))
    A2 = pc.kk.CrsMatrix(R.nrows, R.ncols, A.graph.row_map, A.graph.entries, torch.from_numpy(R2.values).cuda(), backend=be)
    pc.kk.spgemm_numeric(kh, A2, False, A, False, Cd)
    assert int(Cd.graph.entries[::1009].to(torch.int64).sum().item()) == ent_sum, "numeric reuse changed entries(C)"
    rel2 = check(R2, "numeric reuse")
    print("spgemm R-MAT scale 20: nnz(C) %d, row_map identical to the KKMEM symbolic kernel, rows strictly ascending, C x vs A (A x) max rel %.2e / %.2e (reuse)"
          % (nnz, rel1, rel2))


def test_c4_slabs_of_8_rmat_s22(be):
    """BASELINE config 4 AS SPECIFIED -- C = A*A on R-MAT scale 22, edge factor 16 -- at its stated scale on one GPU: the product does not fit a
    GPU (nnz(C) = 7.2e10 = 863 GB), one rank's slab of the 8-GPU row partition does (rows of C are independent given all of B; B
    replicated).  kkamd_dist_spgemm_partition(world = 8) cuts A into slabs of near-equal multiplications; the slab with the FEWEST rows (the
    hub rows: densest rows of C) and the one with the MOST rows (the largest slab of C: 10.6e9 entries, 127 GB) go through
    kkamd_dist_spgemm_symbolic / _numeric.  Checked per slab: (1) row_map bit-exact against the OpenMP KKMEM symbolic kernel of the oracle on those
    rows; (2) every row strictly ascending, columns inside [0, k); (3) C_slab x == A_slab (A x) for a positive and a mixed-sign probe (ties every
    value to its column); (4) numeric again on the same operator with new values of the slab (reuse: entries untouched, C' x == A' (A x))."""
    import torch
    from kokkos_kernels_amd.dist import DistSpgemm
    import gc
    gc.collect(); torch.cuda.empty_cache()                                   # (the tests before this one leave 100+ GB in torch's cache and the library's store of the last product)
    pc.kk._capi.check(be.lib, be.lib.kkamd_release_scratch())
    if torch.cuda.mem_get_info()[0] < 200 * 2**30:
        pytest.skip("needs 200 GB of free HBM")
    world = 8
    R = oracle.rmat(22, 16)
    A = pc.dev(be, R, offset_dtype=np.int64)
    offsets, mults = DistSpgemm.partition(A, A, world)
    assert offsets[0] == 0 and offsets[-1] == R.nrows and all(b > a for a, b in zip(offsets, offsets[1:]))
    tot = oracle.spgemm_mults(R, R)[0]
    assert sum(mults) == tot and max(mults) <= 1.02 * tot / world, (mults, tot)          # balanced by multiplications (contiguous slabs: to within one row's products)
    rows = [offsets[r + 1] - offsets[r] for r in range(world)]
    rng = np.random.default_rng(22)
    x = 0.5 + rng.random(R.ncols)
    xs = (0.5 + rng.random(R.ncols)) * np.where(rng.random(R.ncols) < 0.5, -1.0, 1.0)
    ax = oracle.spmv_omp(R.row_map, R.entries, R.values, 1.0, x, 0.0, np.zeros(R.nrows))            # B = A
    axs = oracle.spmv_omp(R.row_map, R.entries, R.values, 1.0, xs, 0.0, np.zeros(R.nrows))
    axa = oracle.spmv_omp(R.row_map, R.entries, R.values, 1.0, np.abs(xs), 0.0, np.zeros(R.nrows))
    xd, xsd = torch.from_numpy(x).cuda(), torch.from_numpy(xs).cuda()
    for rank in (int(np.argmin(rows)), int(np.argmax(rows))):
        o0, o1 = offsets[rank], offsets[rank + 1]
        e0, e1 = int(R.row_map[o0]), int(R.row_map[o1])
        S = oracle.Crs(o1 - o0, R.ncols, (R.row_map[o0:o1 + 1] - R.row_map[o0]).copy(), R.entries[e0:e1], R.values[e0:e1])
        Sd = pc.kk.CrsMatrix(S.nrows, S.ncols, (A.graph.row_map[o0:o1 + 1] - A.graph.row_map[o0]).contiguous(), A.graph.entries[e0:e1], A.values[e0:e1], backend=be)
        op = DistSpgemm(offsets, rank, be)
        Cd = op.symbolic(Sd, A)
        assert op.query("rows_local") == S.nrows and op.query("mults_local") == mults[rank]
        rm_gold = oracle.spgemm_symbolic_kkmem_omp(S, R)
        assert Cd.nnz() == int(rm_gold[-1]) == op.query("c_nnz_local"), (rank, Cd.nnz(), int(rm_gold[-1]))
        assert torch.equal(Cd.graph.row_map.cpu(), torch.from_numpy(rm_gold)), "rank %d: row_map differs from the KKMEM symbolic kernel's" % rank
        op.numeric(Sd, A, Cd)
        ent, rm, nnz = Cd.graph.entries, Cd.graph.row_map, Cd.nnz()
        lo = int(ent.min().item()); hi = int(ent.max().item())
        assert lo >= 0 and hi < R.ncols, (lo, hi)
        starts = torch.zeros(nnz + 1, dtype=torch.bool, device="cuda")
        starts[rm] = True
        for s0 in range(0, nnz - 1, 1 << 28):                                 # chunked: no 40-GB temporaries
            e = min(nnz - 1, s0 + (1 << 28))
            assert bool(((ent[s0 + 1:e + 1] > ent[s0:e]) | starts[s0 + 1:e + 1]).all().item()), "rank %d: a row of C is not strictly ascending near entry %d" % (rank, s0)
        del starts

        def check(Sh, tag):
            gold = oracle.spmv_omp(Sh.row_map, Sh.entries, Sh.values, 1.0, ax, 0.0, np.zeros(Sh.nrows))
            yd = torch.full((Sh.nrows,), float("nan"), dtype=torch.float64, device="cuda")
            pc.kk.spmv("N", 1.0, Cd, xd, 0.0, yd)
            got = yd.cpu().numpy()
            den = np.abs(gold) + np.abs(got)
            rel = float((np.abs(got - gold) / np.where(den > 0, den, 1.0)).max())
            assert rel <= 1e-12, "rank %d %s: C x differs from A (A x): max rel %g" % (rank, tag, rel)
            golds = oracle.spmv_omp(Sh.row_map, Sh.entries, Sh.values, 1.0, axs, 0.0, np.zeros(Sh.nrows))
            norm = oracle.spmv_omp(Sh.row_map, Sh.entries, Sh.values, 1.0, axa, 0.0, np.zeros(Sh.nrows))      # sum_j |C_ij| |x_j|
            pc.kk.spmv("N", 1.0, Cd, xsd, 0.0, yd)
            rels = float((np.abs(yd.cpu().numpy() - golds) / np.where(norm > 0, norm, 1.0)).max())
            assert rels <= 1e-12, "rank %d %s: mixed-sign probe: max error relative to the row norm %g" % (rank, tag, rels)
            return max(rel, rels)
        rel1 = check(S, "numeric")
        ent_sum = int(ent[::1009].to(torch.int64).sum().item())
        S2 = oracle.Crs(S.nrows, S.ncols, S.row_map, S.entries, 1.0 + 49.0 * rng.random(S.nnz))
        Sd2 = pc.kk.CrsMatrix(S.nrows, S.ncols, Sd.graph.row_map, Sd.graph.entries, torch.from_numpy(S2.values).cuda(), backend=be)
        op.numeric(Sd2, A, Cd)
        assert int(Cd.graph.entries[::1009].to(torch.int64).sum().item()) == ent_sum, "rank %d: numeric reuse changed entries(C)" % rank
        rel2 = check(S2, "numeric reuse")
        print("C4 slab %d of %d (R-MAT s22): %d rows, %d multiplications, nnz(C) %d (%.1f GB): row_map identical to the KKMEM symbolic kernel, rows ascending, C x vs A (A x) %.2e / %.2e (reuse)"
              % (rank, world, S.nrows, mults[rank], nnz, nnz * 12e-9, rel1, rel2))
        del Cd, op, Sd, Sd2, ent, rm
        torch.cuda.empty_cache()


def _check_unstructured(be, A0, name, expect_codes=None):
    import torch
    rng = np.random.default_rng(3)
    x = rng.integers(-20, 20, size=A0.ncols).astype(np.float64)
    A = pc.dev(be, A0)
    xd = torch.from_numpy(x).cuda()
    lens = np.diff(A0.row_map); longest = int(lens.max())
    max_val = float(np.abs(A0.values).max())
    exp = oracle.spmv_serial("N", A0, 1.0, x, 0.0, np.zeros(A0.nrows))
    tol = 10 * EPS * longest * max_val * 20.0
    for handle in (pc.kk.SPMVHandle("SPMV_DEFAULT"), None):
        yd = torch.full((A0.nrows,), float("nan"), dtype=torch.float64, device="cuda")
        args = ("N", 1.0, A, xd, 0.0, yd)
        pc.kk.spmv(handle, *args) if handle is not None else pc.kk.spmv(*args)
        _fspmv(exp, yd.cpu().numpy(), tol, "%s (%s)" % (name, "handle" if handle is not None else "handle-less"))
        if handle is not None and expect_codes is not None:
            assert bool(handle.query("window_codes")) == expect_codes, (name, handle.query("plain_tiles"), handle.query("tiles"))


def test_unstructured_rmat_1e8(be):
    """>= 1e8 nonzeros without structure: R-MAT scale 22, edge factor 32 (hub rows of 1e5 entries, no tile coverable)"""
    R = oracle.rmat(22, 32)
    assert R.nnz >= 100_000_000, R.nnz
    _check_unstructured(be, R, "R-MAT s22 ef32", expect_codes=False)


def test_unstructured_banded_1e8(be):
    """1e7 rows x 12 random columns inside a band of +-20000: tiles are coverable by windows but share no row pattern"""
    A0 = oracle.random_crs(10_000_000, 10_000_000, 12, variance=0, seed=11, bandwidth=20000, sorted_rows=True)
    _check_unstructured(be, A0, "banded random 1e7 x 12", expect_codes=True)


def test_c5_slab_rank3_of_8_against_oracle(be):
    """BASELINE config 5, one rank's piece at its benchmarked size: rank 3 of 8 of the 27-pt 600^3 Laplacian -- rows 81 M .. 108 M (75
    planes of 600 x 600), LOCAL row_map, GLOBAL column indices up to 2.16e8, an x of 216 M doubles -- through the multi-GPU operator
    (kkamd_dist_spmv_*) with a loop-back transport that serves the exchange from the global x.  y of the slab against the Serial
    restatement for the column-range halo (interior / boundary overlap) and the all-gather; the interior view must take the
    row-pattern plan (what the N = 8 bench line's per-GPU time rests on)."""
    import ctypes as C
    import torch
    from kokkos_kernels_amd.dist import DistSpmv
    from dist_loopback import Loopback
    nx = ny = 600; planes = 75; world = 8; rank = 3
    plane = nx * ny; rows = plane * planes; n = rows * world
    offsets = [r * rows for r in range(world + 1)]
    A = pc.kk.laplace_matrix("FE", nx, ny, planes * world, rows=(rank * rows, rows))
    assert A.numRows() == 27_000_000 and A.numCols() == 216_000_000
    rm, ent, val = A.to_host()
    assert int(ent.max()) == offsets[rank + 1] + plane - 1 and int(ent.min()) == offsets[rank] - plane          # global columns, one plane either side
    A0 = oracle.Crs(rows, n, rm.astype(np.int64), ent, val)
    rng = np.random.default_rng(17312837)
    x = rng.integers(-20, 20, size=n).astype(np.float64)
    xd = torch.from_numpy(x).cuda()
    exp = oracle.spmv_serial("N", A0, 1.0, x, 0.0, np.zeros(rows))
    tol = 10 * EPS * 27 * 32.0 * 20.0
    ranges = [(max(0, offsets[p] - plane), min(n, offsets[p + 1] + plane) - 1) for p in range(world)]
    for exchange, mode_name in (("halo", "halo"), ("allgather_collective", "allgather")):      # (the forced collective form: the loop-back transport serves one rank's calls, it cannot run the timed choice of "allgather")
        tr = Loopback(xd, offsets, rank, ranges)
        op = DistSpmv(A, offsets, rank, transport=tr, exchange=exchange)
        assert op.exchange_mode == mode_name, op.exchange_mode
        p_full = C.c_void_p(); pc.kk._capi.check(be.lib, be.lib.kkamd_dist_spmv_x_local(op._op, None, C.byref(p_full)))
        tr.base = p_full.value
        xl = op.x_local(); xl.copy_(xd[offsets[rank]:offsets[rank + 1]])
        y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
        op.apply(1.0, xl, 0.0, y)
        _fspmv(exp, y.cpu().numpy(), tol, "C5 slab rank 3 of 8, exchange %s" % exchange)
        if exchange == "halo":
            assert op.exchange_bytes == 2 * plane * 8 and op.query("parts") == 3
            assert rows - 2 * plane - 16 <= op.interior_rows <= rows - 2 * plane
            # the interior view (73 of the 75 planes) runs the row-pattern plan, like config 2
            assert op.query("part0_rows") == op.interior_rows
            assert op.query("part0_pattern_tiles") >= 0.9 * op.query("part0_tiles"), (op.query("part0_pattern_tiles"), op.query("part0_tiles"))
        else:
            assert op.exchange_bytes == (world - 1) * rows * 8
        y2 = torch.from_numpy(rng.integers(-20, 20, size=rows).astype(np.float64)).cuda(); y2h = y2.cpu().numpy().copy()
        op.apply(2.0, xl, -1.0, y2)                                               # beta != 0 on the same operator
        _fspmv(2.0 * exp - y2h, y2.cpu().numpy(), 10 * EPS * (20.0 + 2 * 27 * 32.0 * 20.0), "C5 slab beta = -1, exchange %s" % exchange)
        del op, tr
        torch.cuda.empty_cache()
