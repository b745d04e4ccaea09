"""CPU tests that PIN the oracle (oracle/) against the reference's own golden material:
generator matrices obtained by interpreting the reference source, the literal fixtures of its
unit tests, its known-answer tests, and scipy as an independent cross-check."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = len(g["row_map"]) - 1
    return oracle.Crs(n, n, g["row_map"], g["entries"], g["values"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "structured_*.npz"))))
def test_generators_match_interpreted_reference(path):
    g = np.load(path)
    kind, stencil, dims, bc = os.path.basename(path)[len("structured_"):-4].split("_")
    dims = [int(v) for v in dims.split("x")]
    if kind == "2d":
        A = oracle.laplace2d(stencil.upper(), *dims, bc=(1,) * 4 if bc == "bc1" else (0,) * 4)
    else:
        A = oracle.laplace3d(stencil.upper(), *dims)
    assert np.array_equal(A.row_map, g["row_map"])
    assert np.array_equal(A.entries, g["entries"])      # includes reference quirks Q1-Q6
    assert np.array_equal(A.values, g["values"])


def test_nnz_formulas_of_the_configs():
    L = oracle.lib()
    assert L.kko_laplace2d_nnz(0, 1000, 1000) == 4_996_000          # C1
    assert L.kko_laplace3d_nnz(1, 300, 300, 300) == 724_150_792     # C2/C3
    assert L.kko_laplace3d_nnz(1, 600, 600, 600) == 5_812_581_592   # C5
    A = oracle.laplace3d("FE", 6, 5, 4)
    assert A.nnz == L.kko_laplace3d_nnz(1, 6, 5, 4)


def test_github_issue_101_known_answer():
    # sparse/unit_test/Test_Sparse_spmv.hpp:823-961: 1x2 matrix [1, eps_f/2], x = 1 -> y == 1 + eps_f/2 EXACTLY
    eps_f = float(np.finfo(np.float32).eps)
    expected = 1.0 + eps_f / 2.0
    rm = np.array([0, 2]); ent = np.array([0, 1])
    A = oracle.Crs(1, 2, rm, ent, np.array([1.0, eps_f / 2.0]))
    y = oracle.spmv_serial("N", A, 1.0, np.ones(2), 0.0, np.zeros(1))
    assert y[0] == expected
    Af = oracle.Crs(1, 2, rm, ent, np.array([1.0, eps_f / 2.0], dtype=np.float32))
    y = oracle.spmv_serial("N", Af, 1.0, np.ones(2), 0.0, np.zeros(1))
    assert y[0] == expected
    for nv in range(1, 23):
        for order in ("C", "F"):
            X = np.ones((2, nv), order=order); Y = np.zeros((1, nv), order=order)
            oracle.spmv_mv_serial("N", A, 1.0, X, 0.0, Y)
            assert (Y == expected).all()


def test_wiki_spmv_example():
    # example/wiki/sparse/KokkosSparse_wiki_spmv.cpp:66-95: 10x10 2-D FD, all BC = 0, x = 1, y0 = 2,
    # alpha = beta = 1  =>  y == 2 exactly (rows sum to zero)
    A = oracle.laplace2d("FD", 10, 10, bc=(0, 0, 0, 0))
    y = oracle.spmv_serial("N", A, 1.0, np.ones(100), 1.0, np.full(100, 2.0))
    assert (y == 2.0).all()


def test_laplacian_row_sums():
    A = oracle.laplace3d("FE", 9, 8, 7)
    y = oracle.spmv_serial("N", A, 1.0, np.ones(A.ncols), 0.0, np.empty(A.nrows))
    lens = np.diff(A.row_map)
    assert (y[lens == 27] == 0.0).all()        # interior 27-pt rows sum to 0 (32 - 12*2 - 8*1)
    assert (y[lens < 27] == 1.0).all()         # boundary rows are identity rows (Q3 moves the 1.0, sum stays 1)


@pytest.mark.parametrize("mode", ["N", "C", "T", "H"])
@pytest.mark.parametrize("alpha,beta", [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0), (1.0, 1.0), (-1.0, 2.5), (2.5, -1.0)])
def test_spmv_serial_vs_sequential_and_scipy(mode, alpha, beta):
    A = oracle.random_crs(1000, 900, 13, variance=6, seed=3)
    rng = np.random.default_rng(5)
    trans = mode in "TH"
    x = rng.random(A.nrows if trans else A.ncols)
    y0 = rng.random(A.ncols if trans else A.nrows)
    y0[::19] = np.nan if beta == 0.0 else y0[::19]      # NaN overwrite rule, Test_Sparse_spmv.hpp:394-408,434-436
    y1 = oracle.spmv_serial(mode, A, alpha, x, beta, y0.copy())
    y2 = oracle.spmv_sequential(mode, A, alpha, x, beta, y0.copy())
    S = A.to_scipy()
    ys = alpha * ((S.T if trans else S) @ x) + (0.0 if beta == 0.0 else beta * y0)
    tol = oracle.spmv_max_error(A, alpha, beta)
    assert not np.isnan(y1).any()
    assert np.abs(y1 - y2).max() <= tol
    assert np.abs(y1 - ys).max() <= tol


@pytest.mark.parametrize("dims", pc_dims := [(10,), (25, 21), (20, 25), (22, 22), (20, 20, 20), (22, 22, 22), (25, 10, 20),
                                             (10, 20, 25), (10, 24, 20), (3, 3), (3, 3, 3), (2, 9)])
def test_spmv_struct_restatement_vs_sequential(dims):
    """The reference's own check (Test_Sparse_spmv.hpp:263-296): spmv_struct against sequential_spmv on its generator's
    matrices, all its sizes (:1096-1104).  Equality also proves the restated exterior index maps cover every
    non-interior row exactly once."""
    rng = np.random.default_rng(13718)
    for st in ((1,) if len(dims) == 1 else (1, 2)):
        if len(dims) == 1:
            A = oracle.laplace1d(dims[0])
        else:
            A = (oracle.laplace2d if len(dims) == 2 else oracle.laplace3d)("FD" if st == 1 else "FE", *dims)
        x = rng.random(A.ncols); y0 = rng.random(A.nrows)
        for mode in "NCTH":
            for alpha, beta in ((1.0, 0.0), (0.0, 1.0), (1.0, 1.0)):
                y = oracle.spmv_struct(mode, st, dims, A, alpha, x, beta, y0.copy())
                ye = oracle.spmv_sequential(mode, A, alpha, x, beta, y0.copy())
                tol = 10 * np.finfo(float).eps * (abs(beta) + abs(alpha) * 27 * 26)
                assert np.abs(y - ye).max() <= tol


def test_laplace1d_matches_reference_fill():
    A = oracle.laplace1d(6)
    assert A.row_map.tolist() == [0, 2, 5, 8, 11, 14, 16]
    assert A.entries.tolist() == [0, 1, 0, 1, 2, 1, 2, 3, 2, 3, 4, 3, 4, 5, 4, 5]
    assert A.values.tolist() == [1, 0, -1, 2, -1, -1, 2, -1, -1, 2, -1, -1, 2, -1, 0, 1]
    B = oracle.laplace1d(4, bc=(0, 0))
    assert B.values.tolist() == [1, -1, -1, 2, -1, -1, 2, -1, -1, 1]


def test_spmv_mv_serial_layouts():
    A = oracle.random_crs(300, 280, 9, variance=4, seed=11)
    rng = np.random.default_rng(2)
    for nv in (1, 5, 16, 17):
        for mode in ("N", "T"):
            nin, nout = (A.nrows, A.ncols) if mode == "T" else (A.ncols, A.nrows)
            X = np.asfortranarray(rng.random((nin, nv))); Y0 = rng.random((nout, nv))
            Y = oracle.spmv_mv_serial(mode, A, 2.5, X, -1.0, Y0.copy())
            S = A.to_scipy()
            ref = 2.5 * ((S.T if mode == "T" else S) @ X) - Y0
            assert np.abs(Y - ref).max() <= oracle.spmv_max_error(A, 2.5, 1.0) * 4


def test_spgemm_vs_scipy_and_issue402():
    A = _load("matrix_issue402")
    At = oracle.transpose(A)
    oracle.sort_crs(A); oracle.sort_crs(At)
    C = oracle.spgemm(A, At)
    # scipy's csr_matmat prunes sums that cancel to exactly 0.0; the reference keeps them (structure is
    # the union of B rows).  So: structure from an all-ones product, values where scipy kept them.
    P, Pt = A.to_scipy().copy(), At.to_scipy().copy()
    P.data[:] = 1.0; Pt.data[:] = 1.0
    S = (P @ Pt).tocsr(); S.sort_indices()
    assert np.array_equal(C.row_map, S.indptr) and np.array_equal(C.entries, S.indices)
    V = (A.to_scipy() @ At.to_scipy()).tocsr()
    dense_rows = (0, 1, 907, 1812)
    for i in dense_rows:
        ref = np.zeros(C.ncols); ref[V.indices[V.indptr[i]:V.indptr[i + 1]]] = V.data[V.indptr[i]:V.indptr[i + 1]]
        got = np.zeros(C.ncols); got[C.entries[C.row_map[i]:C.row_map[i + 1]]] = C.values[C.row_map[i]:C.row_map[i + 1]]
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-18)
    mults, mx = oracle.spgemm_mults(A, At)
    assert mults >= C.nnz and mx >= np.diff(C.row_map).max()


@pytest.mark.parametrize("m,n,k,nnzA,nnzB", [
    (0, 0, 0, 0, 0), (0, 12, 5, 0, 20), (10, 10, 0, 20, 0), (10, 0, 10, 0, 0),
    (10, 10, 10, 0, 0), (10, 10, 10, 20, 0), (10, 10, 10, 0, 20), (50, 40, 30, 200, 160)])
def test_spgemm_degenerate_shapes(m, n, k, nnzA, nnzB):
    # the seven empties of sparse/unit_test/Test_Sparse_spgemm.hpp:491-504 plus a small dense-ish case
    A = oracle.random_crs(m, n, nnzA // m if m else 0, seed=1, sorted_rows=True)
    B = oracle.random_crs(n, k, nnzB // n if n else 0, seed=2, sorted_rows=True)
    C = oracle.spgemm(A, B)
    assert C.row_map[0] == 0 and len(C.row_map) == m + 1
    if m and n and k:
        S = (A.to_scipy() @ B.to_scipy()).tocsr(); S.sort_indices()
        # random_crs may hold duplicate columns; scipy sums them the same way (structure is a set union)
        assert np.array_equal(C.row_map, S.indptr) and np.array_equal(C.entries, S.indices)
        assert np.allclose(C.values, S.data, rtol=1e-12)
    else:
        assert C.nnz == 0


def test_sort_crs_and_first_touch_order():
    A = _load("crs_10x10")
    C = oracle.spgemm(A, A, sort=False)
    Cs = oracle.spgemm(A, A, sort=True)
    assert np.array_equal(C.row_map, Cs.row_map)
    for i in range(C.nrows):
        s, e = C.row_map[i], C.row_map[i + 1]
        assert sorted(C.entries[s:e]) == list(Cs.entries[s:e])
        order = np.argsort(C.entries[s:e], kind="stable")
        assert np.array_equal(C.values[s:e][order], Cs.values[s:e])


def test_sort_and_merge_vs_scipy():
    import scipy.sparse as sp
    A = oracle.random_crs(200, 50, 30, variance=20, seed=12)                 # many duplicate columns per row
    S = sp.csr_matrix((A.values.copy(), A.entries.copy(), A.row_map.copy()), shape=(A.nrows, A.ncols))
    S.sum_duplicates(); S.sort_indices()
    G = oracle.sort_and_merge(oracle.Crs(A.nrows, A.ncols, A.row_map, A.entries.copy(), A.values.copy()))
    assert np.array_equal(G.row_map, S.indptr) and np.array_equal(G.entries, S.indices) and np.allclose(G.values, S.data, rtol=1e-13)


def test_rmat_is_deterministic_and_well_formed():
    A = oracle.rmat(10, 8, seed=7)
    B = oracle.rmat(10, 8, seed=7)
    assert np.array_equal(A.entries, B.entries) and np.array_equal(A.values, B.values)
    assert A.nrows == 1024 and A.nnz <= 8 * 1024 and (A.values >= 1).all() and (A.values < 50).all()
    for i in range(0, A.nrows, 37):
        r = A.entries[A.row_map[i]:A.row_map[i + 1]]
        assert (np.diff(r) > 0).all()
    # skew: the R-MAT head rows are much heavier than the mean
    assert np.diff(A.row_map).max() > 8 * np.diff(A.row_map).mean()


def test_omp_baseline_matches_serial():
    A = oracle.laplace3d("FE", 20, 18, 16)
    rng = np.random.default_rng(0)
    x = rng.random(A.ncols); y0 = rng.random(A.nrows)
    ys = oracle.spmv_serial("N", A, 1.0, x, 1.0, y0.copy())
    yo = oracle.spmv_omp(A.row_map, A.entries, A.values, 1.0, x, 1.0, y0.copy())
    yo32 = oracle.spmv_omp(A.row_map.astype(np.int32), A.entries, A.values, 1.0, x, 1.0, y0.copy())
    tol = oracle.spmv_max_error(A, 1.0, 1.0, max_val=32.0)
    assert np.abs(ys - yo).max() <= tol and np.abs(ys - yo32).max() <= tol
    X = rng.random((A.ncols, 16)); Y = np.zeros((A.nrows, 16))
    oracle.spmv_mv_omp(A.row_map.astype(np.int32), A.entries, A.values, 1.0, X, 0.0, Y)
    Yr = oracle.spmv_mv_serial("N", A, 1.0, X, 0.0, np.zeros((A.nrows, 16)))
    assert np.abs(Y - Yr).max() <= tol


def test_spgemm_kkmem_omp_matches_debug():
    """the OpenMP KKMEM port (CPU baseline of the SpGEMM measurements) against SPGEMM_DEBUG + sort: same structure, values to rounding"""
    for A0, B0 in ((oracle.rmat(9, 8), oracle.rmat(9, 8)),
                   (oracle.random_crs(300, 2000, 12, variance=8, seed=1), oracle.random_crs(2000, 700, 9, variance=7, seed=2)),
                   (oracle.laplace3d("FE", 9, 8, 7),) * 2):
        t = {}
        G = oracle.spgemm(A0, B0)
        K = oracle.spgemm_kkmem_omp(A0, B0, timings=t)
        ok, msg = oracle.is_same_matrix(K, G, 1e-12)
        assert ok, msg
        assert set(t) == {"symbolic_s", "numeric_s", "sort_s"}


# ---- merge matrix: the reference's hand-computed diagonals (sparse/unit_test/Test_Sparse_MergeMatrix.hpp:136-575) ----------------
MERGE_CASES = [
    ([0, 0, 0, 0], [0, 1, 2, 3], None, 0),                                                  # all zero
    ([1, 2, 3, 4], [0, 0, 0, 0], None, 1),                                                  # all one
    ([1, 2, 3, 4], [0, 1, 2, 3], [[], [1], [1, 0], [1, 1, 0], [1, 1, 0, 0], [1, 1, 0], [1, 0], [1]], None),          # case 1 (:161-181)
    ([1, 2, 9], [0, 2, 2, 8, 8, 8], [[], [1], [1, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0], [1]], None),   # case 2 (:184-203)
    ([-1, 9, 9], [0, 2, 7], [[], [0], [1, 0], [1, 1, 0], [1, 1], [1]], None),                 # case 3 (:206-223)
    ([1, 6, 6], [-3, -1, 7], [[], [1], [1, 1], [1, 1, 0], [1, 0], [0]], None),                # case 4 (:226-245)
    ([-3, -2, 2], [-2, 0, 1], [[], [0], [0, 0], [1, 0, 0], [1, 0], [1]], None),               # case 5 (expectations at :339-344)
]


@pytest.mark.parametrize("a,b,diags,const", MERGE_CASES)
def test_merge_matrix_known_diagonals(a, b, diags, const):
    n = len(a) + len(b) - 1
    for d in range(n if diags is None else len(diags)):
        got = oracle.merge_matrix_diagonal(a, b, d)
        if diags is None:
            assert got == [const] * len(got)
        else:
            assert got == diags[d], (d, got, diags[d])
        # diagonal_search = the first entry that is not 1, as a position (ai, bi) with ai + bi = d
        ai, bi = oracle.diagonal_search(a, b, d)
        ones = 0
        while ones < len(got) and got[ones] == 1: ones += 1
        assert ai + bi == d and (d == 0 or bi == (ones if d < len(a) else d + ones - len(a)))


def test_merge_path_split_equals_nnz_split_descriptors():
    """The reference's merge-path SpMV cuts (rows + nnz) with diagonal_search over a = row ends, b = iota(nnz)
    (sparse/impl/KokkosSparse_spmv_impl_merge.hpp:100-160); the nnz-split tiles here cut nnz alone.  On the diagonal that passes
    through a tile boundary p (all rows that end at or before p, p nonzeros) the search must land exactly on it -- which pins the
    closed forms the tile descriptors are checked against (test_emu_spmv.py::test_tile_descriptors_against_merge_path)."""
    rng = np.random.default_rng(5)
    lens = np.concatenate([rng.integers(0, 9, 60), [0, 0, 40, 0, 1]])
    rm = np.concatenate([[0], np.cumsum(lens)])
    ends, nnz = list(rm[1:]), int(rm[-1])
    for tile in (7, 16, 64):
        for p in range(0, nnz + 1, tile):
            f = int(np.searchsorted(rm[1:], p, side="right"))           # rows that end at or before p
            ai, bi = oracle.diagonal_search(ends, list(range(nnz)), f + p)
            assert (ai, bi) == (f, p), (tile, p, ai, bi, f)
