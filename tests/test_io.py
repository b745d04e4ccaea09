"""Matrix file formats: the numpy reader/writers of kokkos-kernels_amd/io.py against the oracle's
line-by-line restatement of the reference reader, the reference's own 6x6 fixtures
(sparse/unit_test/Test_Sparse_IOUtils.hpp:39-54: written as general / symmetric / hermitian / skew-symmetric lower
triangles and read back), scipy.io.mmread where the semantics coincide, and round trips through .mtx / .bin / .crs."""
import os

import numpy as np
import pytest

import oracle
from oracle import kk_io_oracle
import parity_cases as pc
from emu import emu_backend

io = pc.kk.io if hasattr(pc.kk, "io") else None

SYM = np.array([[11, 12, 13, 14, 15, 16], [12, 2, 0, 0, 0, 0], [13, 0, 0, 0, 0, 0], [14, 0, 0, 4, 0, 0], [15, 0, 0, 0, 5, 0],
                [16, 0, 0, 0, 0, 6]], dtype=np.float64)
ASYM = np.array([[1, 0, 0, 9, 0, 0], [0, 2, 0, 0, 0, 0], [0, 0, 0, 0, 0, 8], [0, 0, 0, 4, 0, 0], [0, 7, 0, 0, 5, 0],
                 [0, 0, 0, 0, 0, 6]], dtype=np.float64)


def _dense_to_crs(D):
    rm = [0]; ent = []; val = []
    for r in range(D.shape[0]):
        for c in range(D.shape[1]):
            if D[r, c] != 0:
                ent.append(c); val.append(D[r, c])
        rm.append(len(ent))
    return np.array(rm), np.array(ent, dtype=np.int32), np.array(val)


def _write_mtx(path, rm, ent, val, kind, field="real", comments=("% a comment", "%another")):
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n" % (field, kind))
        for c in comments:
            f.write(c + "\n")
        f.write("%d %d %d\n" % (len(rm) - 1, len(rm) - 1, len(ent)))
        for r in range(len(rm) - 1):
            for j in range(rm[r], rm[r + 1]):
                f.write("%d %d" % (r + 1, ent[j] + 1) + ("" if field == "pattern" else " %r" % float(val[j])) + "\n")


def _same_as_oracle(path, **kw):
    from kokkos_kernels_amd import io as kio
    nr, nc, rm, ent, val = kio.read_mtx(path, **kw)
    onr, onc, orm, oent, oval = kk_io_oracle.read_mtx(path, **kw)
    assert (nr, nc) == (onr, onc) and rm.tolist() == orm and ent.tolist() == oent and val.tolist() == oval
    return nr, nc, rm, ent, val


@pytest.mark.parametrize("kind", ["general", "symmetric", "hermitian", "skew-symmetric"])
def test_reference_fixtures(tmp_path, kind):
    """Test_Sparse_IOUtils.hpp:129-163: the symmetric kinds are written as the lower triangle and must read back as
    the full matrix (skew: the mirrored part negated)."""
    from kokkos_kernels_amd import io as kio
    D = ASYM if kind == "general" else SYM
    rm, ent, val = _dense_to_crs(D if kind == "general" else np.tril(D))
    p = str(tmp_path / ("fixture_%s.mtx" % kind))
    _write_mtx(p, rm, ent, val, kind)
    nr, nc, grm, gent, gval = _same_as_oracle(p, symmetrize=False, remove_diagonal=False, transpose=False)
    exp = D.copy()
    if kind == "skew-symmetric":
        exp = np.tril(D) - np.triu(D.T, 1).T * 0 - np.tril(D, -1).T      # lower part as written, upper part negated
    erm, eent, eval_ = _dense_to_crs(exp)
    assert (nr, nc) == (6, 6) and grm.tolist() == erm.tolist() and gent.tolist() == eent.tolist() and gval.tolist() == eval_.tolist()
    A = kio.read_kokkos_crst_matrix(p, backend=emu_backend.backend())
    r, e, v = A.to_host()
    assert A.numRows() == 6 and A.numCols() == 6 and r.tolist() == erm.tolist() and e.tolist() == eent.tolist() and v.tolist() == eval_.tolist()


def test_reader_options_fields_and_array_format(tmp_path):
    rng = np.random.default_rng(4)
    A0 = oracle.random_crs(30, 30, 5, variance=3, seed=9, sorted_rows=True)           # may contain duplicate columns
    p = str(tmp_path / "g.mtx")
    _write_mtx(p, A0.row_map, A0.entries, A0.values, "general")
    for sym in (False, True):
        for rd in (False, True):
            for tr in (False, True):
                _same_as_oracle(p, symmetrize=sym, remove_diagonal=rd, transpose=tr)
    pp = str(tmp_path / "p.mtx")
    _write_mtx(pp, A0.row_map, A0.entries, A0.values, "symmetric", field="pattern", comments=())
    nr, nc, rm, ent, val = _same_as_oracle(pp, symmetrize=False, remove_diagonal=False, transpose=False)
    assert set(val.tolist()) == {1.0}
    pi = str(tmp_path / "i.mtx")
    _write_mtx(pi, A0.row_map, A0.entries, np.round(A0.values), "general", field="integer")
    _same_as_oracle(pi, symmetrize=False, remove_diagonal=False, transpose=False)
    # dense "array" file: column-major listing, rectangular, zeros are stored entries
    D = rng.integers(-3, 4, size=(4, 7)).astype(np.float64)
    pa = str(tmp_path / "a.mtx")
    with open(pa, "w") as f:
        f.write("%%MatrixMarket matrix array real general\n4 7\n" + "\n".join(repr(float(v)) for v in D.T.reshape(-1)) + "\n")
    nr, nc, rm, ent, val = _same_as_oracle(pa, symmetrize=False, remove_diagonal=False, transpose=False)
    assert (nr, nc) == (4, 7) and np.array_equal(val.reshape(4, 7), D) and rm.tolist() == [0, 7, 14, 21, 28]
    nr, nc, rm, ent, val = _same_as_oracle(pa, symmetrize=False, remove_diagonal=True, transpose=True)
    assert (nr, nc) == (7, 4)
    # scipy agrees on a duplicate-free general file (it sums duplicates; the reference keeps them)
    import scipy.io
    B0 = oracle.laplace2d("FE", 7, 5)
    pb = str(tmp_path / "b.mtx")
    _write_mtx(pb, B0.row_map, B0.entries, B0.values, "general")
    S = scipy.io.mmread(pb).tocsr(); S.sort_indices()
    from kokkos_kernels_amd import io as kio
    nr, nc, rm, ent, val = kio.read_mtx(pb, remove_diagonal=False)
    gold = oracle.Crs(B0.nrows, B0.ncols, B0.row_map, B0.entries.copy(), B0.values.copy()); oracle.sort_crs(gold)
    assert np.array_equal(rm, gold.row_map) and np.array_equal(ent, gold.entries) and np.array_equal(val, gold.values)


def test_header_errors(tmp_path):
    from kokkos_kernels_amd import io as kio
    def w(text):
        p = str(tmp_path / "e.mtx"); open(p, "w").write(text); return p
    with pytest.raises(RuntimeError, match="Line-1"):
        kio.read_mtx(w("%MatrixMarket matrix coordinate real general\n1 1 0\n"))
    with pytest.raises(RuntimeError, match="symmetry"):
        kio.read_mtx(w("%%MatrixMarket matrix coordinate real\n1 1 0\n"))
    with pytest.raises(RuntimeError, match="field"):
        kio.read_mtx(w("%%MatrixMarket matrix coordinate general\n1 1 0\n"))
    with pytest.raises(RuntimeError, match="non-square"):
        kio.read_mtx(w("%%MatrixMarket matrix coordinate real symmetric\n2 3 0\n"))
    with pytest.raises(RuntimeError, match="vector"):
        kio.read_mtx(w("%%MatrixMarket vector coordinate real general\n1 1 0\n"))
    with pytest.raises(RuntimeError, match="cannot be opened"):
        kio.read_mtx(str(tmp_path / "missing.mtx"))
    with pytest.raises(RuntimeError, match="Reader is not available"):
        kio.read_kokkos_crst_matrix(str(tmp_path / "x.foo"), backend=emu_backend.backend())


@pytest.mark.parametrize("ext", [".mtx", ".bin", ".crs"])
@pytest.mark.parametrize("odt", [np.int32, np.int64])
def test_round_trips(tmp_path, ext, odt):
    from kokkos_kernels_amd import io as kio
    be = emu_backend.backend()
    A0 = oracle.laplace3d("FE", 5, 4, 3)
    A = pc.dev(be, A0, odt)
    p = str(tmp_path / ("rt" + ext))
    kio.write_kokkos_crst_matrix(A, p)
    B = kio.read_kokkos_crst_matrix(p, backend=be, offset_dtype=odt)
    r, e, v = B.to_host()
    gold = oracle.Crs(A0.nrows, A0.ncols, A0.row_map, A0.entries.copy(), A0.values.copy())
    if ext == ".mtx":
        oracle.sort_crs(gold)                                   # the MatrixMarket reader orders every row by column
    assert B.numRows() == A0.nrows and B.numCols() == A0.ncols and r.dtype == odt
    assert np.array_equal(r, gold.row_map) and np.array_equal(e, gold.entries) and np.array_equal(v, gold.values)
    # read -> SpMV: the matrix that came back from disk drives the kernels like the original
    x = np.random.default_rng(1).random(A0.ncols)
    y = be.from_numpy(np.zeros(A0.nrows)); pc.kk.spmv("N", 1.0, B, be.from_numpy(x), 0.0, y)
    assert np.allclose(be.to_numpy(y), oracle.spmv_serial("N", A0, 1.0, x, 0.0, np.zeros(A0.nrows)), rtol=1e-13, atol=1e-13)
    if ext != ".mtx":
        rect = pc.dev(be, oracle.random_crs(4, 9, 2, seed=1, sorted_rows=True))
        with pytest.raises(RuntimeError, match="square"):
            kio.write_kokkos_crst_matrix(rect, str(tmp_path / ("rect" + ext)))
