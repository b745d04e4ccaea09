"""On-disk matrix formats of the reference: MatrixMarket (.mtx / .mm), the raw binary dump (.bin)
and the text dump (.crs) of KokkosSparse::Impl::read_kokkos_crst_matrix / write_kokkos_crst_matrix
(sparse/src/KokkosSparse_IOUtils.hpp:488-520, 632-656, 681-694, 741-783, 785-987, 1238-1290).

Host-side plumbing: files are parsed with numpy into CRS arrays and handed to the backend (HBM tensors for the torch
backend).  Semantics follow the reference reader:
  * header: object `matrix`, format `coordinate` | `array`, field `real` | `double` | `integer` | `pattern`
    (`complex` is not supported here: the hot path is real-valued), symmetry `general` | `symmetric` |
    `skew-symmetric` | `hermitian`; comment lines after the banner are skipped;
  * indices are 1-based in the file; `array` files list a dense matrix column by column; `pattern` entries get value 1;
  * a non-general file (or symmetrize=True) adds the mirrored entry of every OFF-DIAGONAL entry (negated for
    skew-symmetric); diagonal entries are kept unless remove_diagonal; transpose swaps the roles of row and column;
  * entries are ordered by (row, column); when symmetrizing, repeated (row, column) pairs collapse to the first one;
  * read_kokkos_crst_matrix uses symmetrize=False, remove_diagonal=False, transpose=False (:1262) and, for .bin/.crs
    (which do not store the column count), numCols = max column + 1 (:1281-1284).
"""
import numpy as np

from .sparse import CrsMatrix, torch_backend


def _parse_banner(line):
    if len(line) < 2 or not line.startswith("%%"):
        raise RuntimeError("Invalid MM file. Line-1")
    if "matrix" not in line:
        if "vector" in line:
            raise RuntimeError('MatrixMarket "vector" is not supported by KokkosKernels read_mtx()')
        raise RuntimeError("MatrixMarket file header is missing the object type.")
    fmt = "coordinate" if "coordinate" in line else ("array" if "array" in line else None)
    if "real" in line or "double" in line:
        field = "real"
    elif "complex" in line:
        raise RuntimeError("scalar_t in read_mtx() incompatible with complex-typed MatrixMarket file.")
    elif "integer" in line:
        field = "integer"
    elif "pattern" in line:
        field = "pattern"
    else:
        field = None
    if "general" in line:
        sym = "general"
    elif "skew-symmetric" in line:
        sym = "skew-symmetric"
    elif "symmetric" in line:
        sym = "symmetric"
    elif "hermitian" in line or "Hermitian" in line:
        sym = "hermitian"
    else:
        sym = None
    if fmt == "array":
        sym = sym or "general"
        if sym != "general":
            raise RuntimeError('array format MatrixMarket file must have general symmetry (optional to include "general")')
    if fmt is None:
        raise RuntimeError("MatrixMarket file header is missing the format.")
    if field is None:
        raise RuntimeError("MatrixMarket file header is missing the field type.")
    if sym is None:
        raise RuntimeError("MatrixMarket file header is missing the symmetry type.")
    return fmt, field, sym


def read_mtx(path, symmetrize=False, remove_diagonal=True, transpose=False, value_dtype=np.float64, offset_dtype=np.int32):
    """KokkosSparse::Impl::read_mtx (same defaults).  Returns (nrows, ncols, row_map, entries, values) as numpy arrays."""
    try:
        f = open(path, "r")
    except OSError:
        raise RuntimeError("File cannot be opened")
    with f:
        fmt, field, sym = _parse_banner(f.readline().rstrip("\n"))
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        head = line.split()
        nr, nc = int(head[0]), int(head[1])
        nnz = int(head[2]) if fmt == "coordinate" else nr * nc
        symmetrize = symmetrize or sym != "general"
        if symmetrize and nr != nc:
            raise RuntimeError("A non-square matrix cannot be symmetrized.")
        if fmt == "array":
            if symmetrize:
                raise RuntimeError("array format MatrixMarket file cannot be symmetrized.")
            if field == "pattern":
                raise RuntimeError("array format MatrixMarket file can't have \"pattern\" field type.")
        ncols_txt = {("coordinate", "pattern"): 2, ("coordinate", "real"): 3, ("coordinate", "integer"): 3}.get((fmt, field), 1)
        if nnz:
            data = np.loadtxt(f, dtype=np.float64, max_rows=nnz, ndmin=2, usecols=range(ncols_txt), comments=None)
        else:
            data = np.zeros((0, ncols_txt))
    if data.shape[0] != nnz:
        raise RuntimeError("MatrixMarket file ends before all %d entries were read" % nnz)
    if fmt == "array":
        idx = np.arange(nnz, dtype=np.int64)
        src, dst, w = idx % nr, idx // nr, data[:, 0]
    else:
        src, dst = data[:, 0].astype(np.int64) - 1, data[:, 1].astype(np.int64) - 1
        w = np.ones(nnz) if field == "pattern" else data[:, 2]
    if transpose:
        src, dst = dst, src
        nr, nc = nc, nr
    diag = src == dst
    keep = ~diag if remove_diagonal else np.ones(nnz, dtype=bool)
    s, d, v = src[keep], dst[keep], w[keep]
    order_hint = np.nonzero(keep)[0] * 2                         # position in the reference's edge list
    if symmetrize:
        off = ~diag
        flip = -w[off] if sym == "skew-symmetric" else w[off]
        s = np.concatenate([s, dst[off]]); d = np.concatenate([d, src[off]]); v = np.concatenate([v, flip])
        order_hint = np.concatenate([order_hint, np.nonzero(off)[0] * 2 + 1])
    # (row, column) order; ties keep the order in which the reference appended the edges
    perm = np.lexsort((order_hint, d, s))
    s, d, v = s[perm], d[perm], v[perm]
    if symmetrize and s.size:
        first = np.ones(s.size, dtype=bool)
        first[1:] = (s[1:] != s[:-1]) | (d[1:] != d[:-1])
        s, d, v = s[first], d[first], v[first]
    row_map = np.zeros(nr + 1, dtype=np.int64)
    np.add.at(row_map, s + 1, 1)
    row_map = np.cumsum(row_map).astype(offset_dtype)
    return nr, nc, row_map, d.astype(np.int32), v.astype(value_dtype)


def write_matrix_mtx(path, nrows, ncols, row_map, entries, values):
    """KokkosSparse::Impl::write_matrix_mtx (:632-656): coordinate real general, 17 significant digits."""
    row_map = np.asarray(row_map); entries = np.asarray(entries); values = np.asarray(values)
    rows = np.repeat(np.arange(nrows, dtype=np.int64), np.diff(row_map.astype(np.int64)))
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (nrows, ncols, len(entries)))
        for r, c, v in zip(rows + 1, entries.astype(np.int64) + 1, values):
            f.write("%d %d %.17e\n" % (r, c, v))


def write_graph_bin(path, row_map, entries, values):
    """KokkosSparse::Impl::write_graph_bin (:488-500): nv, ne, row_map, entries, values in the arrays' native types."""
    row_map = np.ascontiguousarray(row_map); entries = np.ascontiguousarray(entries); values = np.ascontiguousarray(values)
    with open(path, "wb") as f:
        f.write(np.asarray([len(row_map) - 1], dtype=entries.dtype).tobytes())
        f.write(np.asarray([len(entries)], dtype=row_map.dtype).tobytes())
        f.write(row_map.tobytes()); f.write(entries.tobytes()); f.write(values.tobytes())


def read_graph_bin(path, offset_dtype=np.int32, ordinal_dtype=np.int32, value_dtype=np.float64):
    """KokkosSparse::Impl::read_graph_bin (:681-694); the three types must be the ones the file was written with."""
    with open(path, "rb") as f:
        nv = int(np.frombuffer(f.read(np.dtype(ordinal_dtype).itemsize), dtype=ordinal_dtype)[0])
        ne = int(np.frombuffer(f.read(np.dtype(offset_dtype).itemsize), dtype=offset_dtype)[0])
        row_map = np.frombuffer(f.read(np.dtype(offset_dtype).itemsize * (nv + 1)), dtype=offset_dtype).copy()
        entries = np.frombuffer(f.read(np.dtype(ordinal_dtype).itemsize * ne), dtype=ordinal_dtype).copy()
        values = np.frombuffer(f.read(np.dtype(value_dtype).itemsize * ne), dtype=value_dtype).copy()
    if len(row_map) != nv + 1 or len(entries) != ne or len(values) != ne:
        raise RuntimeError("%s: truncated .bin file" % path)
    return nv, row_map, entries, values


def write_graph_crs(path, row_map, entries, values=None):
    """KokkosSparse::Impl::write_graph_crs (:503-520) writes the graph only; values are appended here as one more line
    so that read_graph_crs (:718-738), which does read them, gets them back."""
    row_map = np.asarray(row_map); entries = np.asarray(entries)
    nv = len(row_map) - 1
    with open(path, "w") as f:
        f.write("%d %d\n" % (nv, len(entries)))
        f.write(" ".join(str(int(v)) for v in row_map) + " \n")
        for i in range(nv):
            f.write(" ".join(str(int(c)) for c in entries[int(row_map[i]):int(row_map[i + 1])]) + " \n")
        if values is not None:
            f.write(" ".join("%.17e" % float(v) for v in values) + "\n")


def read_graph_crs(path, offset_dtype=np.int32, value_dtype=np.float64):
    tok = open(path, "r").read().split()
    nv, ne = int(tok[0]), int(tok[1])
    row_map = np.asarray(tok[2:2 + nv + 1], dtype=np.int64).astype(offset_dtype)
    entries = np.asarray(tok[3 + nv:3 + nv + ne], dtype=np.int64).astype(np.int32)
    rest = tok[3 + nv + ne:3 + nv + 2 * ne]
    values = np.asarray(rest, dtype=np.float64).astype(value_dtype) if len(rest) == ne else np.zeros(ne, dtype=value_dtype)
    return nv, row_map, entries, values


def read_kokkos_crst_matrix(path, backend=None, offset_dtype=np.int32, value_dtype=np.float64):
    """KokkosSparse::Impl::read_kokkos_crst_matrix<crsMat_t>(filename) (:1238-1290) -> CrsMatrix on the backend."""
    be = backend or torch_backend()
    if path.endswith(".mtx") or path.endswith(".mm"):
        nr, nc, rm, ent, val = read_mtx(path, False, False, False, value_dtype, offset_dtype)
    elif path.endswith(".bin"):
        nr, rm, ent, val = read_graph_bin(path, offset_dtype, np.int32, value_dtype)
        nc = int(ent.max()) + 1 if len(ent) else 0
    elif path.endswith(".crs"):
        nr, rm, ent, val = read_graph_crs(path, offset_dtype, value_dtype)
        nc = int(ent.max()) + 1 if len(ent) else 0
    else:
        raise RuntimeError("Reader is not available")
    return CrsMatrix.from_host(nr, nc, rm, ent, val, offset_dtype=offset_dtype, backend=be)


def write_kokkos_crst_matrix(A, path):
    """KokkosSparse::Impl::write_kokkos_crst_matrix (:741-783): .mtx / .mm, or .bin / .crs for square matrices."""
    rm, ent, val = A.to_host()
    if path.endswith(".mtx") or path.endswith(".mm"):
        write_matrix_mtx(path, A.numRows(), A.numCols(), rm, ent, val)
        return
    if A.numRows() != A.numCols():
        raise RuntimeError("For formats other than MatrixMarket (suffix .mm or .mtx),\nwrite_kokkos_crst_matrix only supports square matrices")
    if path.endswith(".bin"):
        write_graph_bin(path, rm, ent, val)
    elif path.endswith(".crs"):
        write_graph_crs(path, rm, ent, val)
    else:
        raise RuntimeError("write_kokkos_crst_matrix: File extension on %s does not correspond to a known format" % path)
