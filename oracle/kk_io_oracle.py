"""ORACLE (test infrastructure only) for the MatrixMarket reader: a line-by-line restatement of
KokkosSparse::Impl::read_mtx (sparse/src/KokkosSparse_IOUtils.hpp:785-987) with plain Python loops -- an edge list
that is appended to entry by entry, sorted by (src, dst) and compressed row by row exactly as the reference does.
Pinned by the reference's own fixtures (sparse/unit_test/Test_Sparse_IOUtils.hpp:39-54,129-163): see
tests/test_io.py.  Real / integer / pattern fields only (the hot path is real-valued)."""


def read_mtx(path, symmetrize=False, remove_diagonal=True, transpose=False):
    with open(path, "r") as mmf:
        fline = mmf.readline().rstrip("\n")
        if len(fline) < 2 or fline[0] != "%" or fline[1] != "%":                     # :797-799
            raise RuntimeError("Invalid MM file. Line-1")
        if "matrix" not in fline:                                                     # :808-813
            raise RuntimeError("unsupported object")
        mtx_format = "COORDINATE" if "coordinate" in fline else ("ARRAY" if "array" in fline else None)   # :815-821
        if "real" in fline or "double" in fline: mtx_field = "REAL"                   # :823-858
        elif "integer" in fline: mtx_field = "INTEGER"
        elif "pattern" in fline: mtx_field = "PATTERN"
        else: raise RuntimeError("unsupported field")
        if "general" in fline: mtx_sym = "GENERAL"                                    # :860-870
        elif "skew-symmetric" in fline: mtx_sym = "SKEW_SYMMETRIC"
        elif "symmetric" in fline: mtx_sym = "SYMMETRIC"
        elif "hermitian" in fline or "Hermitian" in fline: mtx_sym = "HERMITIAN"
        else: mtx_sym = "GENERAL" if mtx_format == "ARRAY" else None
        if mtx_format is None or mtx_sym is None:
            raise RuntimeError("incomplete header")
        while True:                                                                   # :885-888
            fline = mmf.readline()
            if not fline.startswith("%"):
                break
        parts = fline.split()
        nr, nc = int(parts[0]), int(parts[1])
        nnz = int(parts[2]) if mtx_format == "COORDINATE" else nr * nc               # :892-896
        symmetrize = symmetrize or mtx_sym != "GENERAL"                               # :898
        if symmetrize and nr != nc:
            raise RuntimeError("A non-square matrix cannot be symmetrized.")
        edges = []
        for i in range(nnz):                                                          # :915-957
            toks = mmf.readline().split()
            if mtx_format == "ARRAY":
                s, d = i % nr + 1, i // nr + 1
                w = float(toks[0])
            else:
                s, d = int(toks[0]), int(toks[1])
                w = 1.0 if mtx_field == "PATTERN" else float(toks[2])
            src, dst = (s - 1, d - 1) if not transpose else (d - 1, s - 1)
            if src == dst:
                if not remove_diagonal:
                    edges.append((src, dst, w))
                continue
            edges.append((src, dst, w))
            if symmetrize:
                edges.append((dst, src, -w if mtx_sym == "SKEW_SYMMETRIC" else w))   # symmetryFlip :604-610
    edges.sort(key=lambda e: (e[0], e[1]))                                            # :959 (Edge::operator< :74-78)
    if transpose:
        nr, nc = nc, nr
    xadj, adj, ew = [0] * (nr + 1), [], []
    eind = 0
    for i in range(nr):                                                               # :972-985
        xadj[i] = len(adj)
        is_first = True
        while eind < len(edges) and edges[eind][0] == i:
            if is_first or not symmetrize or eind == 0 or edges[eind - 1][1] != edges[eind][1]:
                adj.append(edges[eind][1]); ew.append(edges[eind][2])
            is_first = False
            eind += 1
    xadj[nr] = len(adj)
    return nr, nc, xadj, adj, ew
