#!/usr/bin/env python3
"""Generates tests/golden/structured_*.npz by INTERPRETING the reference's own
structured-matrix generator source (test_common/KokkosKernels_Test_Structured_Matrix.hpp):
each `operator()(const <Tag>&, idx)` body is mechanically rewritten into Python
(regex: types dropped, view(i) -> view[i], '/' -> '//') and executed over the same
index ranges `compute()` uses (:257-301 for 2-D, :965-1050 for 3-D).  Kokkos is not
available in the build container, so this is the closest thing to "running the
reference" for its generators; it pins oracle/kk_oracle.c:kko_gen_laplace* including
the reference's quirks (oracle section).

Runs only in the build container (needs /root/reference). The .npz outputs are
committed; nothing on the GPU box reads /root/reference.
"""
import re, sys, os
import numpy as np
SRC = "/root/reference/test_common/KokkosKernels_Test_Structured_Matrix.hpp"
HERE = os.path.dirname(os.path.abspath(__file__))
text = open(SRC).read().split("\n")

def body_of(tag_line):
    """lines of the operator() starting at 1-based line tag_line (the 'void operator' line)."""
    out = []; depth = 0; started = False
    for s in text[tag_line - 1:]:
        depth += s.count("{") - s.count("}")
        if started: out.append(s)
        if "{" in s and not started: started = True
        if started and depth == 0: break
    return out[:-1]

def to_python(lines, fname):
    src = ["def %s(idx, G, rowmap, columns, values):" % fname,
           "    globals().update(G)"]
    ind = 1; stmt = ""
    for s in lines:
        s = re.sub(r"//.*", "", s); s = re.sub(r"/\*.*?\*/", "", s).strip()
        if not s: continue
        stmt = (stmt + " " + s).strip()
        if not (stmt.endswith(";") or stmt.endswith("{") or stmt.endswith("}")): continue
        s, stmt = stmt, ""
        s = s.replace("size_type(", "(").replace("||", " or ").replace("&&", " and ")
        s = re.sub(r"(?<![/])/(?![/])", "//", s)
        m = re.match(r"^\} else \{$", s)
        if m: ind -= 1; src.append("    " * ind + "else:"); ind += 1; continue
        m = re.match(r"^if \((.*)\) \{$", s)
        if m: src.append("    " * ind + "if %s:" % m.group(1)); ind += 1; continue
        if s == "}": ind -= 1; continue
        s = s.rstrip(";")
        s = re.sub(r"^(const\s+)?(ordinal_type|size_type|int)\s+", "", s)
        if "=" not in s: continue          # bare declaration: "i, j"
        s = re.sub(r"\b(rowmap|columns|values)\((.*?)\)\s*=", r"\1[\2] =", s)
        src.append("    " * ind + s)
    return "\n".join(src)

def find_ops(lo, hi):
    ops = {}
    for ln in range(lo, hi):
        m = re.search(r"void operator\(\)\(const (\w+)&", text[ln - 1])
        if m: ops[m.group(1)] = ln
    return ops

def run(ops, ranges, G, nrows, nnz):
    rowmap = np.zeros(nrows + 1, dtype=np.int64); columns = np.full(nnz, -7, dtype=np.int64)
    values = np.full(nnz, np.nan)
    for tag, n in ranges:
        ns = {}
        exec(to_python(body_of(ops[tag]), "f"), ns)
        for idx in range(n): ns["f"](idx, G, rowmap, columns, values)
    assert rowmap[-1] == nnz and (columns != -7).all() and not np.isnan(values).any()
    assert (np.diff(rowmap) > 0).all()
    return rowmap, columns, values

def gen2d(stencil, nx, ny, bc):
    ops = find_ops(168, 782); fd = stencil == "FD"
    il, el, cl = (5, 4, 3) if fd else (9, 6, 4)
    G = dict(nx=nx, ny=ny, leftBC=bc[0], rightBC=bc[1], bottomBC=bc[2], topBC=bc[3],
             interiorStencilLength=il, edgeStencilLength=el, cornerStencilLength=cl)
    G["numEntriesPerGridRow"] = (nx - 2) * il + 2 * el
    G["numEntriesBottomRow"] = (nx - 2) * el + 2 * cl
    nnz = (nx - 2) * (ny - 2) * il + (2 * (nx - 2) + 2 * (ny - 2)) * el + 4 * cl
    G["numEntries"] = nnz
    sfx = "FDTag" if fd else "FETag"
    ranges = [("interior" + sfx, (nx - 2) * (ny - 2)), ("xEdge" + sfx, nx - 2), ("yEdge" + sfx, ny - 2), ("corner" + sfx, 1)]
    return run(ops, ranges, G, nx * ny, nnz)

def gen3d(stencil, nx, ny, nz, bc):
    ops = find_ops(852, 3364); fd = stencil == "FD"
    il, fl, el, cl = (7, 6, 5, 4) if fd else (27, 18, 12, 8)
    G = dict(nx=nx, ny=ny, nz=nz, leftBC=bc[0], rightBC=bc[1], frontBC=bc[2], backBC=bc[3], bottomBC=bc[4], topBC=bc[5],
             interiorStencilLength=il, faceStencilLength=fl, edgeStencilLength=el, cornerStencilLength=cl)
    nI = (nx - 2) * (ny - 2) * (nz - 2); xF = (ny - 2) * (nz - 2); yF = (nx - 2) * (nz - 2); zF = (nx - 2) * (ny - 2)
    xE, yE, zE = nx - 2, ny - 2, nz - 2
    nnz = nI * il + 2 * (xF + yF + zF) * fl + 4 * (xE + yE + zE) * el + 8 * cl
    G.update(numEntries=nnz,
             numEntriesPerGridPlane=zF * il + 2 * xE * fl + 2 * yE * fl + 4 * el,
             numEntriesBottomPlane=zF * fl + 2 * xE * el + 2 * yE * el + 4 * cl,
             numEntriesPerGridRow=xE * il + 2 * fl, numEntriesFrontRow=xE * fl + 2 * el,
             numEntriesBottomFrontRow=xE * el + 2 * cl,
             numInterior=nI, numXFace=xF, numYFace=yF, numZFace=zF, numXEdge=xE, numYEdge=yE, numZEdge=zE)
    sfx = "FDTag" if fd else "FETag"
    ranges = [("interior" + sfx, nI), ("xFace" + sfx, xF), ("yFace" + sfx, yF), ("zFace" + sfx, zF),
              ("xEdge" + sfx, xE), ("yEdge" + sfx, yE), ("zEdge" + sfx, zE), ("corner" + sfx, 1)]
    return run(ops, ranges, G, nx * ny * nz, nnz)

if __name__ == "__main__":
    cases = {
        "structured_2d_fd_7x5_bc1": lambda: gen2d("FD", 7, 5, (1, 1, 1, 1)),
        "structured_2d_fd_10x10_bc0": lambda: gen2d("FD", 10, 10, (0, 0, 0, 0)),   # wiki spmv example
        "structured_2d_fe_6x7_bc1": lambda: gen2d("FE", 6, 7, (1, 1, 1, 1)),
        "structured_3d_fd_5x4x6_bc1": lambda: gen3d("FD", 5, 4, 6, (1,) * 6),
        "structured_3d_fe_5x4x6_bc1": lambda: gen3d("FE", 5, 4, 6, (1,) * 6),
        "structured_3d_fe_7x7x7_bc1": lambda: gen3d("FE", 7, 7, 7, (1,) * 6),
        "structured_3d_fd_5x6x7_bc1": lambda: gen3d("FD", 5, 6, 7, (1,) * 6),
        "structured_3d_fe_5x6x7_bc1": lambda: gen3d("FE", 5, 6, 7, (1,) * 6),
    }
    for name, fn in cases.items():
        rm, col, val = fn()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), row_map=rm.astype(np.int64),
                            entries=col.astype(np.int32), values=val)
        print(name, "rows", len(rm) - 1, "nnz", len(col))
