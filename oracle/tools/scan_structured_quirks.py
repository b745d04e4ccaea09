#!/usr/bin/env python3
"""Dev tool (not shipped, not run on the GPU box): scans the reference's hand-coded
structured-matrix generator (test_common/KokkosKernels_Test_Structured_Matrix.hpp)
and prints, for every boundary block, the stencil offsets and the BC==1 / BC==0
values, flagging any block that deviates from the rule the C oracle implements
(rule: columns = existing neighbours ascending; BC==1 -> 1.0 on the diagonal and
explicit 0.0 elsewhere).  Used once to discover the quirks listed in
oracle/kk_oracle.c (gen_laplace*)."""
import re, sys
SRC = "/root/reference/test_common/KokkosKernels_Test_Structured_Matrix.hpp"
lines = open(SRC).read().split("\n")
NX, NY = 1000, 1000  # decode offsets: a + b*nx + c*nx*ny
def ev(expr):
    return eval(expr, {"nx": NX, "ny": NY, "nz": 7, "rowIdx": 0})
def decode(off):
    c = round(off / (NX * NY)); off -= c * NX * NY
    b = round(off / NX); off -= b * NX
    return (off, b, c)
blocks = []; cur = None; func = None; cond = None
for ln, s in enumerate(lines, 1):
    m = re.search(r"void operator\(\)\(const (\w+)&", s)
    if m: func = m.group(1) + "@%d" % ln
    m = re.search(r"rowmap\(rowIdx \+ 1\)\s*=", s)
    if m:
        cur = {"func": func, "line": ln, "cols": {}, "v1": {}, "v0": {}, "cond": None}; blocks.append(cur); cond = None
    if cur is None: continue
    m = re.search(r"columns\(rowOffset - (\d+)\)\s*=\s*(.*);", s)
    if m: cur["cols"][int(m.group(1))] = decode(ev(m.group(2)))
    m = re.search(r"if \((.*BC.*)\) \{", s)
    if m and "else" not in s: cond = "bc1"; cur["cond"] = m.group(1)
    elif re.search(r"\} else \{", s): cond = "bc0"
    m = re.search(r"values\(rowOffset - (\d+)\)\s*=\s*([-0-9.]+);", s)
    if m:
        k, v = int(m.group(1)), float(m.group(2))
        if cond == "bc1": cur["v1"][k] = v
        elif cond == "bc0": cur["v0"][k] = v
        else: cur["v1"][k] = v; cur["v0"][k] = v
for b in blocks:
    ks = sorted(b["cols"], reverse=True)
    cols = [b["cols"][k] for k in ks]
    asc = all((c1[2], c1[1], c1[0]) < (c2[2], c2[1], c2[0]) for c1, c2 in zip(cols, cols[1:]))
    v1 = [b["v1"].get(k) for k in ks]; v0 = [b["v0"].get(k) for k in ks]
    exp1 = [1.0 if c == (0, 0, 0) else 0.0 for c in cols]
    quirk = "" if (v1 == exp1 or b["cond"] is None) else "  <-- BC1 QUIRK"
    if not asc: quirk += "  <-- NOT ASCENDING"
    if "-v" in sys.argv or quirk:
        print(b["func"], "line", b["line"], "cond:", b["cond"], quirk)
        print("   cols", cols); print("   bc1 ", v1); print("   bc0 ", v0)
print(len(blocks), "blocks scanned")
