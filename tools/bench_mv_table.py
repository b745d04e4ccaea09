#!/usr/bin/env python3
"""SpMV_MV (16 right-hand sides, fp64, row-major and column-major X / Y) off the benchmark matrix: the matrices of
tools/bench_plan_table.py through the default analysed handle, with the kernel the library chose.  One JSON line per matrix."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_plan_table as pt
kk = pt.kk
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 16
only = sys.argv[2] if len(sys.argv) > 2 else ""
for name, A in pt.matrices(only):
    rows, cols, nnz = A.numRows(), A.numCols(), A.nnz()
    alg = nnz * 12 + (rows + 1) * 4 + (cols + rows) * nv * 8
    X = torch.rand(cols, nv, dtype=torch.float64, device="cuda"); Y = torch.zeros(rows, nv, dtype=torch.float64, device="cuda")
    res = {}
    for lay in ("right", "left"):
        Xl = X if lay == "right" else X.t().contiguous().t()
        Yl = Y if lay == "right" else torch.zeros(nv, rows, dtype=torch.float64, device="cuda").t()
        h = kk.SPMVHandle("SPMV_DEFAULT")
        for kv in os.environ.get("KK_KNOBS", "").split(","):
            if kv: h.set(kv.split("=")[0], int(kv.split("=")[1]))
        ms = pt.timeit(lambda: kk.spmv(h, "N", 1.0, A, Xl, 0.0, Yl), it=10)
        res[lay] = {"ms": round(ms, 4), "frac_8TBps": round(alg / ms / 1e6 / 8000, 3), "kernel": "mv4" if h.query("mv4_workgroups") else ("mv5" if h.query("mv5_tiles") else ("mv6" if h.query("mv6_chunks") else "mv2")), "mv5_fill": h.query("mv5_fill_permille") / 1000, "mv5_other_rows": h.query("mv5_other_rows"), "mv_order": h.query("mv_order"), "long_rows": h.query("mv_long_rows")}
        del h
    print(json.dumps({"matrix": name, "nvec": nv, "rows": rows, "nnz": nnz, "alg_GB": round(alg / 1e9, 3), "knobs": os.environ.get("KK_KNOBS", ""), **res}), flush=True)
    del A, X, Y
