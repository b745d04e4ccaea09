#!/usr/bin/env python3
"""SpMV_MV of C3 (27-pt FE 300^3 x 16, both layouts) through the analysed handle for beta = 0 / 1 / -0.5 (the reference's perf driver
   times beta = 1).  Usage: python tools/bench_mv_beta.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
A = kk.laplace_matrix("FE", n, n, n)
for layout in ("right", "left"):
    if layout == "right":
        X = torch.rand(A.numCols(), 16, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numRows(), 16, dtype=torch.float64, device="cuda")
    else:
        X = torch.rand(16, A.numCols(), dtype=torch.float64, device="cuda").t(); Y = torch.zeros(16, A.numRows(), dtype=torch.float64, device="cuda").t()
    h = kk.SPMVHandle("SPMV_DEFAULT")
    out = {"matrix": "27-pt FE %d^3 x 16" % n, "layout": layout, "library": os.environ.get("KKAMD_LIBRARY", "libkkamd.so")}
    for beta in (0.0, 1.0, -0.5):
        out["beta_%g_ms" % beta] = round(min(timeit(lambda: kk.spmv(h, "N", 1.0, A, X, beta, Y)) for _ in range(2)), 4)
    print(json.dumps(out), flush=True)
