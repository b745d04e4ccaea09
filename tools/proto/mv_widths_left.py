#!/usr/bin/env python3
"""27-pt 300^3 x 4 / 8 / 12 / 16 / 24 right-hand sides, column-major X and Y, plane-marching kernel with the column-wise X piece
   order (mv4_xcol 1) and without (0)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it
A = kk.laplace_matrix("FE", 300, 300, 300); rows = A.numRows()
for nv in (4, 8, 12, 16, 24):
    Xl = torch.rand(nv, rows, dtype=torch.float64, device="cuda").t(); Yl = torch.zeros(nv, rows, dtype=torch.float64, device="cuda").t()
    out = {"nvec": nv}; ref = None
    for xcol in (0, 1):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("mv4_xcol", xcol)
        Yl.fill_(float("nan")); kk.spmv(h, "N", 1.0, A, Xl, 0.0, Yl)
        if ref is None: ref = Yl.clone()
        out["xcol_%d_ms" % xcol] = round(timeit(lambda: kk.spmv(h, "N", 1.0, A, Xl, 0.0, Yl)), 4); out["max_abs_diff"] = float((Yl - ref).abs().max())
    print(json.dumps(out), flush=True)
