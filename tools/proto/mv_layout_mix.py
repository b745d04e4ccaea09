#!/usr/bin/env python3
"""C3 (27-pt 300^3 x 16, fp64) through the plane-marching kernel with the four combinations of X / Y layout: which side of the
   LayoutLeft penalty is the X fetch and which the Y store."""
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
n, nv = 300, 16
A = kk.laplace_matrix("FE", n, n, n); rows = A.numRows()
Xr = torch.rand(rows, nv, dtype=torch.float64, device="cuda"); Xl = Xr.t().contiguous().t()
Yr = torch.zeros(rows, nv, dtype=torch.float64, device="cuda"); Yl = torch.zeros(nv, rows, dtype=torch.float64, device="cuda").t()
ref = None
for xcol in (1, 0):
    h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("mv4_xcol", xcol)
    kk.spmv(h, "N", 1.0, A, Xr, 0.0, Yr)
    if ref is None: ref = Yr.clone()
    for xn, X in (("right", Xr), ("left", Xl)):
        for yn, Y in (("right", Yr), ("left", Yl)):
            Y.fill_(float("nan")); kk.spmv(h, "N", 1.0, A, X, 0.0, Y)
            print(json.dumps({"mv4_xcol": xcol, "X": xn, "Y": yn, "ms": round(timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y)), 4), "max_abs_diff": float((Y - ref).abs().max())}), flush=True)
