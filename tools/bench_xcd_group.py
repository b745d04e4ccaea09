#!/usr/bin/env python3
"""Grouped XCD order (G consecutive tiles per XCD inside blocks of 8G) on the rank-1, structured and rank-2 kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
def setk(k, v): kk._capi.check(kk.lib(), kk.lib().kkamd_set_default(k.encode(), v))
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
dims = (300, 300, 300)
A = kk.laplace_matrix("FE", *dims)
x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
ref = None
for rep in range(2):
    for g in (0, 2, 4, 8, 16, 32, 1):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("xcd_remap", g)
        t = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))
        if ref is None: ref = y.clone()
        print("spmv   xcd_remap %2d: %.4f ms  maxdiff %.2g" % (g, t, (y - ref).abs().max().item()), flush=True)
for rep in range(2):
    for g in (0, 2, 4, 8, 16, 32):
        setk("struct_group", g)
        print("struct group %2d: %.4f ms" % (g, timeit(lambda: kk.spmv_struct("N", 2, dims, 1.0, A, x, 0.0, y))), flush=True)
setk("struct_group", 0)
X = torch.rand(A.numCols(), 16, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numRows(), 16, dtype=torch.float64, device="cuda")
for rep in range(2):
    for g in (0, 4, 8, 16, 32, 64, 1):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("mv_remap", g)
        print("mv16   mv_remap %2d: %.4f ms" % (g, timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y), it=8)), flush=True)
