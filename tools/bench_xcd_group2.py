#!/usr/bin/env python3
"""Grouped XCD order, second sweep: plain-entries kernel, larger groups for spmv_struct, other RHS counts, an unstructured matrix."""
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
for rep in range(2):
    for g in (0, 8, 16, 32):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("xcd_remap", g); h.set("window_codes", 0)
        print("spmv plain entries  xcd_remap %2d: %.4f ms" % (g, timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))), flush=True)
for rep in range(2):
    for g in (0, 32, 64, 128, 256):
        setk("struct_group", g)
        print("struct group %3d: %.4f ms" % (g, timeit(lambda: kk.spmv_struct("N", 2, dims, 1.0, A, x, 0.0, y))), flush=True)
setk("struct_group", 0)
for nv in (8, 4):
    X = torch.rand(A.numCols(), nv, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numRows(), nv, dtype=torch.float64, device="cuda")
    for g in (0, 16, 32):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("mv_remap", g)
        print("mv%-2d   mv_remap %2d: %.4f ms" % (nv, g, timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y), it=8)), flush=True)
del A, x, y, X, Y
for name, A in (("7-pt 400^3", kk.laplace_matrix("FD", 400, 400, 400)), ("5-pt 1000^2", kk.laplace_matrix("FD", 1000, 1000)), ("9-pt 4000^2", kk.laplace_matrix("FE", 4000, 4000))):
    x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
    for g in (0, 8, 16, 32):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("xcd_remap", g)
        print("%-12s spmv xcd_remap %2d: %.4f ms" % (name, g, timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))), flush=True)
    del A, x, y
# unstructured: uniformly random columns, 16 per row
n = 4000000
rm = torch.arange(0, (n + 1) * 16, 16, dtype=torch.int32, device="cuda")
ent = torch.randint(0, n, (n * 16,), dtype=torch.int32, device="cuda")
val = torch.rand(n * 16, dtype=torch.float64, device="cuda")
A = kk.CrsMatrix(n, n, rm, ent, val)
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(n, dtype=torch.float64, device="cuda")
for g in (0, 16):
    h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("xcd_remap", g)
    print("random 4e6x16 spmv xcd_remap %2d: %.4f ms  codes=%s" % (g, timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y)), h.query("window_codes")), flush=True)
