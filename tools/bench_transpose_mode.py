#!/usr/bin/env python3
"""Timing of the transposed SpMV modes (atomic kernels) next to the explicit alternative: transpose_matrix once, then a
planned 'N' SpMV on the transpose."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, A in (("27pt 100^3", kk.laplace_matrix("FE", 100, 100, 100)), ("C2 27pt 300^3", kk.laplace_matrix("FE", 300, 300, 300))):
    x = torch.rand(A.numRows(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numCols(), dtype=torch.float64, device="cuda")
    h = kk.SPMVHandle("SPMV_DEFAULT")
    t_T = timeit(lambda: kk.spmv(h, "T", 1.0, A, x, 0.0, y)); yT = y.clone()
    hc = kk.SPMVHandle("SPMV_DEFAULT"); hc.set("explicit_transpose", 1)
    torch.cuda.synchronize(); import time; t0 = time.perf_counter()
    kk.spmv(hc, "T", 1.0, A, x, 0.0, y); torch.cuda.synchronize(); t_first = (time.perf_counter() - t0) * 1e3
    t_C = timeit(lambda: kk.spmv(hc, "T", 1.0, A, x, 0.0, y))
    h2c = kk.SPMVHandle("SPMV_DEFAULT"); h2c.set("explicit_transpose", 2)
    t_C2 = timeit(lambda: kk.spmv(h2c, "T", 1.0, A, x, 0.0, y))
    print("%-14s mode T, cached transpose, constant values promised: %.3f ms per call" % (name, t_C2))
    print("%-14s mode T with the cached transpose: first call %.1f ms, then %.3f ms per call (max diff vs atomics %.2g)" % (name, t_first, t_C, (y - yT).abs().max().item()))
    torch.cuda.synchronize(); import time; t0 = time.perf_counter()
    At = kk.transpose_matrix(A); torch.cuda.synchronize(); t_tr = (time.perf_counter() - t0) * 1e3
    h2 = kk.SPMVHandle("SPMV_DEFAULT")
    t_N = timeit(lambda: kk.spmv(h2, "N", 1.0, At, x, 0.0, y))
    print("%-14s mode T (atomics) %.3f ms | transpose_matrix %.1f ms once, then N on A^T %.3f ms | max diff %.2g"
          % (name, t_T, t_tr, t_N, (y - yT).abs().max().item()))
