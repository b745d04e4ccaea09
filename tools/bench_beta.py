#!/usr/bin/env python3
"""C2 SpMV through the analysed handle for beta = 0 / 1 (the reference's perf drivers time beta = 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
A = kk.laplace_matrix("FE", 300, 300, 300)
x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
h = kk.SPMVHandle("SPMV_DEFAULT")
for rep in range(3):
    for beta in (0.0, 1.0, -0.5):
        print("beta %4.1f: %.4f ms" % (beta, timeit(lambda: kk.spmv(h, "N", 1.0, A, x, beta, y))), flush=True)
