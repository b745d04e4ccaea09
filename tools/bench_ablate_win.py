#!/usr/bin/env python3
import os; os.environ.setdefault("KKAMD_LIBRARY", "libkkamd_ablate.so")   # the -DKK_ABLATE measurement build (csrc: make ablate)
"""C2 SpMV through the analysed handle (window codes + staged x) with parts of the kernel switched off (knob ablate:
4 no y stores, 64 one tile in eight stores, 128 all stores into one 2 KB window, 8 no LDS row reduction, 16 synthetic row bounds instead of row_map loads).  Results are wrong by design."""
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
for rep in range(2):
    for ab in (0, 4, 64, 128, 8, 16, 12):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("ablate", ab)
        print("ablate %2d: %.4f ms" % (ab, timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))), flush=True)
print("streaming read ceiling: see kkamd_bench_read")
