#!/usr/bin/env python3
import os; os.environ.setdefault("KKAMD_LIBRARY", "libkkamd_ablate.so")   # the -DKK_ABLATE measurement build (csrc: make ablate)
"""C3 through the LDS-staged rank-2 kernel with parts switched off (knob ablate: 1 no X staging, 2 no contraction loop,
4 no A staging, 8 no Y store).  Results are wrong by design; this only shows where the time goes."""
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()


def timeit(fn, it=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
A = kk.laplace_matrix("FE", n, n, n)
X = torch.rand(A.numCols(), 16, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numRows(), 16, dtype=torch.float64, device="cuda")
for glds in (1, 0):
    for ab in (0, 1, 2, 4, 8, 3, 5, 7, 15):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("mv_kernel", 3); h.set("mv_order", 2); h.set("ablate", ab); h.set("mv_glds", glds)
        print("glds %d ablate %2d: %.3f ms" % (glds, ab, timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y))), flush=True)
