#!/usr/bin/env python3
"""A/B of the planned SpMV kernel variants on C2 in one run (box-to-box spread is larger than most differences)."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
A = kk.laplace_matrix("FE", n, n, n)
x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
ref = None
for rep in range(2):
    for name, knobs in (("default (quad-dealt gather, 4096)", {}), ("late gather, 4096-nnz tiles", {"stream_variant": 5, "nnz_per_thread": 16}),
                        ("late gather, 2048-nnz tiles", {"stream_variant": 5, "nnz_per_thread": 8}), ("late gather, 1024-nnz tiles", {"stream_variant": 5, "nnz_per_thread": 4}),
                        ("window codes, 4096-nnz tiles", {"stream_variant": 6, "nnz_per_thread": 16}), ("window codes, 2048-nnz tiles", {"stream_variant": 6, "nnz_per_thread": 8}),
                        ("default, 2048-nnz tiles", {"nnz_per_thread": 8})):
        h = kk.SPMVHandle("SPMV_DEFAULT")
        for k, v in knobs.items(): h.set(k, v)
        t = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))
        if ref is None: ref = y.clone()
        print("%-36s %.4f ms  maxdiff %.2g  window_codes=%s" % (name, t, (y - ref).abs().max().item(), h.query("window_codes")))
