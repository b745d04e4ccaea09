#!/usr/bin/env python3
import os; os.environ.setdefault("KKAMD_LIBRARY", "libkkamd_ablate.so")   # the -DKK_ABLATE measurement build (csrc: make ablate)
"""C2 SpMV (window codes + staged x) against workgroups per CU: extra dynamic LDS (knob lds_pad_kb) lowers the occupancy."""
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
for npt, pads in ((16, (0, 8, 21, 48)), (8, (0, 16, 24, 37))):
    for pad in pads:
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("nnz_per_thread", npt); h.set("lds_pad_kb", pad)
        lds = npt * 256 * 8 / 1024 + pad
        print("tile %4d  LDS %4.0f KB -> %d WG/CU by LDS (5 by VGPRs): %.4f ms" % (npt * 256, lds, int(160 // lds), timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))), flush=True)
