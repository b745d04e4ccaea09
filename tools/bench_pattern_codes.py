#!/usr/bin/env python3
"""Row-pattern records against per-nonzero 16-bit codes in the staged SpMV kernel (knob pattern_codes 0 / 1 = default from 1e7 nnz)."""
import os, sys, time
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
def case(name, A):
    x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
    ref = None
    for rep in range(2):
        for pat in (0, 1):
            h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("pattern_codes", pat); h.set("pattern_codes_min_knnz", 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            kk.spmv(h, "N", 1.0, A, x, 0.0, y); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
            t = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))
            if ref is None: ref = y.clone()
            print("%-18s pattern_codes=%d  tile %d  pattern tiles %d / %d  spmv %.4f ms  first call %.1f ms  maxdiff %.2g" % (
                name, pat, h.query("tile"), h.query("pattern_tiles"), h.query("tiles"), t, first, (y - ref).abs().max().item()), flush=True)
case("27-pt 300^3", kk.laplace_matrix("FE", 300, 300, 300))
case("7-pt 400^3", kk.laplace_matrix("FD", 400, 400, 400))
case("9-pt 4000^2", kk.laplace_matrix("FE", 4000, 4000))
case("5-pt 1000^2 (C1)", kk.laplace_matrix("FD", 1000, 1000))
