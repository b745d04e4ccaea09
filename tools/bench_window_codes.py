#!/usr/bin/env python3
"""Window codes (16-bit column codes built by the SpMV analysis): kernel time and analysis time with and without, per matrix."""
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
def case(name, A, dtype=torch.float64):
    x = torch.rand(A.numCols(), dtype=dtype, device="cuda"); y = torch.zeros(A.numRows(), dtype=dtype, device="cuda")
    for knobs in ({"window_codes": 0}, {"window_codes": 2}, {}, {"nnz_per_thread": 8}, {"nnz_per_thread": 16}):
        setup = []
        for rep in range(3):
            h = kk.SPMVHandle("SPMV_DEFAULT")
            for k, v in knobs.items(): h.set(k, v)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            kk.spmv(h, "N", 1.0, A, x, 0.0, y)
            torch.cuda.synchronize(); setup.append((time.perf_counter() - t0) * 1e3)
        t = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))
        print("%-28s %-24s codes=%d staged_x=%d tile=%d  spmv %.4f ms   first call (analysis + spmv) %.2f ms" % (name, knobs, h.query("window_codes"), h.query("window_staged_x"), h.query("tile"), t, min(setup)), flush=True)
case("27-pt 300^3 fp64", kk.laplace_matrix("FE", 300, 300, 300))
case("7-pt 400^3 fp64", kk.laplace_matrix("FD", 400, 400, 400))
case("5-pt 1000^2 fp64 (C1)", kk.laplace_matrix("FD", 1000, 1000))
case("27-pt 150^3 fp64", kk.laplace_matrix("FE", 150, 150, 150))
case("9-pt 4000^2 fp64", kk.laplace_matrix("FE", 4000, 4000))
