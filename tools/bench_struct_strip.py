#!/usr/bin/env python3
"""spmv_struct interior kernel: dispatch order against the strip order (struct_strip = grid lines per XCD strip)."""
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
for name, st, dims in (("FE 300^3", "FE", (300, 300, 300)), ("FD 400^3", "FD", (400, 400, 400)), ("FE 5000^2", "FE", (5000, 5000)), ("FE 150^3", "FE", (150, 150, 150))):
    A = kk.laplace_matrix(st, *dims)
    x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
    code = 2 if st == "FE" else 1
    ref = None
    for rep in range(2):
        out = []
        for strip in (0, 1, 2, 4, 8, 16):
            setk("struct_strip", strip)
            t = timeit(lambda: kk.spmv_struct("N", code, dims, 1.0, A, x, 0.0, y))
            if ref is None: ref = y.clone()
            out.append("strip%d=%.4f" % (strip, t))
            assert (y - ref).abs().max().item() == 0.0
        print(name, " ".join(out), "ms", flush=True)
    setk("struct_strip", 0)
    del A, x, y
