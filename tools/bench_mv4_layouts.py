#!/usr/bin/env python3
"""C3 (27-pt 300^3 x 16, fp64) through the plane-marching kernel with the layouts of X and Y chosen separately: which side of LayoutLeft costs the time.
   Usage: python tools/bench_mv4_layouts.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nv = 16
A = kk.laplace_matrix("FE", n, n, n)
nnz, rows = A.nnz(), A.numRows()
alg = nnz * 12 + (rows + 1) * 4 + 2 * rows * nv * 8
Xr = torch.rand(rows, nv, dtype=torch.float64, device="cuda")
Xl = Xr.t().contiguous().t()
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
ref = None
for xl, yl in ((0, 0), (1, 0), (0, 1), (1, 1)):
    X = Xl if xl else Xr
    Y = torch.zeros(nv, rows, dtype=torch.float64, device="cuda").t() if yl else torch.zeros(rows, nv, dtype=torch.float64, device="cuda")
    h = kk.SPMVHandle("SPMV_DEFAULT")
    kk.spmv(h, "N", 1.0, A, X, 0.0, Y)
    if ref is None: ref = Y.clone()
    err = float((Y - ref).abs().max())
    ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y))
    print(json.dumps({"X": "left" if xl else "right", "Y": "left" if yl else "right", "ms": round(ms, 4), "frac_8TBps": round(alg / ms / 1e6 / 8000, 3), "max_abs_diff": err,
                      "mv4_workgroups": h.query("mv4_workgroups")}), flush=True)
