#!/usr/bin/env python3
"""Modes T / H of an analysed handle on the bench matrix (27-pt 300^3), rank 1 and rank 2 (16 right-hand sides): the cached transpose
under the three value-tracking policies (0 exact shadow comparison = default, 1 caller notifies, 2 fingerprints) and the reference's
atomic scatter (explicit_transpose 0).  One JSON line."""
import json, os, sys
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
A = kk.laplace_matrix("FE", n, n, n)
rows, nnz = A.numRows(), A.nnz()
res = {"matrix": "27-pt FE %d^3" % n, "nnz": nnz}
x = torch.rand(rows, dtype=torch.float64, device="cuda"); y = torch.zeros(A.numCols(), dtype=torch.float64, device="cuda")
alg1 = nnz * 12 + (rows + 1) * 4 + 2 * rows * 8
ref = None
for tag, knobs in (("exact_default", {}), ("caller_notifies", {"values_tracking": 1}), ("fingerprints", {"values_tracking": 2}), ("atomic_scatter", {"explicit_transpose": 0})):
    h = kk.SPMVHandle("SPMV_DEFAULT")
    for k, v in knobs.items(): h.set(k, v)
    ms = timeit(lambda: kk.spmv(h, "T", 1.0, A, x, 0.0, y))
    if ref is None: ref = y.clone()
    res["rank1_" + tag] = {"ms": round(ms, 4), "frac_8TBps": round(alg1 / ms / 1e6 / 8000, 3), "max_diff_vs_exact": float((y - ref).abs().max())}
    del h
nv = 16
X = torch.rand(rows, nv, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numCols(), nv, dtype=torch.float64, device="cuda")
alg2 = nnz * 12 + (rows + 1) * 4 + 2 * rows * nv * 8
ref = None
for tag, knobs in (("exact_default", {}), ("caller_notifies", {"values_tracking": 1}), ("atomic_scatter", {"explicit_transpose": 0})):
    h = kk.SPMVHandle("SPMV_DEFAULT")
    for k, v in knobs.items(): h.set(k, v)
    ms = timeit(lambda: kk.spmv(h, "T", 1.0, A, X, 0.0, Y), it=10)
    if ref is None: ref = Y.clone()
    res["rank2_x16_" + tag] = {"ms": round(ms, 4), "frac_8TBps": round(alg2 / ms / 1e6 / 8000, 3), "max_diff_vs_exact": float((Y - ref).abs().max())}
    del h
print(json.dumps(res))
