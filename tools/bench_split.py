#!/usr/bin/env python3
"""Single-GPU timing of what one rank of the 8-GPU run executes: the 600x600x75 slab (global columns) as one planned
SpMV versus interior + two boundary planes (the compute side of dist.py's overlap path; no communication here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
nx = ny = 600; planes = 75; world = 8; rank = 3
rows = nx * ny * planes
A = kk.laplace_matrix("FE", nx, ny, planes * world, rows=(rank * rows, rows))
n = nx * ny * planes * world
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.zeros(rows, dtype=torch.float64, device="cuda")
def timeit(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
h = kk.SPMVHandle("SPMV_DEFAULT")
t_full = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))
y_ref = y.clone()
rm = A.graph.row_map
parts = []
for a, b in ((nx * ny, rows - nx * ny), (0, nx * ny), (rows - nx * ny, rows)):
    p0, p1 = int(rm[a].item()), int(rm[b].item())
    sub = kk.CrsMatrix(b - a, n, (rm[a:b + 1] - rm[a]).contiguous(), A.graph.entries[p0:p1], A.values[p0:p1])
    parts.append((kk.SPMVHandle("SPMV_DEFAULT"), sub, a, b))
y2 = torch.zeros_like(y)
def split():
    for hh, sub, a, b in parts: kk.spmv(hh, "N", 1.0, sub, x, 0.0, y2[a:b])
t_split = timeit(split)
print("slab %d rows, nnz %d: one SpMV %.4f ms, interior+2 planes %.4f ms, max diff %.3g" % (rows, A.nnz(), t_full, t_split, (y2 - y_ref).abs().max().item()))
