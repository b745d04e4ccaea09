#!/usr/bin/env python3
"""The 8-GPU workload's slabs, one after the other on a single GPU: generator nnz closed form and the A*1 row-sum
identity (interior rows 0, boundary rows 1) for the first, a middle and the last rank of the 600x600x600 case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
nx = ny = 600; planes = 75; world = 8
rows = nx * ny * planes
n = nx * ny * planes * world
ones = torch.ones(n, dtype=torch.float64, device="cuda")
total = 0
for rank in (0, 3, 7):
    A = kk.laplace_matrix("FE", nx, ny, planes * world, rows=(rank * rows, rows))
    y = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    kk.spmv(kk.SPMVHandle("SPMV_DEFAULT"), "N", 1.0, A, ones, 0.0, y)
    lens = A.graph.row_map[1:] - A.graph.row_map[:-1]
    ok = bool((y[lens == 27] == 0).all()) and bool((y[lens < 27] == 1).all())
    cmin, cmax = int(A.graph.entries.min()), int(A.graph.entries.max())
    print("rank %d: nnz %d, columns [%d, %d] (own rows [%d, %d)), row-sum identity %s" % (rank, A.nnz(), cmin, cmax, rank * rows, (rank + 1) * rows, ok))
    assert ok
    del A, y
