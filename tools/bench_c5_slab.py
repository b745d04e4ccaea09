#!/usr/bin/env python3
"""BASELINE config 5, one rank's piece on ONE GPU: rank 3 of 8 of the 27-pt 600^3 Laplacian (27 M rows, 7.27e8 nonzeros, global column
indices, x of 216 M doubles) through the multi-GPU operator with a loop-back transport (tests/dist_loopback.py).  Times the LOCAL SpMV of
the slab (kkamd_dist_spmv_apply what = 2: interior view + boundary views, x as it stands) the way bench.py times a step, and puts it
against the per-GPU algorithmic bytes of SURVEY 8(d): local nnz*12 + (rows+1)*4 + x touched (77 planes)*8 + rows*8.  This is the
per-GPU compute time the N = 8 bench line rests on; the exchange itself (two 2.9-MB planes per neighbour) needs the second GPU."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kk_loader
kk = kk_loader.load()
from kokkos_kernels_amd.dist import DistSpmv
from dist_loopback import Loopback

nx = ny = 600; planes = 75; world = 8; rank = 3
plane = nx * ny; rows = plane * planes; n = rows * world
offsets = [r * rows for r in range(world + 1)]
A = kk.laplace_matrix("FE", nx, ny, planes * world, rows=(rank * rows, rows))
g = torch.Generator(device="cuda"); g.manual_seed(17312837)
x = torch.randint(-20, 20, (n,), device="cuda", generator=g).double()
ranges = [(max(0, offsets[p] - plane), min(n, offsets[p + 1] + plane) - 1) for p in range(world)]
be = kk.torch_backend()
res = {"workload": "spmv_crs_27pt_FE_laplacian_600x600x600_rank3_of_8_slab", "rows": rows, "nnz": A.nnz()}
alg = A.nnz() * 12 + (rows + 1) * 4 + (planes + 2) * plane * 8 + rows * 8
res["algorithmic_bytes_per_call"] = alg
for exchange in ("halo", "allgather_collective"):
    tr = Loopback(x, offsets, rank, ranges)
    op = DistSpmv(A, offsets, rank, transport=tr, exchange=exchange)
    p_full = C.c_void_p(); kk._capi.check(be.lib, be.lib.kkamd_dist_spmv_x_local(op._op, None, C.byref(p_full))); tr.base = p_full.value
    xl = op.x_local(); xl.copy_(x[offsets[rank]:offsets[rank + 1]])
    y = torch.zeros(rows, dtype=torch.float64, device="cuda")
    op.apply(1.0, xl, 0.0, y)                                    # one full step: the halo / the other shards are in place afterwards
    for _ in range(5): op.apply(1.0, xl, 0.0, y, what=2)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); op.apply(1.0, xl, 0.0, y, what=2); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ms = sum(ts) / len(ts)
    res[exchange] = {"local_spmv_ms_mean": round(ms, 4), "min": round(min(ts), 4), "parts": op.query("parts"), "interior_rows": op.interior_rows,
                     "interior_pattern_tiles": op.query("part0_pattern_tiles"), "interior_tiles": op.query("part0_tiles"),
                     "GFLOPs_per_gpu": round(2.0 * A.nnz() / ms / 1e6, 1), "frac_8TBps": round(alg / ms / 1e6 / 8000, 4),
                     "exchange_bytes_received_per_spmv": op.exchange_bytes}
    del op, tr
print(json.dumps(res))
