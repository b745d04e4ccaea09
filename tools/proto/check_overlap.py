#!/usr/bin/env python3
"""Would the per-call value fingerprint pass of the column-slab copy hide beside the slab kernel?  Proxy: the slab kernel (constant
   values promised) on one stream, a streaming read of A.values (torch sum) on another; both together against one after the other."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, kk_loader
kk = kk_loader.load(); dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(11)
n, k = 5_000_000, 20
c = torch.sort(torch.randint(0, n, (n, k), device=dev, generator=g), dim=1).values
rm = (torch.arange(n + 1, device=dev, dtype=torch.int64) * k).to(torch.int32)
val = torch.rand(n * k, device=dev, dtype=torch.float64, generator=g) + 0.5
A = kk.CrsMatrix(n, n, rm, c.reshape(-1).to(torch.int32).contiguous(), val); del c
x = torch.rand(n, device=dev, dtype=torch.float64); y = torch.zeros(n, device=dev, dtype=torch.float64)
h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("colslab", 2); h.set("colslab_const", 1)
kk.spmv(h, "N", 1.0, A, x, 0.0, y); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
out = torch.zeros((), device=dev, dtype=torch.float64)
def both(overlap, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(s1): kk.spmv(h, "N", 1.0, A, x, 0.0, y)
        with torch.cuda.stream(s2 if overlap else s1): out.copy_(val.sum())
        if overlap: s1.wait_stream(s2)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / reps
def only(which, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(s1):
            if which == 0: kk.spmv(h, "N", 1.0, A, x, 0.0, y)
            else: out.copy_(val.sum())
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / reps
for _ in range(2): both(True); both(False)
print(json.dumps({"slab_kernel_ms": round(only(0), 4), "stream_read_ms": round(only(1), 4), "one_after_the_other_ms": round(both(False), 4), "side_by_side_ms": round(both(True), 4)}))
