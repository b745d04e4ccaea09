import sys, json; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, kk_loader
kk = kk_loader.load(); dev = "cuda"
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it
g = torch.Generator(device=dev); g.manual_seed(11)
for n, k in ((20_000_000, 8), (10_000_000, 12), (5_000_000, 20), (5_000_000, 40)):
    c = torch.sort(torch.randint(0, n, (n, k), device=dev, generator=g), dim=1).values
    rm = (torch.arange(n + 1, device=dev, dtype=torch.int64) * k)
    val = torch.rand(n * k, device=dev, dtype=torch.float64, generator=g) + 0.5
    A = kk.CrsMatrix(n, n, rm, c.reshape(-1).to(torch.int32).contiguous(), val); del c
    x = torch.rand(n, device=dev, dtype=torch.float64); y = torch.zeros(n, device=dev, dtype=torch.float64)
    h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("colslab", 0); kk.spmv(h, "N", 1.0, A, x, 0.0, y)
    out = {"n": n, "k": k, "crs": round(timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y)), 3)}
    for shift in (17, 18, 19, 20, 21):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("colslab", 2); h.set("colslab_shift", shift); kk.spmv(h, "N", 1.0, A, x, 0.0, y)
        out["s%d" % shift] = round(timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y)), 3); del h
    print(json.dumps(out), flush=True)
    del A, x, y, val, rm; torch.cuda.empty_cache()
