#!/usr/bin/env python3
"""Config 3 (27-pt 300^3 x 16 right-hand sides, fp64) measured with bench.py's protocol -- per-call fence, 20 calls, both layouts, plus
   10 queued calls between HIP events -- on the library of the source tree given as argument.  Used for a same-lease A/B of two builds
   (round-3 review item 3: is the driver-timed slowdown of C3 the library or the box?):
       python tools/ab_c3.py <repo root> <label>
   Each invocation is its own process (one library per process); tools/gpu_r5.sh alternates the two trees."""
import json, os, sys, time
root = os.path.abspath(sys.argv[1]); label = sys.argv[2] if len(sys.argv) > 2 else root
sys.path.insert(0, root)
import torch
import kk_loader
kk = kk_loader.load()
n, nv = 300, 16
A = kk.laplace_matrix("FE", n, n, n)
rows, nnz = A.numRows(), A.nnz()
alg = nnz * 12 + (rows + 1) * 4 + 2 * rows * nv * 8
g = torch.Generator(device="cuda"); g.manual_seed(7)
X = torch.randint(-20, 20, (A.numCols(), nv), device="cuda", generator=g).double()
# the device's own streaming rate in this process (a box-speed reference next to the kernel's time)
big = torch.empty(1 << 28, dtype=torch.float64, device="cuda").fill_(1.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): s = big.sum()
e1.record(); torch.cuda.synchronize()
stream_GBps = big.numel() * 8 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9
del big
out = {"label": label, "torch_sum_GBps": round(stream_GBps, 1)}
for layout in ("right", "left"):
    Xl = X if layout == "right" else X.t().contiguous().t()
    Y = torch.full((rows, nv), float("nan"), dtype=torch.float64, device="cuda") if layout == "right" \
        else torch.full((nv, rows), float("nan"), dtype=torch.float64, device="cuda").t()
    h = kk.SPMVHandle("SPMV_DEFAULT")
    fn = lambda: kk.spmv(h, "N", 1.0, A, Xl, 0.0, Y)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        t_ = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t_) * 1e3)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    q = e0.elapsed_time(e1) / 10
    out["layout_" + layout] = {"fenced_mean_ms": round(sum(ts) / len(ts), 4), "fenced_min_ms": round(min(ts), 4), "queued_ms": round(q, 4),
                               "frac_fenced": round(alg / (sum(ts) / len(ts)) / 1e6 / 8000, 4), "mv4_workgroups": h.query("mv4_workgroups")}
    del h, Y
print(json.dumps(out), flush=True)
