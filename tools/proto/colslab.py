#!/usr/bin/env python3
"""Measurement prototype (see colslab.hip): would a column-slab copy of a gather-bound matrix beat the CRS stream kernel?
   python tools/proto/colslab.py [uniform|rmat]   -- prints one JSON line per slab width."""
import ctypes, json, os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(os.path.dirname(here)); sys.path.insert(0, root)
import numpy as np, torch, kk_loader, oracle
so = os.path.join(here, "_colslab.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "colslab.hip")):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "colslab.hip")])
if len(sys.argv) > 1 and sys.argv[1] == "build": sys.exit(0)
lib = ctypes.CDLL(so); vp = ctypes.c_void_p
lib.proto_colslab.argtypes = [ctypes.c_int64, vp, vp, vp, vp, vp, ctypes.c_int, vp]
kk = kk_loader.load(); dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "uniform"
if which == "uniform":
    nrows, per = 5_000_000, 20
    g = torch.Generator(device=dev); g.manual_seed(5)
    col = torch.randint(0, nrows, (nrows * per,), device=dev, dtype=torch.int32, generator=g)
    row = torch.arange(nrows, device=dev, dtype=torch.int32).repeat_interleave(per)
    rm = torch.arange(0, nrows * per + 1, per, device=dev, dtype=torch.int64)
    key = row.to(torch.int64) * nrows + col; key, _ = torch.sort(key); row = (key // nrows).to(torch.int32); col = (key % nrows).to(torch.int32); del key
else:
    R = oracle.rmat(22, 16); nrows = R.nrows
    rm = torch.from_numpy(R.row_map.astype(np.int64)).to(dev); col = torch.from_numpy(R.entries).to(dev)
    row = torch.repeat_interleave(torch.arange(nrows, device=dev, dtype=torch.int32), (rm[1:] - rm[:-1]))
nnz = col.numel()
val = torch.rand(nnz, device=dev, dtype=torch.float64) + 0.5
x = torch.rand(nrows, device=dev, dtype=torch.float64); y = torch.zeros(nrows, device=dev, dtype=torch.float64)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
A = kk.CrsMatrix(nrows, nrows, rm, col, val)
h = kk.SPMVHandle(); yref = torch.zeros_like(y)
t_crs = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, yref))
st = torch.cuda.current_stream().cuda_stream
for shift in (15, 16, 17, 18, 19, 20, 23):
    key = (col.to(torch.int64) >> shift) * nrows + row.to(torch.int64)
    _, perm = torch.sort(key, stable=True); del key
    r2, c2, v2 = row[perm].contiguous(), col[perm].contiguous(), val[perm].contiguous(); del perm
    run = lambda variant: lib.proto_colslab(nnz, r2.data_ptr(), c2.data_ptr(), v2.data_ptr(), x.data_ptr(), y.data_ptr(), variant, st)
    y.zero_(); run(0); torch.cuda.synchronize()
    err = float(((y - yref).abs().max() / yref.abs().max()).item())
    t0 = timeit(lambda: run(0)); t1 = timeit(lambda: run(1))
    print(json.dumps({"matrix": which, "nnz": nnz, "slab_cols": 1 << shift, "slabs": (nrows >> shift) + 1, "crs_stream_ms": round(t_crs, 4), "colslab_ms": round(t0, 4),
                      "colslab_loads_only_ms": round(t1, 4), "rel_err": err}), flush=True)
    del r2, c2, v2
