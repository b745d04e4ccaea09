#!/usr/bin/env python3
"""Rank-1 SpMV on gather-bound matrices: the CRS stream kernel, the column-slab copy with its per-call value fingerprints (what the
automatic selection weighs) and with constant values promised.  One JSON line per matrix.  `quick`: three calls of each, for a
counter pass (rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, kk_loader, oracle
kk = kk_loader.load(); dev = "cuda"
quick = "quick" in sys.argv


def timeit(fn, it=3 if quick else 30):
    for _ in range(1 if quick else 3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def matrices():
    g = torch.Generator(device=dev); g.manual_seed(11)
    n, k = 5_000_000, 20
    c = torch.sort(torch.randint(0, n, (n, k), device=dev, generator=g), dim=1).values
    rm = (torch.arange(n + 1, device=dev, dtype=torch.int64) * k).to(torch.int32)
    val = torch.rand(n * k, device=dev, dtype=torch.float64, generator=g) + 0.5
    if "rmat" not in sys.argv: yield "uniform random, 5e6 rows x 20", kk.CrsMatrix(n, n, rm, c.reshape(-1).to(torch.int32).contiguous(), val)
    del c, rm, val
    if quick and "rmat" not in sys.argv: return
    R = oracle.rmat(22, 16)
    yield "R-MAT scale 22, edge factor 16", kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values)
    if quick: return
    n, k = 20_000_000, 8
    c = torch.sort(torch.randint(0, n, (n, k), device=dev, generator=g), dim=1).values
    rm = (torch.arange(n + 1, device=dev, dtype=torch.int64) * k)
    val = torch.rand(n * k, device=dev, dtype=torch.float64, generator=g) + 0.5
    yield "uniform random, 2e7 rows x 8", kk.CrsMatrix(n, n, rm, c.reshape(-1).to(torch.int32).contiguous(), val)


for name, A in matrices():
    nnz, nr, nc = A.nnz(), A.numRows(), A.numCols()
    x = torch.rand(nc, device=dev, dtype=torch.float64); y = torch.zeros(nr, device=dev, dtype=torch.float64); y0 = torch.zeros_like(y)
    out = {"matrix": name, "rows": nr, "nnz": nnz}
    alg = nnz * 12 + (nr + 1) * 4 + nc * 8 + nr * 8
    for tag, knobs in (("crs_stream", {"colslab": 0}), ("default", {}), ("det_forced", {"colslab": 4}), ("det_forced_notified_values", {"colslab": 4, "values_tracking": 1}),
                       ("det_forced_8MB_slabs", {"colslab": 4, "colslab_shift": 20}),
                       ("colslab_forced", {"colslab": 2}), ("colslab_const_values", {"colslab": 2, "colslab_const": 1})):
        h = kk.SPMVHandle("SPMV_DEFAULT")
        for k_, v_ in knobs.items(): h.set(k_, v_)
        kk.spmv(h, "N", 1.0, A, x, 0.0, y)
        if tag == "crs_stream": y0.copy_(y)
        ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))
        out[tag + "_ms"] = round(ms, 4); out[tag + "_frac_8TBps_crs_bytes"] = round(alg / ms / 1e6 / 8000, 3)
        if tag == "default":
            out["default_kept_copy"] = h.query("colslab"); out["default_deterministic_form"] = h.query("colslab_deterministic"); out["lines_of_x_per_nonzero"] = h.query("colslab_lines_permille") / 1000
            y1 = y.clone(); kk.spmv(h, "N", 1.0, A, x, 0.0, y); out["default_bit_stable"] = bool((y1 == y).all().item())
        if tag == "colslab_forced": out["slabs"] = h.query("colslab_slabs"); out["copy_bytes_per_nnz"] = round(h.query("colslab_bytes") / nnz, 2)
        out.setdefault("max_rel_diff_vs_crs", 0.0)
        out["max_rel_diff_vs_crs"] = max(out["max_rel_diff_vs_crs"], float(((y - y0).abs().max() / y0.abs().max()).item()))
        del h
    print(json.dumps(out), flush=True)
    del A, x, y, y0
    torch.cuda.empty_cache()
