#!/usr/bin/env python3
"""The planned SpMV off the benchmark matrix: a table of structured and unstructured matrices, each timed through the default
analysed handle (per-tile column modes) and through the plain nnz-split kernel (window_codes = 0) in the same run, with what
the analysis decided (tile modes), the bytes the plan keeps and the bytes one SpMV streams.  One JSON line per matrix."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kk_loader
kk = kk_loader.load()
dev = "cuda"


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def fixed_rows(n, ncols, cols2d):
    """CrsMatrix with the same number of entries in every row; cols2d: (n, k) int tensor, sorted along dim 1"""
    k = cols2d.shape[1]
    rm = (torch.arange(n + 1, device=dev, dtype=torch.int64) * k).to(torch.int32)
    g = torch.Generator(device=dev); g.manual_seed(5)
    val = torch.rand(n * k, device=dev, dtype=torch.float64, generator=g) + 0.5
    return kk.CrsMatrix(n, ncols, rm, cols2d.reshape(-1).to(torch.int32).contiguous(), val)


def multi_dof(A, ndof):
    """every entry of A becomes a dense ndof x ndof block (a vector-valued finite-element matrix on A's mesh), built on the device"""
    rm = A.graph.row_map.to(torch.int64); ent = A.graph.entries.to(torch.int64)
    n = A.numRows()
    lens = (rm[1:] - rm[:-1]) * ndof
    new_rm = torch.zeros(n * ndof + 1, device=dev, dtype=torch.int64)
    new_rm[1:] = torch.cumsum(torch.repeat_interleave(lens, ndof), 0)
    E = (ent.unsqueeze(1) * ndof + torch.arange(ndof, device=dev).unsqueeze(0)).reshape(-1)
    nnz = int(new_rm[-1].item())
    p = torch.arange(nnz, device=dev)
    R = torch.searchsorted(new_rm, p, right=True) - 1
    src = rm[R // ndof] * ndof + (p - new_rm[R])
    del p, R
    new_ent = E[src].to(torch.int32)
    del E, src
    g = torch.Generator(device=dev); g.manual_seed(7)
    val = torch.rand(nnz, device=dev, dtype=torch.float64, generator=g) + 0.5
    return kk.CrsMatrix(n * ndof, A.numCols() * ndof, new_rm.to(torch.int32), new_ent, val)


def matrices(only=""):
    for name, make in _matrix_makers():
        if only and not any(o in name for o in only.split('|')): continue
        yield name, make()


def _matrix_makers():
    g = torch.Generator(device=dev); g.manual_seed(11)
    yield "27-pt FE 300^3 (C2)", lambda: kk.laplace_matrix("FE", 300, 300, 300)
    yield "7-pt FD 300^3", lambda: kk.laplace_matrix("FD", 300, 300, 300)
    yield "9-pt FE 4000^2", lambda: kk.laplace_matrix("FE", 4000, 4000)
    def banded():
        n = 10_000_000
        c = (torch.arange(n, device=dev).unsqueeze(1) + torch.randint(-20000, 20001, (n, 12), device=dev, generator=g)) % n
        return fixed_rows(n, n, torch.sort(c, dim=1).values)
    yield "banded random, 1e7 rows x 12 in +-20000", banded
    def blockdiag():
        n = 2_000_000
        c = (torch.arange(n, device=dev) // 32 * 32).unsqueeze(1) + torch.arange(32, device=dev).unsqueeze(0)
        return fixed_rows(n, n, c)
    yield "block diagonal, 32 x 32 blocks, 2e6 rows", blockdiag
    def uniform():
        n = 5_000_000
        c = torch.randint(0, n, (n, 20), device=dev, generator=g)
        return fixed_rows(n, n, torch.sort(c, dim=1).values)
    yield "uniform random, 5e6 rows x 20", uniform
    def scattered():
        # a stencil with a few rows that couple to columns all over the matrix: those tiles read entries, the rest keep their modes
        A = kk.laplace_matrix("FE", 200, 200, 200)
        rm = A.graph.row_map
        rows = torch.arange(100_003, 7_900_000, 123_457, device=dev)              # 64 interior rows
        for r in rows.tolist():
            lo, hi = int(rm[r].item()), int(rm[r + 1].item())
            A.graph.entries[lo:hi] = torch.sort(torch.randint(0, 8_000_000, (hi - lo,), device=dev, generator=g)).values.to(torch.int32)
        return A
    yield "27-pt FE 200^3 with 64 rows of scattered columns", scattered
    def rmat():
        import oracle
        R = oracle.rmat(22, 16)
        return kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values)
    yield "R-MAT scale 22, edge factor 16", rmat
    yield "3 dof per node on 27-pt FE 100^3", lambda: multi_dof(kk.laplace_matrix("FE", 100, 100, 100), 3)


def main():
  for name, A in matrices():
      nnz, nr, nc = A.nnz(), A.numRows(), A.numCols()
      g = torch.Generator(device=dev); g.manual_seed(3)
      x = torch.randint(-20, 20, (nc,), device=dev, generator=g).double()
      y1 = torch.full((nr,), float("nan"), dtype=torch.float64, device=dev); y0 = torch.empty_like(y1)
      t0 = time.perf_counter()
      h = kk.SPMVHandle("SPMV_DEFAULT"); kk.spmv(h, "N", 1.0, A, x, 0.0, y1); torch.cuda.synchronize()
      t_analysis = time.perf_counter() - t0
      hp = kk.SPMVHandle("SPMV_DEFAULT"); hp.set("window_codes", 0); hp.set("colslab", 0); kk.spmv(hp, "N", 1.0, A, x, 0.0, y0)
      diff = float((y1 - y0).abs().max().item())
      ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y1)); ms_plain = timeit(lambda: kk.spmv(hp, "N", 1.0, A, x, 0.0, y0))
      cs = {"colslab": h.query("colslab"), "colslab_selection_us": [h.query("colslab_crs_us"), h.query("colslab_us")]}
      if h.query("colslab_tried") and h.query("colslab_crs_us"):       # the column-slab copy was considered: what it does with constant values promised
          hc = kk.SPMVHandle("SPMV_DEFAULT"); hc.set("colslab", 2); hc.set("colslab_const", 1); kk.spmv(hc, "N", 1.0, A, x, 0.0, y0)
          cs["ms_colslab_const_values"] = round(timeit(lambda: kk.spmv(hc, "N", 1.0, A, x, 0.0, y0)), 4)
          cs["colslab_bytes_per_nnz"] = round(hc.query("colslab_bytes") / nnz, 2); cs["colslab_slabs"] = hc.query("colslab_slabs")
          del hc
      alg = nnz * 12 + (nr + 1) * 4 + nc * 8 + nr * 8
      tiles, tile = h.query("tiles"), h.query("tile")
      pat, code, plain = h.query("pattern_tiles"), h.query("code_tiles"), h.query("plain_tiles")
      streamed = alg
      if h.query("window_codes"):
          streamed = alg - nnz * 4 + tiles * 4 + (tiles - plain) * 256 + code * tile * 2 + pat * 672 + plain * tile * 4
      print(json.dumps({"matrix": name, "rows": nr, "nnz": nnz, "ms_default_plan": round(ms, 4), "ms_plain_kernel": round(ms_plain, 4),
                        "speedup_vs_plain": round(ms_plain / ms, 3), "frac_8TBps_crs_bytes": round(alg / ms / 1e6 / 8000, 3),
                        "frac_8TBps_streamed_bytes": round(streamed / ms / 1e6 / 8000, 3), "GFLOPs": round(2 * nnz / ms / 1e6, 1),
                        "tiles": tiles, "tile_nnz": tile, "pattern_tiles": pat, "code_tiles": code, "staged_tiles": h.query("staged_tiles"),
                        "plain_tiles": plain, "plan_bytes_per_nnz": round(h.query("plan_bytes") / nnz, 4), "first_call_incl_analysis_ms": round(t_analysis * 1e3, 1),
                        "max_abs_diff_vs_plain": diff, **cs}), flush=True)
      del A, h, hp, x, y0, y1
      torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
