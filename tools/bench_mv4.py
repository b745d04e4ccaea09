#!/usr/bin/env python3
"""C3 (27-pt 300^3 x 16 right-hand sides, fp64): the plane-marching rank-2 kernel (mv_kernel 4) beside the wave-private gather
   kernel (mv_kernel 2, strip order), results compared element by element, and the k-chunk knob swept.
   Usage: python tools/bench_mv4.py [n] [quick]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()


def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
quick = len(sys.argv) > 2
nv = 16
for stencil in (("FE",) if quick else ("FE", "FD")):
    A = kk.laplace_matrix(stencil, n, n, n)
    nnz, rows = A.nnz(), A.numRows()
    X = torch.rand(A.numCols(), nv, dtype=torch.float64, device="cuda")
    alg = nnz * 12 + (rows + 1) * 4 + 2 * rows * nv * 8
    ref = None
    cases = [("mv2 wave-private gather, strip order", {"mv_kernel": 2})]
    cases += [("mv4 plane marching, %d workgroups per CU" % w, {"mv_kernel": 4, "mv4_wg_per_cu": w}) for w in ((8,) if quick else (8, 3, 5, 12, 20))]
    for name, knobs in cases:
        h = kk.SPMVHandle("SPMV_DEFAULT")
        for k, v in knobs.items(): h.set(k, v)
        Y = torch.full((rows, nv), float("nan"), dtype=torch.float64, device="cuda")
        kk.spmv(h, "N", 1.0, A, X, 0.0, Y)
        if ref is None:
            ref = Y.clone(); err = 0.0
        else:
            err = float((Y - ref).abs().max()); assert err == err, "NaN left in Y"
        Y2 = torch.rand(rows, nv, dtype=torch.float64, device="cuda"); Y3 = Y2.clone()
        kk.spmv(h, "N", 2.0, A, X, -1.0, Y3)
        err_b = float((Y3 - (2.0 * ref - Y2)).abs().max())
        ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y))
        print(json.dumps({"case": name, "stencil": stencil, "n": n, "nvec": nv, "ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1),
                          "frac_8TBps": round(alg / ms / 1e6 / 8000, 3), "GFLOPs": round(2.0 * nnz * nv / ms / 1e6, 1),
                          "max_abs_diff_vs_mv2": err, "max_abs_diff_beta": err_b, "workgroups": h.query("mv4_workgroups"),
                          "other_rows": h.query("mv4_other_rows"), "stencil_entries": h.query("mv4_stencil"),
                          "mv_plan_bytes": h.query("mv_plan_bytes")}), flush=True)
    # LayoutLeft X and Y (the plane-marching kernel reads X where it lies; the gather kernel packs it per call)
    Xl = X.t().contiguous().t(); Yl = torch.zeros(nv, rows, dtype=torch.float64, device="cuda").t()
    for name, knobs in (cases[0], cases[1]):
        h = kk.SPMVHandle("SPMV_DEFAULT")
        for k, v in knobs.items(): h.set(k, v)
        kk.spmv(h, "N", 1.0, A, Xl, 0.0, Yl)
        err = float((Yl - ref).abs().max())
        ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, Xl, 0.0, Yl), it=5)
        print(json.dumps({"case": name, "stencil": stencil, "n": n, "nvec": nv, "layout": "left", "ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1),
                          "frac_8TBps": round(alg / ms / 1e6 / 8000, 3), "max_abs_diff_vs_mv2_right": err}), flush=True)
