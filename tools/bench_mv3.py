#!/usr/bin/env python3
"""C3 (27-pt 300^3 x 16 right-hand sides, fp64): the rank-2 kernels against each other in one run.
   mv_kernel 2 = wave-private gather kernel (round 1), 3 = LDS-staged X tiles with tile order mv_order 0 / 1 / 2.
   Usage: python tools/bench_mv3.py [n] [quick]"""
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
sq = quick and sys.argv[2] == "sq"          # counter passes: the default kernel and the LDS-staged kernel once each
A = kk.laplace_matrix("FE", n, n, n)
nnz, rows = A.nnz(), A.numRows()
for nv in ((16,) if quick else (16, 8, 32)):
    X = torch.rand(A.numCols(), nv, dtype=torch.float64, device="cuda")
    Y = torch.zeros(rows, nv, dtype=torch.float64, device="cuda")
    alg = nnz * 12 + (rows + 1) * 4 + 2 * rows * nv * 8
    cases = [("mv2 wave-private gather, grouped order 16", {"mv_kernel": 2, "mv_order": 0}),
             ("mv2 wave-private gather, strip order", {"mv_kernel": 2, "mv_order": 2})]
    if not quick:
        cases += [("mv2 strip order, L2 budget %d KB" % kb, {"mv_kernel": 2, "mv_order": 2, "mv_strip_l2_kb": kb}) for kb in (1000, 1600, 3500)]
        cases += [("mv3 LDS-staged, order %d" % o, {"mv_kernel": 3, "mv_order": o}) for o in (1, 2)]
    if sq:
        cases = [cases[1], ("mv3 LDS-staged, order 2", {"mv_kernel": 3, "mv_order": 2})]
    for rep in range(2):
        for name, knobs in cases:
            h = kk.SPMVHandle("SPMV_DEFAULT")
            for k, v in knobs.items(): h.set(k, v)
            ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y))
            print(json.dumps({"case": name, "n": n, "nvec": nv, "layout": "right", "ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1),
                              "frac_8TBps": round(alg / ms / 1e6 / 8000, 3), "GFLOPs": round(2.0 * nnz * nv / ms / 1e6, 1),
                              "mv_tiles": h.query("mv_tiles"), "mv_staged": h.query("mv_staged_tiles"), "order_used": h.query("mv_order"),
                              "mv_plan_bytes": h.query("mv_plan_bytes")}), flush=True)
    if nv == 16 and not sq:
        Xl = torch.rand(nv, A.numCols(), dtype=torch.float64, device="cuda").t(); Yl = torch.zeros(nv, rows, dtype=torch.float64, device="cuda").t()
        for name, knobs in (cases[0], cases[1]):
            h = kk.SPMVHandle("SPMV_DEFAULT")
            for k, v in knobs.items(): h.set(k, v)
            ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, Xl, 0.0, Yl), it=5)
            print(json.dumps({"case": name, "n": n, "nvec": nv, "layout": "left (packed per call)", "ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1)}), flush=True)
