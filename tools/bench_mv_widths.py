import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
A = kk.laplace_matrix("FE", 300, 300, 300)
nnz, rows = A.nnz(), A.numRows()
for nv in ([int(w) for w in os.environ["KK_WIDTHS"].split(",")] if os.environ.get("KK_WIDTHS") else (4, 8, 12, 24, 32, 40)):
    X = torch.rand(rows, nv, dtype=torch.float64, device="cuda"); Y = torch.zeros(rows, nv, dtype=torch.float64, device="cuda")
    alg = nnz * 12 + (rows + 1) * 4 + 2 * rows * nv * 8
    res = {}
    for name, knobs in ((("default", {}),) if os.environ.get("KK_NO_GATHER") else (("default", {}), ("gather (mv_kernel 2)", {"mv_kernel": 2}))):
        h = kk.SPMVHandle("SPMV_DEFAULT")
        for k, v in knobs.items(): h.set(k, v)
        ms = timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y))
        res[name] = round(ms, 3)
        wg = h.query("mv4_workgroups")
    print(json.dumps({"config": "27-pt 300^3 x %d right-hand sides, row-major" % nv, "ms": res, "frac_8TBps_default": round(alg / res["default"] / 1e6 / 8000, 3)}), flush=True)
    del X, Y
