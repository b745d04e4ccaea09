import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
n = 300
A = kk.laplace_matrix("FE", n, n, n)
for nv in (16, 8, 4):
    X = torch.rand(A.numCols(), nv, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numRows(), nv, dtype=torch.float64, device="cuda")
    res = []
    for rep in range(3):
        for rm in (1, 0, 2):
            h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("mv_remap", rm)
            res.append((rm, timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y))))
    print("nvec %2d:" % nv, " ".join("remap%d=%.3f" % r for r in res))
