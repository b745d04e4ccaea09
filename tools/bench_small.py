import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, A in (("C1 5pt 1000^2", kk.laplace_matrix("FD", 1000, 1000)), ("9pt 1500^2", kk.laplace_matrix("FE", 1500, 1500)), ("27pt 100^3", kk.laplace_matrix("FE", 100, 100, 100)), ("27pt 160^3", kk.laplace_matrix("FE", 160, 160, 160))):
    x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
    out = []
    for npt in (4, 8, 16):
        h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("nnz_per_thread", npt)
        out.append("npt%d %.4f" % (npt, timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))))
    h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("kernel", 1)
    out.append("vector %.4f" % timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y)))
    print("%-14s nnz %9d: %s ms" % (name, A.nnz(), ", ".join(out)))
