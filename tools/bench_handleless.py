import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
lib = kk.torch_backend().lib
def timeit(fn, it=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, A in (("C1 5pt 1000^2", kk.laplace_matrix("FD", 1000, 1000)), ("9pt 2000^2", kk.laplace_matrix("FE", 2000, 2000)), ("27pt 100^3", kk.laplace_matrix("FE", 100, 100, 100)), ("C2 27pt 300^3", kk.laplace_matrix("FE", 300, 300, 300))):
    x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
    h = kk.SPMVHandle("SPMV_DEFAULT")
    t_plan = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y)); yp = y.clone()
    res = {}
    for knob in (0, 1):
        kk._capi.check(lib, lib.kkamd_set_default(b"transient_min_knnz", knob))
        res[knob] = timeit(lambda: kk.spmv("N", 1.0, A, x, 0.0, y))
        assert (y - yp).abs().max().item() < 1e-9
    kk._capi.check(lib, lib.kkamd_set_default(b"transient_min_knnz", 10000))
    print("%-14s nnz %10d  planned %.4f ms | handle-less: vector kernel %.4f ms, on-the-fly analysis %.4f ms" % (name, A.nnz(), t_plan, res[0], res[1]))
