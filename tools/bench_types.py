import sys; sys.path.insert(0, sys.argv[1])
import numpy as np, torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
n = 300
for vdt, xdt, name in ((np.float32, torch.float32, "f32/f32"), (np.float32, torch.float64, "f32 matrix, f64 vectors"), (np.float64, torch.float64, "f64/f64")):
    A = kk.laplace_matrix("FE", n, n, n, value_dtype=vdt)
    x = torch.rand(A.numCols(), dtype=xdt, device="cuda"); y = torch.zeros(A.numRows(), dtype=xdt, device="cuda")
    h = kk.SPMVHandle("SPMV_DEFAULT")
    t = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y))
    h16 = kk.SPMVHandle("SPMV_DEFAULT"); h16.set("nnz_per_thread", 16)
    t16 = timeit(lambda: kk.spmv(h16, "N", 1.0, A, x, 0.0, y))
    print("   nnz_per_thread=16: %.3f ms" % t16)
    vs = 4 if vdt == np.float32 else 8; xs = 4 if xdt == torch.float32 else 8
    by = A.nnz() * (4 + vs) + (A.numRows() + 1) * 4 + A.numCols() * xs + A.numRows() * xs
    ts = timeit(lambda: kk.spmv_struct("N", 2, (n, n, n), 1.0, A, x, 0.0, y))
    print("%-26s CRS %.3f ms (%.0f GB/s, %.2f of 8 TB/s) | struct %.3f ms" % (name, t, by / t / 1e6, by / t / 1e6 / 8000, ts))
    del A
