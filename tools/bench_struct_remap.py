"""spmv_struct interior kernel: dispatch-order vs XCD-contiguous workgroup order (knob struct_remap), C2 and 2-D cases."""
import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, st, dims in (("FE 300^3", "FE", (300, 300, 300)), ("FD 400^3", "FD", (400, 400, 400)), ("FE 5000^2", "FE", (5000, 5000))):
    A = kk.laplace_matrix(st, *dims)
    x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
    code = 2 if st == "FE" else 1
    res = []
    for rep in range(3):
        for rm in (0, 1):
            kk._capi.check(kk.lib(), kk.lib().kkamd_set_default(b"struct_remap", rm))
            res.append((rm, timeit(lambda: kk.spmv_struct("N", code, dims, 1.0, A, x, 0.0, y))))
    kk.lib().kkamd_set_default(b"struct_remap", 0)
    print(name, " ".join("remap%d=%.3f" % r for r in res), "ms")
