import os; os.environ.setdefault("KKAMD_LIBRARY", "libkkamd_ablate.so")   # the -DKK_ABLATE measurement build (csrc: make ablate)
"""spmv_struct interior kernel on C2 against workgroups per CU (extra dynamic LDS lowers the occupancy)."""
import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
def setk(k, v): kk._capi.check(kk.lib(), kk.lib().kkamd_set_default(k.encode(), v))
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
dims = (300, 300, 300)
A = kk.laplace_matrix("FE", *dims)
x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
for rep in range(2):
    for pad in (0, 3, 7, 13, 21, 34):
        setk("struct_lds_pad_kb", pad)
        print("pad %2d KB (%d WG/CU): %.3f ms" % (pad, int(160 // (19.1 + pad)), timeit(lambda: kk.spmv_struct("N", 2, dims, 1.0, A, x, 0.0, y))))
setk("struct_lds_pad_kb", 0)
