import os; os.environ.setdefault("KKAMD_LIBRARY", "libkkamd_ablate.so")   # the -DKK_ABLATE measurement build (csrc: make ablate)
"""spmv_struct interior kernel on C2 with parts switched off (struct_remap bits: 2 no y store, 4 no old-y load, 8 no x loads, 16 no stencil loop)."""
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
    for ab in (0, 2, 4, 6, 8, 16, 14, 30):
        setk("struct_remap", ab)
        print("ablate %2d: %.3f ms" % (ab, timeit(lambda: kk.spmv_struct("N", 2, dims, 1.0, A, x, 0.0, y))))
setk("struct_remap", 0)
print("read ceiling: %.0f GB/s" % kk.bench_read(1 << 31) if hasattr(kk, "bench_read") else "")
