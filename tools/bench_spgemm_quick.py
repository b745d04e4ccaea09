#!/usr/bin/env python3
"""C = A*A on R-MAT (scale argv[1], edge factor 16): symbolic and numeric wall times of three fresh handles (best of three),
   nothing else -- the loop tools/run_configs.py c4 times, without its CPU baselines.  Usage: python tools/bench_spgemm_quick.py [scale]
   KK_DEFAULTS=a=1,b=2 sets library defaults first; KK_SWEEP="a=1;a=2,b=3" times the same matrix once per ';'-separated set."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, kk_loader, oracle
kk = kk_loader.load()
def set_defaults(spec):
    for kv in spec.split(","):                                          # e.g. spgemm_mid_hash=1
        if kv: kk._capi.check(kk.torch_backend().lib, kk.torch_backend().lib.kkamd_set_default(kv.split("=")[0].encode(), int(kv.split("=")[1])))
set_defaults(os.environ.get("KK_DEFAULTS", ""))
sweep = os.environ.get("KK_SWEEP", "").split(";")
def case_matrix(name):
    """an R-MAT scale ("18"), "feN" = 27-point FE Laplacian N^3, "fdN" = 7-point, "rndN" = N rows of 20 uniformly random columns"""
    if name.startswith("fe") or name.startswith("fd"):
        n = int(name[2:])
        fe = name[1] == "e"
        return "%s %d^3" % ("27-pt FE" if fe else "7-pt FD", n), oracle.laplace3d("FE" if fe else "FD", n, n, n)
    if name.startswith("rnd"):
        n = int(name[3:])
        return "uniform random %d x 20" % n, oracle.random_crs(n, n, 20, variance=0, seed=3, sorted_rows=True)
    return "R-MAT scale %d ef 16" % int(name), oracle.rmat(int(name), 16)
for name in (sys.argv[1] if len(sys.argv) > 1 else "18").split(","):
    label, R = case_matrix(name)
    M = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
    for sw in sweep:
        set_defaults(sw)
        best = None
        for rep in range(int(os.environ.get("KK_REPS", "3"))):
            kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
            if os.environ.get("KK_VERBOSE") and rep == 0: kh.get_spgemm_handle().set("verbose", 1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Cm = kk.spgemm_symbolic(kh, M, False, M, False)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            kk.spgemm_numeric(kh, M, False, M, False, Cm)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            sh = kh.get_spgemm_handle(); mults = sh.get(1); nnzC = Cm.nnz(); src = (sh.get(12), sh.get(14))
            kk.spgemm_numeric(kh, M, False, M, False, Cm)                   # numeric reuse: same handle, same C arrays
            torch.cuda.synchronize(); t3 = time.perf_counter()
            cur = (t2 - t0, t1 - t0, t2 - t1, t3 - t2, sh.get(11))
            best = cur if best is None or cur[0] < best[0] else best
            kh.destroy_spgemm_handle(); del Cm
        b_num = R.nnz * 12 + (R.nrows + 1) * 8 + mults * 12 + nnzC * 12 + (R.nrows + 1) * 8
        b_sym = R.nnz * 4 + (R.nrows + 1) * 8 + mults * 4 + (R.nrows + 1) * 8
        print(json.dumps({"case": label, "mults": mults, "nnzC": nnzC, "symbolic_ms": round(best[1] * 1e3, 3),
                          "numeric_ms": round(best[2] * 1e3, 3), "numeric_reuse_ms": round(best[3] * 1e3, 3), "entries_kept": best[4], "rows_from_bitmaps": src[0], "rows_from_lists": src[1], "numeric_frac_of_gather_model": round(b_num / best[2] / 8e12, 4),
                          "symbolic_frac_of_gather_model": round(b_sym / best[1] / 8e12, 4), "defaults": ",".join(v for v in (os.environ.get("KK_DEFAULTS", ""), sw) if v)}), flush=True)
