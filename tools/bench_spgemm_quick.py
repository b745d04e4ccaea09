#!/usr/bin/env python3
"""C = A*A on R-MAT (scale argv[1], edge factor 16): symbolic and numeric wall times of three fresh handles (best of three),
   nothing else -- the loop tools/run_configs.py c4 times, without its CPU baselines.  Usage: python tools/bench_spgemm_quick.py [case,case,...]
   (cases: see case_products)
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
def aggregation(n):
    """P of a 2 x 2 x 2 aggregation of an n^3 grid (n even): fine node (i, j, k) -> coarse node (i/2, j/2, k/2), one entry of 1 per row"""
    i = np.arange(n); c = n // 2
    ci = (i[:, None, None] // 2) * c * c + (i[None, :, None] // 2) * c + (i[None, None, :] // 2)      # node = x * n^2 + y * n + z
    nf = n ** 3
    return oracle.Crs(nf, c ** 3, np.arange(nf + 1, dtype=np.int64), ci.reshape(-1).astype(np.int32), np.ones(nf))
def to_scipy(M):
    import scipy.sparse as sp
    return sp.csr_matrix((M.values, M.entries, M.row_map), shape=(M.nrows, M.ncols))
def from_scipy(S):
    S = S.tocsr(); S.sort_indices()
    return oracle.Crs(S.shape[0], S.shape[1], S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data.astype(np.float64))
def case_products(name):
    """[(label, A, B)]: an R-MAT scale ("18"), "feN" = 27-point FE Laplacian N^3, "fdN" = 7-point, "rndN" = N rows of 20 uniformly random columns (each
    times itself), "rapN" = the two products of a Galerkin coarse operator R A P on the 27-point Laplacian N^3 with a 2 x 2 x 2 aggregation
    (Test_Sparse_spgemm.hpp:491-504 has the shape list; perf_test/sparse/KokkosSparse_spgemm.cpp:372-423 the driver): A P (27 products
    per row, 8 .. 27 entries) and R (A P) (rows of 8 x 27 = 216 products, 27 entries) -- a mixed-size product"""
    if name.startswith("rap"):
        n = int(name[3:])
        A = oracle.laplace3d("FE", n, n, n); P = aggregation(n)
        AP = from_scipy(to_scipy(A) @ to_scipy(P)); R = from_scipy(to_scipy(P).T)
        return [("A P, 27-pt FE %d^3, 2x2x2 aggregation" % n, A, P), ("R (A P), 27-pt FE %d^3, 2x2x2 aggregation" % n, R, AP)]
    if name.startswith("fe") or name.startswith("fd"):
        n = int(name[2:])
        fe = name[1] == "e"
        M = oracle.laplace3d("FE" if fe else "FD", n, n, n)
        return [("%s %d^3" % ("27-pt FE" if fe else "7-pt FD", n), M, M)]
    if name.startswith("rnd"):
        n = int(name[3:])
        M = oracle.random_crs(n, n, 20, variance=0, seed=3, sorted_rows=True)
        return [("uniform random %d x 20" % n, M, M)]
    M = oracle.rmat(int(name), 16)
    return [("R-MAT scale %d ef 16" % int(name), M, M)]
for name in (sys.argv[1] if len(sys.argv) > 1 else "18").split(","):
  for label, RA, RB in case_products(name):
    MA = kk.CrsMatrix.from_host(RA.nrows, RA.ncols, RA.row_map, RA.entries, RA.values, offset_dtype=np.int64)
    MB = MA if RB is RA else kk.CrsMatrix.from_host(RB.nrows, RB.ncols, RB.row_map, RB.entries, RB.values, offset_dtype=np.int64)
    for sw in sweep:
        set_defaults(sw)
        best = None
        for rep in range(int(os.environ.get("KK_REPS", "3"))):
            kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
            if os.environ.get("KK_VERBOSE") and rep == int(os.environ.get("KK_VERBOSE_REP", "0")): kh.get_spgemm_handle().set("verbose", int(os.environ["KK_VERBOSE"]))
            for kv in os.environ.get("KK_HANDLE_OPTS", "").split(","):            # handle options, e.g. KK_HANDLE_OPTS=compression=2
                if kv: kh.get_spgemm_handle().set(kv.split("=")[0], float(kv.split("=")[1]) if "." in kv.split("=")[1] else int(kv.split("=")[1]))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Cm = kk.spgemm_symbolic(kh, MA, False, MB, False)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if os.environ.get("KK_SYM_ONLY"):                                 # measurement builds whose symbolic phase leaves no usable structure
                best = (t1 - t0, t1 - t0, 0.0, 0.0, 0) if best is None or t1 - t0 < best[0] else best
                mults = kh.get_spgemm_handle().get(1); nnzC = Cm.nnz(); src = (0, 0, 0)
                kh.destroy_spgemm_handle(); del Cm
                continue
            kk.spgemm_numeric(kh, MA, False, MB, False, Cm)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            sh = kh.get_spgemm_handle(); mults = sh.get(1); nnzC = Cm.nnz(); src = (sh.get(12), sh.get(14), sh.get(16))
            kk.spgemm_numeric(kh, MA, False, MB, False, Cm)                   # numeric reuse: same handle, same C arrays
            torch.cuda.synchronize(); t3 = time.perf_counter()
            cur = (t2 - t0, t1 - t0, t2 - t1, t3 - t2, sh.get(11))
            best = cur if best is None or cur[0] < best[0] else best
            kh.destroy_spgemm_handle(); del Cm
        # gather model: A, B row maps and entries once, 12 B per product, C written once
        b_num = RA.nnz * 12 + (RA.nrows + 1) * 8 + (RB.nrows + 1) * 8 + mults * 12 + nnzC * 12 + (RA.nrows + 1) * 8
        b_sym = RA.nnz * 4 + (RA.nrows + 1) * 8 + (RB.nrows + 1) * 8 + mults * 4 + (RA.nrows + 1) * 8
        print(json.dumps({"case": label, "mults": mults, "nnzC": nnzC, "symbolic_ms": round(best[1] * 1e3, 3),
                          "numeric_ms": round(best[2] * 1e3, 3), "numeric_reuse_ms": round(best[3] * 1e3, 3), "entries_kept": best[4], "rows_from_bitmaps": src[0], "rows_from_lists": src[1],
                          "rows_column_blocks": src[2], "numeric_frac_of_gather_model": round(b_num / max(best[2], 1e-9) / 8e12, 4), "reuse_frac_of_gather_model": round(b_num / max(best[3], 1e-9) / 8e12, 4),
                          "symbolic_frac_of_gather_model": round(b_sym / best[1] / 8e12, 4), "defaults": ",".join(v for v in (os.environ.get("KK_DEFAULTS", ""), sw, os.environ.get("KK_HANDLE_OPTS", "")) if v)}), flush=True)
