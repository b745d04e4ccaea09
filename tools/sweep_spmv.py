#!/usr/bin/env python3
"""GPU sweep of the SpMV / SpMV_MV / SpGEMM kernel variants (expert knobs) -- the evidence behind the
defaults chosen in kk_spmv.hip.  Prints one JSON line per variant; run through gpurun, output kept under
profiles/.  Not part of the product path."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import kk_loader

kk = kk_loader.load()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--what", default="copy,spmv,mv,spgemm")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    what = set(args.what.split(","))
    n = args.n
    torch.cuda.set_device(0)
    be = kk.torch_backend()

    if "copy" in what:
        a = torch.empty(1 << 29, dtype=torch.float64, device="cuda").normal_()     # 4 GiB
        b = torch.empty_like(a)
        med, mn = timeit(lambda: b.copy_(a), args.iters)
        emit(kind="device_copy", bytes=2 * a.numel() * 8, ms_med=med, ms_min=mn, GBps=2 * a.numel() * 8 / med / 1e6)
        med, mn = timeit(lambda: torch.add(a, b, alpha=2.0, out=b), args.iters)
        emit(kind="device_triad_like", bytes=3 * a.numel() * 8, ms_med=med, ms_min=mn, GBps=3 * a.numel() * 8 / med / 1e6)
        out = torch.zeros(8, dtype=torch.float64, device="cuda")
        for loads in (2, 4, 8):
            for nt in (0, 1):
                for persist in (0, 1):
                    fn = lambda: kk._capi.check(be.lib, be.lib.kkamd_bench_read(a.data_ptr(), a.numel() * 8, loads, nt, persist, out.data_ptr(), be.stream()))
                    med, mn = timeit(fn, args.iters)
                    emit(kind="device_read", loads=loads, nontemporal=nt, persistent=persist, bytes=a.numel() * 8, ms_med=med, ms_min=mn,
                         GBps=a.numel() * 8 / med / 1e6, GBps_best=a.numel() * 8 / mn / 1e6)
        del a, b

    A = kk.laplace_matrix("FE", n, n, n)
    nr, nnz = A.numRows(), A.nnz()
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randint(-20, 20, (nr,), device="cuda", generator=g).double()
    y = torch.zeros(nr, dtype=torch.float64, device="cuda")
    bytes0 = nnz * 12 + (nr + 1) * 4 + nr * 8 + nr * 8

    if "spmv" in what:
        ref = torch.empty_like(y)
        kk.spmv("N", 1.0, A, x, 0.0, ref)
        for lpr in (8, 16):
            for remap in (0,):
                h = kk.SPMVHandle("SPMV_FAST_SETUP"); h.set("lanes_per_row", lpr); h.set("xcd_remap", remap)
                med, mn = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y), args.iters)
                emit(kind="spmv_vector", n=n, lpr=lpr, xcd_remap=remap, ms_med=med, ms_min=mn, GBps=bytes0 / med / 1e6,
                     GFLOPs=2 * nnz / med / 1e6, maxdiff=float((y - ref).abs().max()))
        variants = [(1, 0, 0), (5, 4, 0), (5, 5, 0), (5, 6, 0), (5, 8, 0)]
        for npt in (8, 16):
            for nt in (0,):
                for var, wg, remap in variants:
                    h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("nnz_per_thread", npt); h.set("nontemporal", nt); h.set("xcd_remap", remap)
                    h.set("stream_variant", var); h.set("wg_per_cu", wg)
                    for beta in (0.0, 1.0):
                        if beta == 1.0 and not (nt == 0 and npt == 8):
                            continue
                        med, mn = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, beta, y), args.iters)
                        by = bytes0 + (nr * 8 if beta else 0)
                        kk.spmv(h, "N", 1.0, A, x, 0.0, y)
                        emit(kind="spmv_stream", n=n, nnz_per_thread=npt, nontemporal=nt, variant=var, wg_per_cu=wg, xcd_remap=remap, beta=beta,
                             ms_med=round(med, 4), ms_min=round(mn, 4), GBps=round(by / med / 1e6, 1), GFLOPs=round(2 * nnz / med / 1e6, 1),
                             frac_of_8TBps=round(by / med / 1e6 / 8000, 4), maxdiff=float((y - ref).abs().max()))

    if "ablate" in what:
        for npt in (8, 16):
            for abl in (0, 1024, 2048, 0):
                h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("nnz_per_thread", npt); h.set("stream_variant", 1); h.set("ablate", abl)
                med, mn = timeit(lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y), args.iters)
                emit(kind="spmv_ablate", npt=npt, ablate=abl, ms_med=round(med, 4), ms_min=round(mn, 4), GBps=round(bytes0 / med / 1e6, 1))

    if "mv" in what:
        nv = 16
        for layout in ("right", "left"):
            if layout == "right":
                X = torch.randint(-20, 20, (nr, nv), device="cuda", generator=g).double(); Y = torch.zeros(nr, nv, dtype=torch.float64, device="cuda")
            else:
                X = torch.randint(-20, 20, (nv, nr), device="cuda", generator=g).double().t(); Y = torch.zeros(nv, nr, dtype=torch.float64, device="cuda").t()
            y1 = torch.empty_like(y); kk.spmv("N", 1.0, A, X[:, 3].contiguous(), 0.0, y1)
            by = nnz * 12 + (nr + 1) * 4 + 2 * nr * nv * 8
            for mvk, rm in ((0, 0), (3, 0), (3, 1)):
                h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("mv_kernel", mvk); h.set("mv_remap", rm)
                med, mn = timeit(lambda: kk.spmv(h, "N", 1.0, A, X, 0.0, Y), max(5, args.iters // 4), warm=1)
                emit(kind="spmv_mv", n=n, nvec=nv, layout=layout, mv_kernel=mvk, mv_remap=rm, ms_med=round(med, 3), ms_min=round(mn, 3), GBps=round(by / med / 1e6, 1),
                     GFLOPs=round(2 * nnz * nv / med / 1e6, 1), frac_of_8TBps=round(by / med / 1e6 / 8000, 4), maxdiff_col3=float((Y[:, 3] - y1).abs().max()))
            del X, Y

    if "spgemm" in what:
        import oracle
        del A, x, y
        torch.cuda.empty_cache()
        cases = [("laplace27_100^3", lambda: kk.laplace_matrix("FE", 100, 100, 100))]
        for scale in (16, 18, 20):
            def mk(scale=scale):
                R = oracle.rmat(scale, 16)
                return kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
            cases.append(("rmat_s%d_ef16" % scale, mk))
        for name, mk in cases:
            t0 = time.perf_counter(); M = mk(); tgen = time.perf_counter() - t0
            for rep in range(2):
                kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                Cm = kk.spgemm_symbolic(kh, M, False, M, False)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                kk.spgemm_numeric(kh, M, False, M, False, Cm)
                torch.cuda.synchronize(); t2 = time.perf_counter()
                sh = kh.get_spgemm_handle()
                mults = sh.get(1)
                emit(kind="spgemm", case=name, rep=rep, rows=M.numRows(), nnzA=M.nnz(), nnzC=Cm.nnz(), mults=mults,
                     max_row_flops=sh.get(2), max_row_nnz=sh.get(3), gen_s=tgen, symbolic_ms=(t1 - t0) * 1e3,
                     numeric_ms=(t2 - t1) * 1e3, GFLOPs_numeric=2 * mults / (t2 - t1) / 1e9)
                kh.destroy_spgemm_handle(); del Cm
            del M; torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
