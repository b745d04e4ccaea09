#!/usr/bin/env python3
"""Measures every BASELINE.json configuration that fits one GPU, with the CPU baseline beside it (same run, same
box) and a parity check against the oracle.  Output: one JSON line per result (kept in profiles/<round>/).
C5 (8 GPUs) is measured by the driver through bench.py --gpus N."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "threads")
import numpy as np
import torch
import kk_loader
import oracle

kk = kk_loader.load()
oracle.set_omp_threads(oracle.usable_cpus())


def emit(**kw):
    print(json.dumps(kw), flush=True)


def gpu_time(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ts) / len(ts), ts[0]


def cpu_time(fn, min_s=3.0, min_it=3):
    fn()
    t0 = time.perf_counter(); it = 0
    while True:
        fn(); it += 1
        el = time.perf_counter() - t0
        if el >= min_s and it >= min_it:
            return el / it, it


def spmv_bytes(nnz, nrows, ncols, beta=0.0):
    return nnz * 12 + (nrows + 1) * 4 + ncols * 8 + nrows * 8 + (nrows * 8 if beta else 0)


def c1():
    """5-pt 2-D FD Laplacian 1000x1000, fp64: Kokkos::Serial restatement on one host core + the GPU on the same matrix"""
    A0 = oracle.laplace2d("FD", 1000, 1000)
    rng = np.random.default_rng(17312837)
    x = rng.integers(-20, 20, size=A0.ncols).astype(np.float64); y = np.zeros(A0.nrows)
    t, it = cpu_time(lambda: oracle.spmv_serial("N", A0, 1.0, x, 0.0, y))
    by = spmv_bytes(A0.nnz, A0.nrows, A0.ncols)
    emit(config="C1", device="CPU Kokkos::Serial restatement, 1 core", rows=A0.nrows, nnz=A0.nnz, ms=t * 1e3, GFLOPs=2 * A0.nnz / t / 1e9,
         GBps=by / t / 1e9, iterations=it)
    A = kk.laplace_matrix("FD", 1000, 1000)
    rm, ent, val = A.to_host()
    same = bool(np.array_equal(rm, A0.row_map) and np.array_equal(ent, A0.entries) and np.array_equal(val, A0.values))
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(A0.nrows, dtype=torch.float64, device="cuda")
    h = kk.SPMVHandle("SPMV_DEFAULT")
    mean, mn = gpu_time(lambda: kk.spmv(h, "N", 1.0, A, xd, 0.0, yd))
    err = float(np.abs(yd.cpu().numpy() - y).max())
    emit(config="C1", device="1x MI355X", rows=A0.nrows, nnz=A0.nnz, ms=mean, ms_min=mn, GFLOPs=2 * A0.nnz / mean / 1e6, GBps=by / mean / 1e6,
         generator_bit_identical=same, max_abs_diff_vs_oracle=err, tol=oracle.spmv_max_error(A0, 1.0, 0.0, max_x=20, max_val=4))


def c2_c3():
    n = 300
    A = kk.laplace_matrix("FE", n, n, n)
    nr, nnz = A.numRows(), A.nnz()
    g = torch.Generator(device="cuda"); g.manual_seed(17312837)
    x = torch.randint(-20, 20, (nr,), device="cuda", generator=g).double(); y = torch.zeros(nr, dtype=torch.float64, device="cuda")
    h = kk.SPMVHandle("SPMV_DEFAULT")
    for beta in (0.0, 1.0):
        mean, mn = gpu_time(lambda: kk.spmv(h, "N", 1.0, A, x, beta, y), iters=100)
        by = spmv_bytes(nnz, nr, nr, beta)
        emit(config="C2", device="1x MI355X", beta=beta, rows=nr, nnz=nnz, ms=mean, ms_min=mn, GFLOPs=2 * nnz / mean / 1e6, GBps=by / mean / 1e6,
             frac_of_8TBps=by / mean / 1e6 / 8000, driver_formula_GBps=(nnz * 12 + nr * 4 + (nnz + nr) * 8) / mean / 1e6)
    mean, mn = gpu_time(lambda: kk.spmv("N", 1.0, A, x, 0.0, y), iters=30)
    emit(config="C2", device="1x MI355X", variant="handle-less (SPMV_FAST_SETUP, vector kernel)", ms=mean, GFLOPs=2 * nnz / mean / 1e6,
         GBps=spmv_bytes(nnz, nr, nr) / mean / 1e6)
    # modes T / H: cached transpose (default: compare-and-move refresh; 2: constant values promised) against the atomic scatter (0)
    for et, label in ((1, "cached transpose, values compared every call (default)"), (2, "cached transpose, constant values promised"), (0, "atomic scatter (the reference's algorithm)")):
        ht = kk.SPMVHandle("SPMV_DEFAULT"); ht.set("explicit_transpose", et)
        t0 = time.perf_counter(); kk.spmv(ht, "T", 1.0, A, x, 0.0, y); torch.cuda.synchronize(); first = time.perf_counter() - t0
        mean, mn = gpu_time(lambda: kk.spmv(ht, "T", 1.0, A, x, 0.0, y), iters=10, warm=2)
        emit(config="C2-transpose", device="1x MI355X", variant=label, ms=mean, ms_min=mn, first_call_s=first, GFLOPs=2 * nnz / mean / 1e6,
             frac_of_8TBps=spmv_bytes(nnz, nr, nr) / mean / 1e6 / 8000, transpose_cached=ht.query("transpose_cached"))
        del ht
        torch.cuda.empty_cache()
    # N2: the structured path on the same matrix (what the reference's Laplacian perf driver times first)
    ys = torch.zeros_like(y)
    mean, mn = gpu_time(lambda: kk.spmv_struct("N", 2, (n, n, n), 1.0, A, x, 0.0, ys), iters=100)
    kk.spmv(h, "N", 1.0, A, x, 0.0, y)
    by_s = nnz * 8 + (nr + 1) * 4 + nr * 8 + nr * 8              # values + row_map + x + y written (beta = 0: y is not read)
    emit(config="C2-struct", device="1x MI355X", rows=nr, nnz=nnz, ms=mean, ms_min=mn, GFLOPs=2 * nnz / mean / 1e6, GBps=by_s / mean / 1e6,
         frac_of_8TBps=by_s / mean / 1e6 / 8000, crs_equivalent_GBps=spmv_bytes(nnz, nr, nr) / mean / 1e6,
         max_abs_diff_vs_crs_kernel=float((ys - y).abs().max().item()))
    del ys
    nv = 16
    by = nnz * 12 + (nr + 1) * 4 + 2 * nr * nv * 8
    for layout in ("LayoutRight", "LayoutLeft"):
        if layout == "LayoutRight":
            X = torch.randint(-20, 20, (nr, nv), device="cuda", generator=g).double(); Y = torch.zeros(nr, nv, dtype=torch.float64, device="cuda")
        else:
            X = torch.randint(-20, 20, (nv, nr), device="cuda", generator=g).double().t(); Y = torch.zeros(nv, nr, dtype=torch.float64, device="cuda").t()
        y1 = torch.empty(nr, dtype=torch.float64, device="cuda"); kk.spmv(h, "N", 1.0, A, X[:, 5].contiguous(), 0.0, y1)
        tol = 10 * float(np.finfo(np.float64).eps) * 27 * 32.0 * 20.0      # the reference's bound: 10 eps max_nnz_row max_val max_x
        for knobs, label in (({}, "default"), ({"mv_kernel": 2}, "gather kernel (mv_kernel 2)")):
            hm = kk.SPMVHandle("SPMV_DEFAULT")
            for k_, v_ in knobs.items(): hm.set(k_, v_)
            t0 = time.perf_counter(); kk.spmv(hm, "N", 1.0, A, X, 0.0, Y); torch.cuda.synchronize(); first = time.perf_counter() - t0
            mean, mn = gpu_time(lambda: kk.spmv(hm, "N", 1.0, A, X, 0.0, Y), iters=20, warm=2)
            emit(config="C3", device="1x MI355X", layout=layout, nvec=nv, knobs=label, ms=mean, ms_min=mn, GFLOPs=2 * nnz * nv / mean / 1e6,
                 GBps=by / mean / 1e6, frac_of_8TBps=by / mean / 1e6 / 8000, first_call_s=first,
                 rank2_kernel=("plane marching, %d workgroups, %d rows left to the gather kernel" % (hm.query("mv4_workgroups"), hm.query("mv4_other_rows"))
                               if hm.query("mv4_workgroups") else "wave-private gather"),
                 rank2_plan_bytes=hm.query("mv_plan_bytes"), col5_max_abs_diff_vs_rank1=float((Y[:, 5] - y1).abs().max().item()), tol=tol)
        del X, Y
    del A, x, y
    torch.cuda.empty_cache()
    # CPU baselines on bounded samples (OpenMP port of the host functors)
    m = 160
    A0 = oracle.laplace3d("FE", m, m, m)
    ft = oracle.first_touch
    rm32 = ft(A0.row_map.astype(np.int32)); ent = ft(A0.entries); val = ft(A0.values)
    rng = np.random.default_rng(1)
    xs = ft(rng.integers(-20, 20, size=A0.ncols).astype(np.float64)); ys = ft(np.zeros(A0.nrows))
    t, it = cpu_time(lambda: oracle.spmv_omp(rm32, ent, val, 1.0, xs, 0.0, ys), min_s=5)
    emit(config="C2", device="CPU OpenMP port, %d threads" % oracle.omp_threads(), sample="27-pt %d^3" % m, rows=A0.nrows, nnz=A0.nnz, ms=t * 1e3,
         GFLOPs=2 * A0.nnz / t / 1e9, GBps=spmv_bytes(A0.nnz, A0.nrows, A0.ncols) / t / 1e9, iterations=it)
    Xs = ft(rng.integers(-20, 20, size=(A0.ncols, nv)).astype(np.float64)); Ys = ft(np.zeros((A0.nrows, nv)))
    t, it = cpu_time(lambda: oracle.spmv_mv_omp(rm32, ent, val, 1.0, Xs, 0.0, Ys), min_s=5)
    emit(config="C3", device="CPU OpenMP port, %d threads" % oracle.omp_threads(), sample="27-pt %d^3 x %d RHS (row-major)" % (m, nv), ms=t * 1e3,
         GFLOPs=2 * A0.nnz * nv / t / 1e9, GBps=(A0.nnz * 12 + (A0.nrows + 1) * 4 + 2 * A0.nrows * nv * 8) / t / 1e9, iterations=it)
    # parity of the GPU rank-2 path at this size against the Serial oracle
    Ad = kk.CrsMatrix.from_host(A0.nrows, A0.ncols, A0.row_map, A0.entries, A0.values)
    Xd = torch.from_numpy(Xs).cuda(); Yd = torch.zeros(A0.nrows, nv, dtype=torch.float64, device="cuda")
    kk.spmv(kk.SPMVHandle("SPMV_DEFAULT"), "N", 1.0, Ad, Xd, 0.0, Yd)
    err = float(np.abs(Yd.cpu().numpy() - Ys).max())
    emit(config="C3", check="GPU vs OpenMP port at 160^3 x 16", max_abs_diff=err, tol=oracle.spmv_max_error(A0, 1.0, 0.0, max_x=20, max_val=32))


def c4(scales=(14, 16, 18, 20)):
    for scale in scales:
        t0 = time.perf_counter(); R = oracle.rmat(scale, 16); tgen = time.perf_counter() - t0
        M = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
        best = None
        for rep in range(3):
            kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Cm = kk.spgemm_symbolic(kh, M, False, M, False)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            kk.spgemm_numeric(kh, M, False, M, False, Cm)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            sh = kh.get_spgemm_handle(); mults = sh.get(1); nnzC = Cm.nnz(); mx = (sh.get(2), sh.get(3))
            cur = (t2 - t0, t1 - t0, t2 - t1)
            best = cur if best is None or cur[0] < best[0] else best
            if scale > 16 or rep < 2:
                kh.destroy_spgemm_handle(); del Cm
        so = 8
        b_num = R.nnz * 12 + (R.nrows + 1) * so + mults * 12 + nnzC * 12 + (R.nrows + 1) * so
        b_sym = R.nnz * 4 + (R.nrows + 1) * so + mults * 4 + (R.nrows + 1) * so
        emit(config="C4", device="1x MI355X", case="R-MAT scale %d ef 16, C = A*A, int32 ordinals, int64 offsets" % scale, rows=R.nrows, nnzA=R.nnz,
             nnzC=nnzC, mults=mults, max_row_flops=mx[0], max_row_nnz=mx[1], symbolic_ms=best[1] * 1e3, numeric_ms=best[2] * 1e3,
             total_ms=best[0] * 1e3, GFLOPs_numeric=2 * mults / best[2] / 1e9, gather_model_GBps_numeric=b_num / best[2] / 1e9,
             gather_model_GBps_symbolic=b_sym / best[1] / 1e9, host_generation_s=tgen)
        if scale <= 16:
            # CPU baseline = the library default on Serial AND OpenMP (SPGEMM_SERIAL) + parity of the GPU result
            t0 = time.perf_counter(); Cg = oracle.spgemm(R, R); tc = time.perf_counter() - t0
            rm, ent, val = Cm.to_host()
            ok, msg = oracle.is_same_matrix(oracle.Crs(R.nrows, R.ncols, rm.astype(np.int64), ent, val), Cg)
            emit(config="C4", device="CPU SPGEMM_SERIAL restatement (+ sort), 1 core", case="R-MAT scale %d" % scale, mults=mults, total_ms=tc * 1e3,
                 GFLOPs=2 * mults / tc / 1e9, gpu_result_identical_structure_and_values_1e7=ok, msg=msg)
            kh.destroy_spgemm_handle(); del Cm
        if scale <= 18:
            # SPGEMM_KK on the host cores: OpenMP port of the KKMEM hash kernels (what SPGEMM_KK runs on Kokkos::OpenMP for this k)
            oracle.set_omp_threads(oracle.usable_cpus())
            tk = {}
            oracle.spgemm_kkmem_omp(R, R, timings=tk)
            emit(config="C4", device="CPU SPGEMM_KK (KKMEM hash accumulators, OpenMP port), %d threads" % oracle.omp_threads(), case="R-MAT scale %d" % scale,
                 mults=mults, symbolic_ms=tk["symbolic_s"] * 1e3, numeric_ms=tk["numeric_s"] * 1e3, sort_ms=tk["sort_s"] * 1e3,
                 GFLOPs_numeric=2 * mults / tk["numeric_s"] / 1e9)
        del M
        torch.cuda.empty_cache()


def c4_laplace(n=100, compression=1, algo="SPGEMM_KK"):
    """structured SpGEMM (the reference perf test's usual input): 27-pt Laplacian squared"""
    M = kk.laplace_matrix("FE", n, n, n)
    best = None
    for rep in range(3):
        kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle(algo); kh.get_spgemm_handle().set("compression", compression)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Cm = kk.spgemm_symbolic(kh, M, False, M, False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        kk.spgemm_numeric(kh, M, False, M, False, Cm)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        mults = kh.get_spgemm_handle().get(1); nnzC = Cm.nnz(); comp = (kh.get_spgemm_handle().get(6), kh.get_spgemm_handle().get(7))
        cur = (t2 - t0, t1 - t0, t2 - t1)
        best = cur if best is None or cur[0] < best[0] else best
        kh.destroy_spgemm_handle(); del Cm
    nr, nnz = M.numRows(), M.nnz()
    b_num = nnz * 12 + (nr + 1) * 4 + mults * 12 + nnzC * 12 + (nr + 1) * 4
    emit(config="C4-structured", device="1x MI355X", case="27-pt %d^3 Laplacian, C = A*A" % n, algorithm=algo, compression_option=compression,
         b_compressed=bool(comp[0]), symbolic_insertions=comp[1], rows=nr, nnzA=nnz, nnzC=nnzC, mults=mults,
         symbolic_ms=best[1] * 1e3, numeric_ms=best[2] * 1e3, GFLOPs_numeric=2 * mults / best[2] / 1e9,
         gather_model_GBps_numeric=b_num / best[2] / 1e9)


def c4_symbolic_only(scale=22):
    """C4 as specified (R-MAT scale 22): the numeric phase does not fit one GPU (nnz(C) ~ 7e10), the symbolic phase does."""
    t0 = time.perf_counter(); R = oracle.rmat(scale, 16); tgen = time.perf_counter() - t0
    M = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
    best = None
    for rep in range(2):
        kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Cm = kk.spgemm_symbolic(kh, M, False, M, False, allocate=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sh = kh.get_spgemm_handle(); mults = sh.get(1); nnzC = sh.get_c_nnz(); mx = (sh.get(2), sh.get(3))
        best = (t1 - t0) if best is None or (t1 - t0) < best else best
        kh.destroy_spgemm_handle(); del Cm
    b_sym = R.nnz * 4 + (R.nrows + 1) * 8 + mults * 4 + (R.nrows + 1) * 8
    emit(config="C4", device="1x MI355X", case="R-MAT scale %d ef 16, C = A*A, SYMBOLIC PHASE ONLY" % scale, rows=R.nrows, nnzA=R.nnz, nnzC=nnzC,
         mults=mults, max_row_flops=mx[0], max_row_nnz=mx[1], symbolic_ms=best * 1e3, gather_model_GBps_symbolic=b_sym / best / 1e9,
         bytes_of_C_if_numeric_ran=nnzC * 12, host_generation_s=tgen)


def c4_slab(scale=22, world=8, ranks=(0, 7)):
    """C4 as specified, the way it fits: rows of A in `world` slabs of equal multiplication count, B replicated, no
    communication.  One GPU is available here, so the slabs of the chosen ranks run one after the other: each line is
    what ONE of the eight GPUs does (symbolic + numeric on its slab, C slab resident in its HBM)."""
    from kokkos_kernels_amd.dist import work_balanced_offsets
    R = oracle.rmat(scale, 16)
    lenB = np.diff(R.row_map)
    cs = np.concatenate([[0], np.cumsum(lenB[R.entries], dtype=np.int64)])
    flops = cs[R.row_map[1:]] - cs[R.row_map[:-1]]
    offs = work_balanced_offsets(flops, world)
    B = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
    for r in ranks:
        a, b = offs[r], offs[r + 1]
        sl = slice(R.row_map[a], R.row_map[b])
        A = kk.CrsMatrix.from_host(b - a, R.ncols, R.row_map[a:b + 1] - R.row_map[a], R.entries[sl], R.values[sl], offset_dtype=np.int64)
        kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Cm = kk.spgemm_symbolic(kh, A, False, B, False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        kk.spgemm_numeric(kh, A, False, B, False, Cm)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        sh = kh.get_spgemm_handle(); mults = sh.get(1); nnzC = Cm.nnz()
        # size-independent check of the slab: every row sorted and duplicate-free, row sums of C == A * (B * 1)
        ent = Cm.graph.entries; rm = Cm.graph.row_map
        sorted_ok = bool(((ent[1:] > ent[:-1]) | torch.isin(torch.arange(1, nnzC, device="cuda"), rm[1:-1])).all().item()) if nnzC < 2**31 else None
        ones = torch.ones(R.ncols, dtype=torch.float64, device="cuda")
        b1 = torch.zeros(R.nrows, dtype=torch.float64, device="cuda"); kk.spmv("N", 1.0, B, ones, 0.0, b1)
        ab1 = torch.zeros(b - a, dtype=torch.float64, device="cuda"); kk.spmv("N", 1.0, A, b1, 0.0, ab1)
        c1 = torch.zeros(b - a, dtype=torch.float64, device="cuda"); kk.spmv("N", 1.0, Cm, ones, 0.0, c1)
        rel = float(((c1 - ab1).abs() / ab1.abs().clamp_min(1.0)).max().item())
        emit(config="C4", device="1x MI355X = rank %d of %d" % (r, world), case="R-MAT scale %d ef 16, row slab [%d, %d) of A times all of B" % (scale, a, b),
             rows=b - a, mults=mults, nnzC_slab=nnzC, C_slab_GB=nnzC * 12 / 1e9, symbolic_ms=(t1 - t0) * 1e3, numeric_ms=(t2 - t1) * 1e3,
             GFLOPs_numeric=2 * mults / (t2 - t1) / 1e9, rows_sorted_unique=sorted_ok, rowsum_identity_max_rel_err=rel)
        kh.destroy_spgemm_handle(); del Cm, A
        torch.cuda.empty_cache()


if __name__ == "__main__":
    what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c1", "c2", "c4"]
    torch.cuda.set_device(0)
    if "c1" in what: c1()
    if "c2" in what: c2_c3()
    if "c4" in what: c4(tuple(int(v) for v in os.environ.get("KK_C4_SCALES", "14,16,18,20").split(",")))
    if "c4lap" in what:
        c4_laplace(compression=0); c4_laplace(compression=1); c4_laplace(algo="SPGEMM_KK_DENSE")
    if "c4s22" in what: c4_symbolic_only(22)
    if "c4slab" in what: c4_slab(int(os.environ.get("KK_C4_SLAB_SCALE", "22")))
