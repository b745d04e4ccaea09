#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json): CSR SpMV GFLOP/s + achieved HBM GB/s on the
27-point 3-D FE Laplacian, fp64, 1/2/4/8 MI355X.

  N = 1   workload C2: 300^3 grid (27,000,000 rows, 724,150,792 stored nnz), one planned SpMV per step.
  N > 1   workload C5 family: 600 x 600 x (75*N) grid, 1-D row slabs of 75 z-planes (27,000,000 rows,
          ~7.27e8 nnz) per GPU, so N = 8 is exactly the 600^3 case; a step = RCCL all-gather of the x shards
          over xGMI + the local planned SpMV.  Per-GPU work is fixed: "scaling": "weak".
          --exchange auto (default) sends only the column-range halo each slab references (one 600^2 plane per
          neighbour, 5.8 MB per GPU); --exchange allgather moves every shard to every GPU (1.5 GB per GPU at N = 8).

Protocol (mirrors perf_test/sparse/KokkosSparse_kk_spmv.cpp:121-167): inputs generated in HBM, handle/plan
creation outside the timed region, W warm-ups, then exactly K steps bracketed by barrier + device sync,
max over ranks; EVERY step is followed by a fence of the execution space and timed on its own, as the reference's
loop does (`space.fence(); timer.reset(); spmv; space.fence(); totalTime += timer.seconds()`), so the line carries the
mean (= ms_per_step), min and max of the per-call times (--no-fence queues the K steps back to back instead; --flush
adds the reference's --flush option: 4 x 1 GB of fills between calls, outside the per-call timers, reported beside).
alpha = 1, beta = 0 (the driver's default), x/y = integers in [-20, 20) as fp64.
GFLOP/s = 2 * stored nnz / t (stored zeros counted, like the reference's drivers).

N = 1 also measures, under a time cap (--extras-seconds, default 150; 0 = off), the other SURVEY 8(d) metrics so that the
driver's record holds them: "spmv_mv" = BASELINE config 3 (the same matrix x 16 right-hand sides, LayoutRight and
LayoutLeft X / Y) and "spgemm" = the largest single-GPU member of config 4 (C = A*A on R-MAT scale 20, edge factor 16:
symbolic, numeric, numeric reuse), each with its CPU baseline (OpenMP ports in oracle/, bounded samples).

The JSON line also carries
  roofline     algorithmic bytes of ONE local SpMV (nnz*12 + (rows+1)*4 + x touched*8 + rows*8)
               / the SpMV's average duration measured with HIP events on the launch stream, vs 8 TB/s HBM3E;
  cpu_baseline the OpenMP port of the reference's host SpMV (oracle/kk_oracle_omp.c) on a bounded sample
               of the same workload, timed on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is what a plain copy reaches
PMC_FILE = os.path.join(ROOT, "profiles", "round6", "bench_n1_pmc_hbm.json")
MV_PMC_FILE = os.path.join(ROOT, "profiles", "round6", "mv4_pmc_hbm.json")


def kernel_source_sha(unit="kk_spmv.hip"):
    """sha256 over one kernel translation unit and the headers every unit shares: stamps every committed counter file, so that
    bench.py can tell whether the traffic figure it quotes was measured on the code it is running (.git does not travel to
    the GPU box).  The rank-1 SpMV the bench times lives in kk_spmv.hip, config 3's kernels in kk_spmv_mv.hip, SpGEMM in kk_spgemm.hip:
    a file is stamped with the hash of the unit it measures (tools/extract_profiles.py: unit_of_tag)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kokkos-kernels_amd", "csrc")
    for f in (unit, "kk_spmv_plan.h", "kk_common.h", "kk_rt.h", "kk_scan.h"):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def all_unit_shas():
    """{unit: sha} for every kernel translation unit (stamped into summaries that span several units, e.g. the whole bench line)"""
    d = os.path.join(ROOT, "kokkos-kernels_amd", "csrc")
    return {f: kernel_source_sha(f) for f in sorted(os.listdir(d)) if f.endswith(".hip")}


def cpu_baseline(sample_n=300, min_seconds=6.0):
    """Reference host path (port) on the workload's own matrix (27-pt FE Laplacian 300^3; a 160^3 sample only if the host
    cannot hold it), all usable host cores, about ten seconds of SpMVs.  N > 1: rank 0 runs it after the timed region on the
    same 27,000,000-row problem one GPU's slab holds (the other ranks wait at the closing barrier)."""
    # thread placement Kokkos recommends for its OpenMP backend; must be set before the OpenMP runtime starts
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "threads")
    import numpy as np
    import oracle
    oracle.set_omp_threads(oracle.usable_cpus())     # the cgroup CPU quota, not the visible core count (see usable_cpus)
    try:
        A = oracle.laplace3d("FE", sample_n, sample_n, sample_n)
        note = "the full workload matrix"
    except MemoryError:
        sample_n = 160
        A = oracle.laplace3d("FE", sample_n, sample_n, sample_n)
        note = "FALLBACK sample (the host could not hold 300^3)"
    ft = oracle.first_touch          # pages spread over NUMA nodes the way a Kokkos::View's would be
    rm32 = ft(A.row_map.astype(np.int32)); ent = ft(A.entries); val = ft(A.values)
    rng = np.random.default_rng(17312837)
    x = ft(rng.integers(-20, 20, size=A.ncols).astype(np.float64))
    y = ft(np.zeros(A.nrows))
    A.entries, A.values = ent, val
    oracle.spmv_omp(rm32, A.entries, A.values, 1.0, x, 0.0, y)        # warm-up
    t0 = time.perf_counter(); it = 0
    while True:
        oracle.spmv_omp(rm32, A.entries, A.values, 1.0, x, 0.0, y); it += 1
        el = time.perf_counter() - t0
        if el >= min_seconds and it >= 5:
            break
    gflops = 2.0 * A.nnz * it / el / 1e9
    gbps = (A.nnz * 12 + (A.nrows + 1) * 4 + A.ncols * 8 + A.nrows * 8) * it / el / 1e9
    return {"value": round(gflops, 3), "unit": "GFLOP/s", "cores": oracle.omp_threads(), "kind": "port",
            "sample": "%s: 27-pt FE Laplacian %d^3 (%d rows, %d nnz), fp64, OpenMP dynamic schedule (nnz > 1e7), %d iterations, %.1f algorithmic GB/s"
                      % (note, sample_n, A.nrows, A.nnz, it, gbps)}


def _fenced(fn, sync, reps, flush=None):
    """the reference's timing loop (KokkosSparse_kk_spmv.cpp:139-167): fence, timer, call, fence -- per call"""
    ts = []
    for _ in range(reps):
        if flush is not None:
            for rep in range(4): flush.fill_(rep + 1)
        sync()
        t_ = time.perf_counter(); fn(); sync()
        ts.append((time.perf_counter() - t_) * 1e3)
    return {"mean_ms": round(sum(ts) / len(ts), 5), "min_ms": round(min(ts), 5), "max_ms": round(max(ts), 5), "calls": len(ts)}


def bench_spmv_mv(kk, torch, A, budget_s, cpu):
    """BASELINE config 3: Y = A X with 16 fp64 right-hand sides on the bench matrix (27-pt 300^3), X / Y LayoutRight (row-major)
    and LayoutLeft (column-major, the reference's default layout), one analysed handle each, the reference's per-call-fence loop.
    Algorithmic bytes (beta = 0): nnz*12 + (rows+1)*4 + 2*rows*16*8."""
    t_start = time.perf_counter()
    nv, rows, nnz = 16, A.numRows(), A.nnz()
    alg = nnz * 12 + (rows + 1) * 4 + 2 * rows * nv * 8
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    X = torch.randint(-20, 20, (A.numCols(), nv), device="cuda", generator=g).double()
    out = {"workload": "spmv_mv_crs_27pt_FE_laplacian_300^3_x16_fp64", "nvec": nv, "algorithmic_bytes_per_call": alg, "bound": "hbm", "peak_GBps": HBM_PEAK_GBPS}
    ref = None
    for layout in ("right", "left"):
        Xl = X if layout == "right" else X.t().contiguous().t()
        Y = torch.full((rows, nv), float("nan"), dtype=torch.float64, device="cuda") if layout == "right" \
            else torch.full((nv, rows), float("nan"), dtype=torch.float64, device="cuda").t()
        h = kk.SPMVHandle("SPMV_DEFAULT")
        fn = lambda: kk.spmv(h, "N", 1.0, A, Xl, 0.0, Y)
        torch.cuda.synchronize(); t_first = time.perf_counter(); fn(); torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t_first) * 1e3            # the handle's first call: analysis (plan creation) + one product
        for _ in range(4): fn()
        torch.cuda.synchronize()
        r = _fenced(fn, torch.cuda.synchronize, 20)
        r["first_call_ms"] = round(first_ms, 3); r["plan_create_ms"] = round(first_ms - r["mean_ms"], 3)
        if ref is None: ref = Y.clone()
        else: assert float((Y - ref).abs().max()) == 0.0, "spmv_mv: layouts disagree"
        r.update(GFLOPs=round(2.0 * nnz * nv / r["mean_ms"] / 1e6, 1), achieved_GBps=round(alg / r["mean_ms"] / 1e6, 1),
                 frac=round(alg / r["mean_ms"] / 1e6 / HBM_PEAK_GBPS, 4),
                 kernel=_mv_kernel_name(h))
        r.update(_mv_traffic(layout, r["mean_ms"]))
        out["layout_" + layout] = r
        del h, Y
    if cpu and time.perf_counter() - t_start < budget_s:
        out["cpu_baseline"] = cpu_baseline_mv(nv)
    return out


def _mv_traffic(layout, mean_ms):
    """memory-side bytes per launch of the plane-marching kernel from the committed counter file (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes
    of tools/bench_mv4.py; FETCH_SIZE x 2 on gfx950), quoted only when the file was measured on the kernel source this library was built from"""
    sha = kernel_source_sha("kk_spmv_mv4.h")
    out = {"traffic": None, "kernel_source_sha": sha}
    try:
        d = json.load(open(MV_PMC_FILE))
        if d.get("kernel_source_sha") != sha:
            out["traffic_source"] = "%s is stale (measured on kernel source %s): traffic not quoted" % (os.path.relpath(MV_PMC_FILE, ROOT), d.get("kernel_source_sha"))
            return out
        tag = "true, 1, false>" if layout == "right" else "true, 2, false>"  # <.., BETA0 (the bench runs beta = 0), X layout mode (row-major 1, column-major 2), PART>
        rd = wr = 0.0
        for k, v in d.get("counters", {}).items():
            if "spmv_mv4_kernel" in k and tag in k:
                if "FETCH_SIZE" in k: rd = v["mean_KB"] * 1024 * 2
                if "WRITE_SIZE" in k: wr = v["mean_KB"] * 1024
        if rd and wr:
            out.update(traffic=int(rd + wr), moved_frac=round((rd + wr) / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                       traffic_source="%s (FETCH_SIZE x 2 + WRITE_SIZE per launch)" % os.path.relpath(MV_PMC_FILE, ROOT))
    except Exception:
        pass
    return out


def _mv_kernel_name(h):
    if h.query("mv4_workgroups"): return "kk::spmv_mv4_kernel (plane marching)"
    if h.query("mv5_tiles"): return "kk::spmv_mv5_kernel (matrix core, v_mfma_f64_16x16x4)"
    if h.query("mv6_chunks"): return "kk::spmv_mv6_kernel (nonzero split)"
    return "kk::spmv_mv2_kernel (gather)"


def bench_spmv_mv_off_lattice(kk, torch, budget_s):
    """SpMV_MV with 16 fp64 right-hand sides OFF the lattice: a block-diagonal matrix (32 x 32 blocks, 2e6 rows) and a
    3-dof-per-node 27-pt finite-element matrix (100^3 nodes) -- both taken by the matrix-core kernel --, and R-MAT scale 22 (the
    nonzero-split kernel); LayoutRight, and LayoutLeft for the first.  Algorithmic bytes nnz*12 + (rows+1)*4 + (rows+cols)*16*8."""
    import numpy as np
    t_start = time.perf_counter()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_plan_table as pt
    out = {"nvec": 16, "peak_GBps": HBM_PEAK_GBPS, "cases": {}}
    for name, layouts in (("block diagonal, 32 x 32 blocks, 2e6 rows", ("right", "left")), ("3 dof per node on 27-pt FE 100^3", ("right",)), ("R-MAT scale 22, edge factor 16", ("right",))):
        if time.perf_counter() - t_start > budget_s: break
        (_, A), = list(pt.matrices(name))
        rows, cols, nnz, nv = A.numRows(), A.numCols(), A.nnz(), 16
        alg = nnz * 12 + (rows + 1) * 4 + (rows + cols) * nv * 8
        g = torch.Generator(device="cuda"); g.manual_seed(7)
        X = torch.rand(cols, nv, dtype=torch.float64, device="cuda", generator=g)
        case = {"rows": rows, "nnz": nnz, "algorithmic_bytes_per_call": alg}
        for layout in layouts:
            Xl = X if layout == "right" else X.t().contiguous().t()
            Y = torch.zeros(rows, nv, dtype=torch.float64, device="cuda") if layout == "right" else torch.zeros(nv, rows, dtype=torch.float64, device="cuda").t()
            h = kk.SPMVHandle("SPMV_DEFAULT")
            fn = lambda: kk.spmv(h, "N", 1.0, A, Xl, 0.0, Y)
            torch.cuda.synchronize(); t_first = time.perf_counter(); fn(); torch.cuda.synchronize()
            first_ms = (time.perf_counter() - t_first) * 1e3        # analysis (plan creation, for the matrix-core kernel its tile descriptors) + one product
            for _ in range(2): fn()
            torch.cuda.synchronize()
            r = _fenced(fn, torch.cuda.synchronize, 20)
            r["first_call_ms"] = round(first_ms, 3); r["plan_create_ms"] = round(first_ms - r["mean_ms"], 3)
            r.update(frac=round(alg / r["mean_ms"] / 1e6 / HBM_PEAK_GBPS, 4), kernel=_mv_kernel_name(h))
            if h.query("mv5_tiles"): r["mfma_instructions_per_call"] = h.query("mv5_blocks"); r["operand_fill"] = h.query("mv5_fill_permille") / 1000
            case["layout_" + layout] = r
            del h, Y
        out["cases"][name] = case
        del A, X
        torch.cuda.empty_cache()
    return out


def cpu_baseline_mv(nv, n=300, min_seconds=6.0):
    """the reference's host SpMV_MV functor (OpenMP port, oracle/kk_oracle_omp.c) on the workload's own matrix, LayoutRight"""
    import numpy as np
    import oracle
    oracle.set_omp_threads(oracle.usable_cpus())
    A = oracle.laplace3d("FE", n, n, n)
    ft = oracle.first_touch
    rm32 = ft(A.row_map.astype(np.int32)); ent = ft(A.entries); val = ft(A.values)
    rng = np.random.default_rng(3)
    X = ft(rng.integers(-20, 20, size=(A.ncols, nv)).astype(np.float64)); Y = ft(np.zeros((A.nrows, nv)))
    oracle.spmv_mv_omp(rm32, ent, val, 1.0, X, 0.0, Y)
    t0 = time.perf_counter(); it = 0
    while True:
        oracle.spmv_mv_omp(rm32, ent, val, 1.0, X, 0.0, Y); it += 1
        el = time.perf_counter() - t0
        if el >= min_seconds and it >= 3: break
    return {"value": round(2.0 * A.nnz * nv * it / el / 1e9, 3), "unit": "GFLOP/s", "cores": oracle.omp_threads(), "kind": "port",
            "sample": "the full workload: 27-pt FE Laplacian %d^3 x %d right-hand sides, LayoutRight, %d iterations" % (n, nv, it)}


def bench_rank1_gather_bound(kk, torch):
    """Rank-1 SpMV off the stencil: uniform random columns, 5e6 rows x 20 (1e8 nonzeros, x = 40 MB), fp64.
    The CRS stream kernel (colslab 0), the default handle (round 5: the DETERMINISTIC column-slab form, chosen by rule at the first call
    -- per-slab partial sums, no atomics, bit-stable; its values follow A.values exactly through a shadow comparison), the same with the
    caller notifying value changes (no per-call pass), and the atomic forms of round 3 / 4 (opt-in); fractions are CRS bytes
    (nnz*12 + (rows+1)*4 + cols*8 + rows*8) at 8 TB/s."""
    n, k = 5_000_000, 20
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    c = torch.sort(torch.randint(0, n, (n, k), device="cuda", generator=g), dim=1).values
    rm = (torch.arange(n + 1, device="cuda", dtype=torch.int64) * k).to(torch.int32)
    val = torch.rand(n * k, device="cuda", dtype=torch.float64, generator=g) + 0.5
    A = kk.CrsMatrix(n, n, rm, c.reshape(-1).to(torch.int32).contiguous(), val); del c
    x = torch.randint(-20, 20, (n,), device="cuda", generator=g).double(); y = torch.zeros(n, dtype=torch.float64, device="cuda")
    alg = A.nnz() * 12 + (n + 1) * 4 + 2 * n * 8
    out = {"workload": "spmv_crs_uniform_random_5e6x20_fp64", "nnz": A.nnz(), "algorithmic_bytes_per_call": alg, "peak_GBps": HBM_PEAK_GBPS}
    ref = None
    for tag, knobs in (("crs_kernel", {"colslab": 0}), ("default_handle", {}), ("default_handle_caller_notifies", {"values_tracking": 1}),
                       ("column_slab_copy_opt_in", {"colslab": 1}), ("column_slab_copy_caller_notifies", {"colslab": 2, "values_tracking": 1})):
        h = kk.SPMVHandle("SPMV_DEFAULT")
        for k_, v_ in knobs.items(): h.set(k_, v_)
        fn = lambda: kk.spmv(h, "N", 1.0, A, x, 0.0, y)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        r = _fenced(fn, torch.cuda.synchronize, 20)
        r["frac"] = round(alg / r["mean_ms"] / 1e6 / HBM_PEAK_GBPS, 4)
        if ref is None: ref = y.clone()
        r["max_rel_diff_vs_crs"] = float(((y - ref).abs().max() / ref.abs().max()).item())
        assert r["max_rel_diff_vs_crs"] < 1e-12, "rank-1 gather-bound case: kernels disagree"
        if tag == "column_slab_copy_opt_in": r["column_slab_copy_selected"] = int(h.query("colslab")); r["selection_us_crs_vs_copy"] = [h.query("colslab_crs_us"), h.query("colslab_us")]
        if tag in ("default_handle", "crs_kernel"):
            r["column_slab_form"] = int(h.query("colslab")); r["deterministic"] = bool(h.query("colslab_deterministic")) or not h.query("colslab")
            r["lines_of_x_per_nonzero_sampled"] = h.query("colslab_lines_permille") / 1000
            y_a = y.clone(); fn(); torch.cuda.synchronize()
            r["bit_stable_across_calls"] = bool((y_a == y).all().item())
        out[tag] = r
        del h
    return out


def _spgemm_case(kk, torch, scale, t_start, budget_s, reps=3):
    """C = A*A on R-MAT `scale`: one warm-up, then `reps` repetitions with a fresh handle each (symbolic, numeric, numeric again on the same handle)"""
    import numpy as np
    import oracle
    R = oracle.rmat(scale, 16)
    M = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
    sync = torch.cuda.synchronize
    runs = []; warm = []
    WARM = 2        # not counted: the first use of every kernel, and the product after it -- the library returns its store of bitmaps / entry lists
                    # to the device with a process's first product and keeps it once the process comes back for more (a hipMalloc of GBs after a
                    # hipFree stalls for a second every other time on this runtime: profiles/round5/probe_malloc.txt)
    for rep in range(reps + WARM):
        kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
        sync(); t0 = time.perf_counter()
        Cm = kk.spgemm_symbolic(kh, M, False, M, False)
        sync(); t1 = time.perf_counter()
        kk.spgemm_numeric(kh, M, False, M, False, Cm)
        sync(); t2 = time.perf_counter()
        kk.spgemm_numeric(kh, M, False, M, False, Cm)
        sync(); t3 = time.perf_counter()
        sh = kh.get_spgemm_handle(); mults = sh.get(1); nnzC = Cm.nnz(); nblk_rows = sh.get(16)
        if rep >= WARM: runs.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
        else: warm.append([round((t1 - t0) * 1e3, 3), round((t2 - t1) * 1e3, 3), round((t3 - t2) * 1e3, 3)])
        kh.destroy_spgemm_handle(); del Cm
        if runs and time.perf_counter() - t_start > budget_s: break
    sym, num, reuse = (sum(r[i] for r in runs) / len(runs) for i in range(3))
    b_sym = R.nnz * 4 + 2 * (R.nrows + 1) * 8 + mults * 4
    b_num = R.nnz * 12 + 2 * (R.nrows + 1) * 8 + mults * 12 + nnzC * 12
    out = {"workload": "spgemm_AxA_rmat_scale%d_ef16_fp64_int32_ordinals_int64_offsets" % scale, "rows": R.nrows, "nnz_A": R.nnz,
           "multiplications": mults, "nnz_C": nnzC, "repetitions": len(runs), "protocol": "two warm-up products, then a fresh handle per repetition, fence after every phase, mean of the repetitions",
           "symbolic_ms": round(sym, 3), "numeric_ms": round(num, 3), "numeric_reuse_ms": round(reuse, 3), "total_ms": round(sym + num, 3),
           "min_ms": {"symbolic": round(min(r[0] for r in runs), 3), "numeric": round(min(r[1] for r in runs), 3), "numeric_reuse": round(min(r[2] for r in runs), 3)},
           "GFLOPs": round(2.0 * mults / (sym + num) / 1e6, 1), "numeric_GFLOPs": round(2.0 * mults / num / 1e6, 1),
           "rows_through_column_block_kernel": nblk_rows,
           "first_two_products_of_the_process_ms": {"what": "[symbolic, numeric, numeric reuse] of the two warm-up products: the first loads the code objects and sizes the "
                                                     "library's store of kept structure, the second may pay the runtime's slow hipMalloc after the store was returned "
                                                     "(knob spgemm_pool_keep; profiles/round5/probe_malloc.txt)", "products": warm},
           "roofline": {"bound": "hbm", "model": "gather model, SURVEY 8(d)", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "symbolic": {"bytes": b_sym, "achieved": round(b_sym / sym / 1e6, 1), "frac": round(b_sym / sym / 1e6 / HBM_PEAK_GBPS, 4)},
                        "numeric": {"bytes": b_num, "achieved": round(b_num / num / 1e6, 1), "frac": round(b_num / num / 1e6 / HBM_PEAK_GBPS, 4)},
                        "numeric_reuse": {"bytes": b_num - nnzC * 4, "achieved": round((b_num - nnzC * 4) / reuse / 1e6, 1),
                                          "frac": round((b_num - nnzC * 4) / reuse / 1e6 / HBM_PEAK_GBPS, 4)}}}
    del M
    return out


def _spgemm_c4_slabs(kk, torch, t_start, budget_s, world=8, scale=22, verbose=0, ranks=None):
    """BASELINE config 4 as specified (R-MAT scale 22, edge factor 16) does not fit one GPU (nnz(C) = 7.2e10 = 863 GB); one rank's slab of the
    8-GPU row partition does (B replicated, no data-path communication: SURVEY 8e).  kkamd_dist_spgemm_partition(world = 8) cuts A into slabs of
    near-equal multiplications; the slab with the fewest rows (hub rows) and the one with the most rows (the largest piece of C) are timed through
    kkamd_dist_spgemm_symbolic / _numeric: two warm-ups, then the mean of two repetitions with a fresh operator each.  What an 8-GPU run would take is the
    slowest slab (no exchange)."""
    import numpy as np
    import oracle
    from kokkos_kernels_amd.dist import DistSpgemm
    R = oracle.rmat(scale, 16)
    M = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
    be = M.backend
    sync = torch.cuda.synchronize
    offsets, mults = DistSpgemm.partition(M, M, world)
    rows = [offsets[r + 1] - offsets[r] for r in range(world)]
    out = {"workload": "spgemm_AxA_rmat_scale%d_ef16_fp64_row_slabs_of_%d_B_replicated" % (scale, world), "rows": R.nrows, "nnz_A": R.nnz,
           "multiplications": int(sum(mults)), "multiplications_per_rank": [int(v) for v in mults], "rows_per_rank": rows,
           "balance_max_over_mean": round(max(mults) / (sum(mults) / world), 4), "slabs": {}}
    for rank in (ranks if ranks is not None else sorted({int(np.argmin(rows)), int(np.argmax(rows))})):
        if time.perf_counter() - t_start > budget_s: out["slabs"]["rank %d" % rank] = {"skipped": "time cap"}; continue
        o0, o1 = offsets[rank], offsets[rank + 1]
        e0, e1 = int(R.row_map[o0]), int(R.row_map[o1])
        Sd = kk.CrsMatrix(o1 - o0, R.ncols, (M.graph.row_map[o0:o1 + 1] - M.graph.row_map[o0]).contiguous(), M.graph.entries[e0:e1], M.values[e0:e1], backend=be)
        runs = []
        WARM = 2            # (as in _spgemm_case: the first use of every kernel, and the product after it, in which the library's store is sized for this slab)
        for rep in range(WARM + 2):
            op = DistSpgemm(offsets, rank, be)
            if verbose and rep == WARM + 1: op.set("verbose", verbose)
            sync(); t0 = time.perf_counter()
            Cm = op.symbolic(Sd, M)
            sync(); t1 = time.perf_counter()
            op.numeric(Sd, M, Cm)
            sync(); t2 = time.perf_counter()
            op.numeric(Sd, M, Cm)
            sync(); t3 = time.perf_counter()
            nnzC = Cm.nnz()
            if rep >= WARM: runs.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            del Cm, op          # (the slab of C stays in torch's cache for the next repetition: returning 100+ GB to the runtime and asking for them
                                # again stalls the next hipMalloc by seconds, profiles/round5/probe_malloc.txt)
        torch.cuda.empty_cache()
        sym, num, reuse = (sum(r[i] for r in runs) / len(runs) for i in range(3))
        nnzA_s, mu = e1 - e0, int(mults[rank])
        b_sym = nnzA_s * 4 + (o1 - o0 + 1) * 16 + (R.nrows + 1) * 8 + mu * 4
        b_num = nnzA_s * 12 + (o1 - o0 + 1) * 16 + (R.nrows + 1) * 8 + mu * 12 + nnzC * 12
        out["slabs"]["rank %d" % rank] = {
            "rows": o1 - o0, "nnz_A_slab": nnzA_s, "multiplications": mu, "nnz_C_slab": nnzC, "C_slab_GB": round(nnzC * 12e-9, 1),
            "symbolic_ms": round(sym, 3), "numeric_ms": round(num, 3), "numeric_reuse_ms": round(reuse, 3),
            "GFLOPs": round(2.0 * mu / (sym + num) / 1e6, 1),
            "roofline": {"bound": "hbm", "model": "gather model, SURVEY 8(d)", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "symbolic_frac": round(b_sym / sym / 1e6 / HBM_PEAK_GBPS, 4), "numeric_frac": round(b_num / num / 1e6 / HBM_PEAK_GBPS, 4),
                         "numeric_reuse_frac": round((b_num - nnzC * 4) / reuse / 1e6 / HBM_PEAK_GBPS, 4)}}
        del Sd
    done = [v for v in out["slabs"].values() if "symbolic_ms" in v]
    if done:
        out["slowest_slab_ms"] = {"symbolic": max(v["symbolic_ms"] for v in done), "numeric": max(v["numeric_ms"] for v in done)}
        out["projected_8gpu_GFLOPs_if_every_slab_took_the_slowest_time"] = round(2.0 * sum(mults) / (out["slowest_slab_ms"]["symbolic"] + out["slowest_slab_ms"]["numeric"]) / 1e6, 1)
    del M
    return out


def bench_spgemm(kk, torch, budget_s, cpu, scale=20):
    """BASELINE config 4 family: C = A*A on R-MAT (edge factor 16, 64-bit offsets), the largest scale whose C fits one GPU
    (scale 20: nnz(C) 9.69e9 = 116 GB; scale 22 as specified needs 863 GB).  Per repetition a fresh handle: symbolic, numeric,
    numeric again on the same handle (the reference's reuse case).  Gather model of SURVEY 8(d): symbolic nnz(A)*4 + mults*4 +
    2 row maps, numeric nnz(A)*12 + mults*12 + nnz(C)*12 + 2 row maps, against 8 TB/s.  The CPU baseline (OpenMP port of SPGEMM_KK)
    runs on scale 18 -- nnz(C) at scale 20 is 116 GB on the host --, so the GPU's scale-18 times stand beside it ("same_matrix_as_cpu_baseline")."""
    t_start = time.perf_counter()
    if torch.cuda.mem_get_info()[0] < 170 * 2**30: scale = 18
    out = _spgemm_case(kk, torch, scale, t_start, budget_s)
    if scale != 18:
        out["same_matrix_as_cpu_baseline"] = _spgemm_case(kk, torch, 18, t_start, budget_s)
    if cpu and time.perf_counter() - t_start < budget_s:
        out["cpu_baseline"] = cpu_baseline_spgemm()
        ref18 = out.get("same_matrix_as_cpu_baseline", out)
        out["cpu_baseline"]["gpu_over_cpu_same_matrix"] = round(ref18["GFLOPs"] / out["cpu_baseline"]["value"], 1)
    # config 4 at its stated scale: one rank's slab of R-MAT scale 22 (needs most of the GPU: after everything else, when the time cap leaves a minute)
    torch.cuda.empty_cache()
    if scale != 18 and torch.cuda.mem_get_info()[0] >= 220 * 2**30 and budget_s - (time.perf_counter() - t_start) >= 45:
        try:
            out["c4_slab_of_8"] = _spgemm_c4_slabs(kk, torch, t_start, budget_s)
        except Exception as e:
            out["c4_slab_of_8"] = {"error": repr(e)[:200]}
    else:
        out["c4_slab_of_8"] = {"skipped": "time cap or memory"}
    return out


def cpu_baseline_spgemm(scale=18):
    """SPGEMM_KK on the host cores: OpenMP port of the reference's KKMEM hash kernels (oracle/kk_oracle_omp.c) + its row sort, on a
    bounded sample of the same family (R-MAT scale 18: 2.9e9 multiplications)"""
    import oracle
    oracle.set_omp_threads(oracle.usable_cpus())
    R = oracle.rmat(scale, 16)
    tm = {}
    Cm = oracle.spgemm_kkmem_omp(R, R, sort=True, timings=tm)
    mults = oracle.spgemm_mults(R, R)[0]
    tot = tm["symbolic_s"] + tm["numeric_s"] + tm["sort_s"]
    return {"value": round(2.0 * mults / tot / 1e9, 3), "unit": "GFLOP/s", "cores": oracle.omp_threads(), "kind": "port",
            "symbolic_ms": round(tm["symbolic_s"] * 1e3, 1), "numeric_ms": round(tm["numeric_s"] * 1e3, 1), "sort_ms": round(tm["sort_s"] * 1e3, 1),
            "sample": "R-MAT scale %d ef 16, C = A*A (%d multiplications, nnz(C) %d): symbolic + numeric + row sort" % (scale, mults, Cm.nnz)}


def self_launch(n):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): run the same command line as N ranks of one node
    under torch.distributed.run -- one process per GPU, rendezvous on 127.0.0.1 at a free port -- and pass its output and exit code
    through.  Under torch.distributed.run (the driver's form) WORLD_SIZE is set and this is never reached."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "1")
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: launching %s\n" % (n, " ".join(cmd[1:9])))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid-edge", "--n", dest="n", type=int, default=0,
                    help="override the grid edge (debug: smaller problem); spell it --grid-edge under torch.distributed.run, "
                         "whose own parser claims every unambiguous prefix of its options")
    ap.add_argument("--beta", type=float, default=0.0)
    ap.add_argument("--algo", default="SPMV_DEFAULT")
    ap.add_argument("--knob", action="append", default=[], help="key=value expert knob for the SpMV plan")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fence", action="store_true", help="queue the K timed steps back to back (no per-call fence, no per-call times)")
    ap.add_argument("--flush", action="store_true", help="also report the per-call times with the reference's --flush (4 x 1 GB of fills between calls)")
    ap.add_argument("--extras-seconds", type=float, default=180.0,
                    help="N = 1: time cap for the spmv_mv (config 3) and spgemm (config 4) sections of the line; 0 skips them")
    ap.add_argument("--emulate", action="store_true",
                    help="debug / CI: run the whole flow on CPU -- gloo instead of RCCL, the kernels under the SIMT emulator of "
                         "tests/emu, a tiny grid -- to exercise the multi-process control flow without GPUs (numbers are meaningless)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1, halo exchange: do not split the slab into interior / boundary rows (no compute-communication overlap)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "halo", "halo_set", "allgather", "allgather_collective", "allgather_p2p"],
                    help="N > 1: how the x entries a slab references reach it (auto = column-range halo when it is smaller)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    import numpy as np
    import torch
    import kk_loader
    kk = kk_loader.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    emu = args.emulate
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu import emu_backend
        be, dev, tb = emu_backend.backend(), "cpu", (lambda t: t.numpy())

        class Event:                               # stands in for torch.cuda.Event
            def __init__(self, enable_timing=True): self.t = 0.0
            def record(self): self.t = time.perf_counter()
            def elapsed_time(self, other): return (other.t - self.t) * 1e3
        device_sync = lambda: None
    else:
        torch.cuda.set_device(local_rank)
        be, dev, tb, Event, device_sync = None, "cuda", (lambda t: t), torch.cuda.Event, torch.cuda.synchronize
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- workload ---------------------------------------------------------------------------
    if emu and not args.n:
        args.n = 16
    if world == 1:
        n = args.n or 300
        nx = ny = nz = n
        planes_per_rank = nz
        workload = "spmv_crs_27pt_FE_laplacian_%d^3_fp64" % n
    else:
        n = args.n or 600
        nx = ny = n
        planes_per_rank = (args.n // 8) if args.n else 75
        nz = planes_per_rank * world
        workload = "spmv_crs_27pt_FE_laplacian_%dx%dx%d_fp64_row_slabs_%d_planes_per_gpu" % (nx, ny, nz, planes_per_rank)
    rows_per_rank = nx * ny * planes_per_rank
    nrows_global = nx * ny * nz
    A = kk.laplace_matrix("FE", nx, ny, nz, rows=(rank * rows_per_rank, rows_per_rank), backend=be)
    nnz_local = A.nnz()
    g = torch.Generator(device=dev); g.manual_seed(17312837 + rank)
    x_shard = torch.randint(-20, 20, (rows_per_rank,), device=dev, generator=g).double()
    y_shard = torch.randint(-20, 20, (rows_per_rank,), device=dev, generator=g).double()
    alpha, beta = 1.0, args.beta
    # N = 1 with a named exchange: the one rank still goes through the row-partitioned operator and the library's own RCCL transport
    # (a one-rank communicator: ncclAllGather in place / groups without peers) -- the N-rank code path and its keys on one GPU
    use_dist = world > 1 or (args.exchange != "auto" and not emu)

    if not use_dist:
        handle = kk.SPMVHandle(args.algo)
        for kv in args.knob:
            k, v = kv.split("="); handle.set(k, int(v))
        def spmv_only(): kk.spmv(handle, "N", alpha, A, tb(x_shard), beta, tb(y_shard))
        def step(ev0, ev1):
            ev0.record(); spmv_only(); ev1.record()
    else:
        from kokkos_kernels_amd.dist import DistSpmv
        offsets = [r * rows_per_rank for r in range(world + 1)]
        lib_ = (be or kk.torch_backend()).lib
        for kv in args.knob:                       # the slab plans are created inside the library: knobs go in as defaults
            k, v = kv.split("="); kk._capi.check(lib_, lib_.kkamd_set_default(k.encode(), int(v)))
        op = DistSpmv(A, offsets, rank, algo=args.algo, exchange=args.exchange, overlap=not args.no_overlap,
                      to_backend=tb if emu else None, transport="rccl" if world == 1 else None)
        xl = op.x_local(); xl.copy_(x_shard); x_shard = xl      # x lives in the operator's window: no per-step copy
        def step(ev0, ev1):
            op.apply(alpha, x_shard, beta, y_shard, events=(ev0, ev1))

    def barrier():
        if dist is not None:
            dist.barrier()
        device_sync()

    evs = [(Event(enable_timing=True), Event(enable_timing=True)) for _ in range(args.steps)]
    w0, w1 = Event(enable_timing=True), Event(enable_timing=True)
    # the handle's FIRST call (outside the W warm-up steps and the timed region): analysis of the matrix (plan creation) + one SpMV
    device_sync(); t_first = time.perf_counter(); step(w0, w1); device_sync()
    first_call_ms = (time.perf_counter() - t_first) * 1e3
    for _ in range(args.warmup):
        step(w0, w1)
    barrier()
    per_call = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        if args.no_fence:
            step(*evs[i])
        else:                                   # the reference's loop: fence, timer, call, fence
            tc = time.perf_counter(); step(*evs[i]); device_sync()
            per_call.append((time.perf_counter() - tc) * 1e3)
    barrier()
    t1 = time.perf_counter()
    ms_step = (t1 - t0) * 1e3 / args.steps
    kern_ms = sum(a.elapsed_time(b) for a, b in evs) / args.steps      # local SpMV kernels (N > 1 with overlap: incl. the wait for the halo)

    nnz_total = torch.tensor([float(nnz_local)], device=dev, dtype=torch.float64)
    pc_min, pc_max = (min(per_call), max(per_call)) if per_call else (0.0, 0.0)
    tmax = torch.tensor([ms_step, kern_ms, pc_min, pc_max], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(nnz_total, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    nnz_global = int(nnz_total.item()); ms_step, kern_ms, pc_min, pc_max = tmax.tolist()
    flushed = None
    if args.flush and not emu:
        fl = torch.empty(1 << 30, dtype=torch.int8, device=dev)
        w2 = (Event(enable_timing=True), Event(enable_timing=True))
        flushed = _fenced(lambda: step(*w2), barrier, max(3, min(args.steps, 20)), flush=fl)
        del fl

    # ---- N > 1: what each exchange costs (outside the timed region above): halo and all-gather steps, and the exchange alone ----
    xchg = None
    def all_reduce_(t, how):
        if dist is not None: dist.all_reduce(t, op=how)
    RED_MIN, RED_MAX = (dist.ReduceOp.MIN, dist.ReduceOp.MAX) if dist is not None else (None, None)
    if use_dist:
        def timed(fn, n=max(3, min(args.steps, 20))):
            fn(); barrier()
            t_ = time.perf_counter()
            for _ in range(n): fn()
            barrier()
            return (time.perf_counter() - t_) * 1e3 / n
        xchg = {}
        ops = {op.exchange_mode: op}
        # every exchange the library has, side by side: the column-range halo, the column-set halo, the all-gather through the
        # collective and the all-gather by peer-to-peer pulls (the fallback if RCCL's schedule is a ring; it maps
        # device memory between processes, so not under --emulate)
        for other in ("halo", "halo_set", "allgather", "allgather_collective") + (() if emu else ("allgather_p2p",)):
            if other in ops: continue
            o_new, ok_ = None, 1.0
            try:
                o_new = DistSpmv(A, offsets, rank, algo=args.algo, exchange=other, overlap=not args.no_overlap, to_backend=tb if emu else None,
                                 transport="rccl" if world == 1 else None)
            except Exception as e:                 # e.g. not enough memory for another operator: report what there is
                ok_ = 0.0
                xchg.setdefault("notes", []).append("no %s operator on rank %d: %s" % (other, rank, str(e)[:120]))
            okt = torch.tensor([ok_], device=dev, dtype=torch.float64)
            all_reduce_(okt, RED_MIN)          # an operator only counts when EVERY rank has it (else the collectives would not match)
            if okt.item() == 1.0: ops[other] = o_new
            else:
                del o_new
                if ok_ == 1.0: xchg.setdefault("notes", []).append("no %s operator: another rank could not create it" % other)
        for name, o in ops.items():
            xs_ = o.x_local()
            if o is not op: xs_.copy_(x_shard)
            t_step = timed(lambda: o.apply(alpha, xs_, beta, y_shard))
            t_x = timed(lambda: o.apply(alpha, xs_, beta, y_shard, what=1))
            t_l = timed(lambda: o.apply(alpha, xs_, beta, y_shard, what=2))
            tt = torch.tensor([t_step, t_x, t_l], device=dev, dtype=torch.float64)
            all_reduce_(tt, RED_MAX)
            xchg[name] = {"step_ms": round(tt[0].item(), 5), "exchange_only_ms": round(tt[1].item(), 5), "local_spmv_only_ms": round(tt[2].item(), 5),
                          "bytes_received_per_gpu": o.exchange_bytes, "aggregate_GFLOPs": round(2.0 * nnz_global / (tt[0].item() * 1e-3) / 1e9, 1)}
            if name == "allgather":          # the operator's own choice among its all-gather forms, and the times it was made on
                xchg[name]["form_chosen"] = ("collective", "send_receive", "p2p_pulls")[o.allgather_form] if o.allgather_form >= 0 else None
                xchg[name]["form_us_at_creation"] = o.allgather_us
        ops.clear()

    # ---- sanity inside the bench: A*1 over the slab must be the row-sum vector (0 interior, 1 boundary) ----
    chk = torch.empty(rows_per_rank, dtype=torch.float64, device=dev)
    if not use_dist:
        ones = torch.ones(nrows_global, dtype=torch.float64, device=dev)
        kk.spmv(handle, "N", 1.0, A, tb(ones), 0.0, tb(chk))
    else:
        x_keep = x_shard.clone(); x_shard.fill_(1.0)
        op.apply(1.0, x_shard, 0.0, chk)               # through the exchange: every rank's halo must arrive as ones
        x_shard.copy_(x_keep)
    lens = A.graph.row_map[1:] - A.graph.row_map[:-1]
    if emu:
        lens = torch.from_numpy(np.asarray(lens))
    assert bool((chk[lens == 27] == 0).all()) and bool((chk[lens < 27] == 1).all()), "bench self-check failed"

    # ---- roofline of the dominant kernel (local SpMV) -------------------------------------------
    if world == 1:
        x_touched = nrows_global
    else:
        lo = max(0, rank * planes_per_rank - 1); hi = min(nz, (rank + 1) * planes_per_rank + 1)
        x_touched = (hi - lo) * nx * ny
    alg_bytes = nnz_local * 12 + (rows_per_rank + 1) * 4 + x_touched * 8 + rows_per_rank * 8 + (rows_per_rank * 8 if beta != 0 else 0)
    achieved_events = alg_bytes / (kern_ms * 1e-3) / 1e9     # HIP events around the launches
    achieved = alg_bytes / (ms_step * 1e-3) / 1e9            # the step's own clock (per-call fence included; N > 1: the exchange too): what "frac" is quoted on
    gflops = 2.0 * nnz_global / (ms_step * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "SpMV GFLOP/s (CSR, 27-pt 3-D FE Laplacian, fp64)", "value": round(gflops, 2), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not emu else "synthetic; EMULATED ON CPU (control-flow check only, not a measurement)",
            "config": {"workload": workload, "rows": nrows_global, "nnz": nnz_global, "rows_per_gpu": rows_per_rank,
                       "alpha": alpha, "beta": beta, "offsets": "int32", "ordinals": "int32", "algorithm": args.algo,
                       "partition": ("1-D row slabs; x exchange = %s over RCCL/xGMI, %d bytes received per GPU per SpMV%s"
                                     % (op.exchange_mode, op.exchange_bytes,
                                        "; %d interior rows overlap the exchange" % op.interior_rows if op.query("parts") > 1 else ""))
                                    if use_dist else "single GPU",
                       "knobs": args.knob},
            "achieved_hbm_GBps_per_gpu": round(achieved_events, 1),
            "spmv_kernel_ms": round(kern_ms, 5),
            "protocol": ("per-call fence (KokkosSparse_kk_spmv.cpp:139-167): ms_per_step is the mean over the K calls incl. the fence" if not args.no_fence
                         else "K calls queued back to back between two fences"),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "kernel": "kk::spmv_stream3_kernel (+ fix-up kernel)",
                         "clock": "algorithmic bytes / ms_per_step (the timed loop's own clock); *_kernel_events: the same bytes / the mean HIP-event time around the launches",
                         "achieved_kernel_events": round(achieved_events, 1), "frac_kernel_events": round(achieved_events / HBM_PEAK_GBPS, 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        # What the plan really streams: the analysis replaces the 4-byte column indices by 16-bit codes or, on locally
        # Toeplitz tiles, by one small record per tile; "achieved" / "frac" stay on the CRS algorithmic bytes.
        try:
            h_ = handle if not use_dist else None
            if h_ is not None and h_.query("window_codes"):
                tile_ = h_.query("tile"); tiles_ = h_.query("tiles")
                pat_, code_, plain_ = h_.query("pattern_tiles"), h_.query("code_tiles"), h_.query("plain_tiles")
                # per tile: 4 B mode word; 256 B window meta unless plain; 2 B per nonzero of code tiles; 672 B record of pattern
                # tiles; plain tiles read their 4-byte columns
                streamed_ = (alg_bytes - nnz_local * 4 + tiles_ * 4 + (tiles_ - plain_) * 256 + code_ * tile_ * 2 + pat_ * 672
                             + plain_ * tile_ * 4)
                out["roofline"]["plan"] = {"tiles": tiles_, "pattern_record_tiles": pat_, "code_tiles": code_, "plain_tiles": plain_,
                                           "x_staged_in_lds": bool(h_.query("window_staged_x")), "tile_nnz": tile_,
                                           "plan_bytes": h_.query("plan_bytes"), "streamed_bytes_per_launch": streamed_}
        except Exception:
            pass
        # The device's own streaming-read rate next to the 8 TB/s spec ("re-measure on the build box and quote both"): the
        # matrix values (5.8 GB on C2) through kkamd_bench_read -- 16-byte loads, 256 lanes, 4 independent loads in flight per lane.
        if world == 1 and not emu:
            try:
                vals_ = A.values; nbytes_ = int(vals_.numel() * vals_.element_size()) // 16 * 16
                sink_ = torch.zeros(8, dtype=torch.float64, device=dev)
                lib_ = kk.torch_backend().lib
                st_ = torch.cuda.current_stream().cuda_stream
                run_ = lambda: kk._capi.check(lib_, lib_.kkamd_bench_read(vals_.data_ptr(), nbytes_, 4, 0, 0, sink_.data_ptr(), st_))
                for _ in range(3): run_()
                e0_, e1_ = Event(enable_timing=True), Event(enable_timing=True)
                e0_.record()
                for _ in range(10): run_()
                e1_.record(); torch.cuda.synchronize()
                rd_gbps = nbytes_ * 10 / (e0_.elapsed_time(e1_) * 1e-3) / 1e9
                out["roofline"]["measured_stream_read_GBps"] = round(rd_gbps, 1)
                out["roofline"]["frac_of_measured_stream_read"] = round(achieved_events / rd_gbps, 4)
            except Exception as e:
                out["roofline"]["measured_stream_read_GBps"] = None
                out["roofline"]["measured_stream_read_error"] = repr(e)[:120]
        # Memory-side traffic comes from rocprofv3 PMC passes of this same command (it cannot be counted live); the file is
        # stamped with the hash of the kernel sources it was measured on and is IGNORED when that differs from the sources here.
        out["roofline"]["kernel_source_sha"] = kernel_source_sha()
        if not use_dist and not args.n and not args.knob and os.path.exists(PMC_FILE):
            try:
                d = json.load(open(PMC_FILE))
                if d.get("kernel_source_sha") != out["roofline"]["kernel_source_sha"]:
                    out["roofline"]["traffic_source"] = ("%s is stale (measured on kernel sources %s): traffic not quoted"
                                                         % (os.path.relpath(PMC_FILE, ROOT), d.get("kernel_source_sha")))
                else:
                    rd = wr = 0.0                     # one SpMV = the per-mode launches of the stream kernel + the fix-up kernel
                    for k, v in d.get("counters", {}).items():
                        if "spmv_stream" in k and "FETCH_SIZE" in k: rd += v["mean_KB"] * 1024 * 2   # gfx950 x2 correction
                        if "spmv_stream" in k and "WRITE_SIZE" in k: wr += v["mean_KB"] * 1024
                    if rd and wr:
                        out["roofline"]["traffic"] = int(rd + wr)
                        # what actually crossed the memory side per second, against the same peak
                        out["roofline"]["moved_frac"] = round((rd + wr) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
                        if out["roofline"].get("measured_stream_read_GBps"):
                            out["roofline"]["moved_frac_of_measured_stream_read"] = round((rd + wr) / (kern_ms * 1e-3) / 1e9 / out["roofline"]["measured_stream_read_GBps"], 4)
                        out["roofline"]["traffic_source"] = ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                                                             "FETCH_SIZE x2 per MI355X_MICROARCH.md)" % os.path.relpath(PMC_FILE, ROOT))
            except Exception:
                pass
        out["first_call_ms"] = round(first_call_ms, 3)
        out["plan_create_ms"] = round(first_call_ms - ms_step, 3)      # what a solver that re-creates its handle pays again (the handle-less overloads do)
        if per_call:
            out["per_call_ms"] = {"mean": round(sum(per_call) / len(per_call), 5) if world == 1 else round(ms_step, 5),
                                  "min": round(pc_min, 5), "max": round(pc_max, 5)}
        if flushed is not None:
            out["per_call_ms_flushed"] = flushed
        if xchg is not None:
            out["exchange"] = xchg
        if emu:       # control-flow check: the same leg on the emulated grid (a fraction of a second)
            out["cpu_baseline"] = cpu_baseline(sample_n=n, min_seconds=0.2)
            out["cpu_baseline"]["sample"] = "EMULATED RUN, tiny grid: " + out["cpu_baseline"]["sample"]
        elif not args.no_cpu_baseline:     # every GPU count (north_star: the host path "in the same run"); rank 0, after the timed region
            out["cpu_baseline"] = cpu_baseline()
        # ---- the other SURVEY 8(d) metrics, driver-visible: config 3 (SpMV_MV) and config 4 (SpGEMM), each under its share of the cap
        if not use_dist and not emu and not args.n and args.extras_seconds > 0:
            t_x = time.perf_counter()
            del x_shard, y_shard, chk
            try:
                out["spmv_mv"] = bench_spmv_mv(kk, torch, A, 0.4 * args.extras_seconds, not args.no_cpu_baseline)
            except Exception as e:
                out["spmv_mv"] = {"error": repr(e)[:200]}
            del handle
            A = None
            torch.cuda.empty_cache()
            try:
                out["spmv_gather_bound"] = bench_rank1_gather_bound(kk, torch)
            except Exception as e:
                out["spmv_gather_bound"] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
            try:
                left = args.extras_seconds - (time.perf_counter() - t_x)
                out["spmv_mv_off_lattice"] = bench_spmv_mv_off_lattice(kk, torch, min(40.0, 0.3 * left)) if left > 60 else {"skipped": "time cap"}
            except Exception as e:
                out["spmv_mv_off_lattice"] = {"error": repr(e)[:200]}
            torch.cuda.empty_cache()
            try:
                left = args.extras_seconds - (time.perf_counter() - t_x)
                out["spgemm"] = bench_spgemm(kk, torch, max(20.0, left), not args.no_cpu_baseline) if left > 20 else {"skipped": "time cap"}
            except Exception as e:
                out["spgemm"] = {"error": repr(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
