"""Randomised parity sweep against the CPU oracle: shapes are drawn to hit every SpGEMM bin / window / hub path, unsorted and
duplicate-laden inputs, both offset types, the SpMV kernels on irregular rows, the plan modes and the rank-2 kernels.
`run(be, budget_s, seed0)` is used by tests/test_gpu_parity.py::test_fuzz_slice (a 60-second slice the driver runs) and by
tools/fuzz_gpu.py (long runs by hand through gpurun)."""
import time

import numpy as np

import oracle
import parity_cases as pc

kk = pc.kk


def hubby(rng, n, ncols, base, nhubs, hublen, sort=True):
    lens = rng.integers(0, 2 * base + 1, size=n)
    for h in rng.choice(n, size=min(nhubs, n), replace=False):
        lens[h] = min(ncols, int(hublen * (0.5 + rng.random())))
    rm = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=rm[1:])
    ent = np.empty(rm[-1], dtype=np.int32)
    for i in range(n):
        c = rng.choice(ncols, size=lens[i], replace=False)
        ent[rm[i]:rm[i + 1]] = np.sort(c) if sort else c
    return oracle.Crs(n, ncols, rm, ent, 1 + 49 * rng.random(rm[-1]))


def run(be, budget, seed0=0, max_cases=None, kinds=None):
    """runs random cases until `budget` seconds are over (or max_cases); returns (cases passed, per-kind counts, last seed).
    kinds: only these kinds (indices into KINDS) are run, the others' seeds are skipped"""
    t_end = time.time() + budget
    n_ok = 0; case = 0; per_kind = [0] * 8
    while time.time() < t_end and (max_cases is None or case < max_cases):
        rng = np.random.default_rng(seed0 + case); kind = case % 8; case += 1
        if kinds is not None and kind not in kinds: continue
        odt = np.int64 if rng.random() < 0.5 else np.int32
        vdt = np.float32 if rng.random() < 0.3 else np.float64
        if kind in (0, 1, 2, 6):    # SpGEMM kinds: the column-block class under random block widths, item capacities and forms (round 5)
            spgemm_knobs = {b"spgemm_block_w": int(rng.choice([64, 256, 1024, 16384, 16384])), b"spgemm_item_cap": int(rng.choice([64, 300, 6144, 6144])),
                            b"spgemm_items": int(rng.choice([0, 1, 1])), b"spgemm_block_min_pct": int(rng.choice([1, 4, 20])), b"spgemm_unit_bits": int(rng.choice([8, 11, 14, 18, 18]))}
            for k_, v_ in spgemm_knobs.items(): kk._capi.check(be.lib, be.lib.kkamd_set_default(k_, v_))
        try:
            if kind == 0:      # SpGEMM, skewed: few long rows of A against hub rows of B
                n = int(rng.integers(50, 400)); k = int(rng.integers(3000, 60000))
                if rng.random() < 0.4:     # rows of C with a few hundred to a few thousand entries out of about as many products: wave table, LDS sort, bitmap
                    k = int(rng.integers(20000, 200000))
                    B = hubby(rng, n, k, int(rng.integers(5, 40)), 0, 1)
                    A = hubby(rng, int(rng.integers(5, 80)), n, min(int(rng.integers(8, 50)), n // 2 - 1), int(rng.integers(0, 3)), min(n, int(rng.integers(60, 300))))
                else:
                    B = hubby(rng, n, k, int(rng.integers(2, 30)), int(rng.integers(1, 6)), int(rng.integers(500, 20000)))
                    A = hubby(rng, int(rng.integers(3, 60)), n, int(rng.integers(1, 8)), int(rng.integers(0, 3)), n)
                pc.check_spgemm(be, A, B, offset_dtype=odt, value_dtype=vdt)
            elif kind == 1:    # SpGEMM, R-MAT square
                s = int(rng.integers(8, 14)); R = oracle.rmat(s, int(rng.integers(4, 24)), seed=int(rng.integers(1, 1 << 30)))
                pc.check_spgemm(be, R, R, offset_dtype=odt, value_dtype=vdt, algo=("SPGEMM_KK_DENSE" if s <= 10 and rng.random() < 0.3 else "SPGEMM_KK"),
                                options={"compression": int(rng.integers(0, 3))})
            elif kind == 2:    # SpGEMM, unsorted inputs with duplicates (B unsorted -> HBM accumulators for dense rows)
                A = pc.randomized(oracle.random_crs(int(rng.integers(20, 200)), 150, int(rng.integers(2, 40)), seed=int(rng.integers(1, 1 << 30))))
                B = hubby(rng, 150, int(rng.integers(2000, 30000)), int(rng.integers(2, 20)), 3, 6000, sort=False)
                pc.check_spgemm(be, A, B, offset_dtype=odt, value_dtype=vdt)
            elif kind == 3:    # SpMV on irregular rows, every algorithm / mode
                M = hubby(rng, int(rng.integers(100, 5000)), int(rng.integers(100, 5000)), int(rng.integers(1, 40)), int(rng.integers(0, 4)), 3000, sort=bool(rng.integers(0, 2)))
                for algo in (None, "SPMV_DEFAULT", "SPMV_MERGE_PATH"):
                    for mode in "NT":
                        # analysed handles also through the column-slab copy (forced: slabs of 2^4 .. 2^12 columns) and the fingerprint-refreshed transpose
                        kn = {"colslab": int(rng.choice([2, 4, 4])), "colslab_shift": int(rng.integers(4, 13)), "colslab_const": int(rng.integers(0, 2)), "explicit_transpose_min_knnz": 0} if algo and rng.random() < 0.5 else None
                        pc.check_spmv(be, M, mode, float(rng.integers(-3, 4)), float(rng.integers(-2, 3)), algo=algo, offset_dtype=odt, max_val=50.0, seed=case, knobs=kn)
                # rank 2 on the same matrix: the nonzero-split kernel (forced, or chosen by the long-row share), the gather kernels, and modes T / H
                # through the cached transpose under a random value-tracking policy
                for mode in "NT":
                    beta = float(rng.integers(-1, 2))
                    pc.check_spmv_mv(be, M, int(rng.choice([2, 5, 16, 16, 21, 32])), mode, float(rng.integers(-3, 4)), beta, str(rng.choice(["C", "F"])), str(rng.choice(["C", "F"])),
                                     algo="SPMV_DEFAULT", seed=case, max_val=50.0, nans=(beta == 0.0 and rng.random() < 0.5), offset_dtype=odt,
                                     knobs={"mv6": int(rng.choice([0, 1, 2, 2])), "explicit_transpose_min_knnz": 0, "explicit_transpose": int(rng.choice([0, 1, 1])),
                                            "values_tracking": int(rng.integers(0, 3))})
            elif kind == 4:    # sort / merge / transpose
                M = hubby(rng, int(rng.integers(5, 300)), int(rng.integers(50, 40000)), int(rng.integers(1, 30)), int(rng.integers(0, 3)), int(rng.integers(9000, 40000)), sort=False)
                M.entries[rng.integers(0, M.nnz, size=M.nnz // 7)] = M.entries[rng.integers(0, M.nnz, size=M.nnz // 7)]   # duplicates
                Cm = kk.sort_and_merge_matrix(pc.dev(be, M, odt))
                gm = oracle.sort_and_merge(oracle.Crs(M.nrows, M.ncols, M.row_map, M.entries.copy(), M.values.copy()))
                r, e, v = Cm.to_host()
                assert np.array_equal(r, gm.row_map) and np.array_equal(e, gm.entries) and np.allclose(v, gm.values, rtol=1e-12)
                Tt = kk.transpose_matrix(Cm); gt = oracle.transpose(gm)
                r, e, v = Tt.to_host()
                assert np.array_equal(r, gt.row_map) and np.array_equal(e, gt.entries) and np.allclose(v, gt.values, rtol=1e-12)
            elif kind == 6:    # SpGEMM, long rows of A: LDS hub kernel (512 < entries <= 4096) and the HBM-accumulator path beyond
                n = int(rng.integers(700, 7000)); k = int(rng.integers(5000, 50000))
                B = hubby(rng, n, k, int(rng.integers(2, 12)), int(rng.integers(0, 4)), int(rng.integers(300, 3000)))
                A = hubby(rng, int(rng.integers(2, 12)), n, 3, int(rng.integers(1, 4)), int(rng.integers(600, min(n, 6500))))
                pc.check_spgemm(be, A, B, offset_dtype=odt, value_dtype=vdt)
            elif kind == 7:    # SpMV through an analysed handle with the 16-bit window codes forced on: structured, clustered, banded, random
                sub = int(rng.integers(0, 5))
                if sub == 0:       # several diagonals at random offsets
                    n = int(rng.integers(500, 30000)); nc = int(rng.integers(n, 40 * n)); step = max(1, nc // n)
                    offs = np.unique(rng.integers(0, nc - step * n + 1, size=int(rng.integers(2, 30))))
                    cols = np.arange(n)[:, None] * step + offs[None, :]
                    rm = np.arange(n + 1, dtype=np.int64) * len(offs)
                    M = oracle.Crs(n, nc, rm, cols.reshape(-1).astype(np.int32), 1 + rng.random(n * len(offs)))
                elif sub == 1:     # k column clusters per row
                    n = int(rng.integers(300, 8000)); k = int(rng.integers(1, 22)); width = int(rng.integers(1, 5000)); gap = int(rng.integers(width, 30000))
                    cols = np.sort((np.arange(k)[None, :] * gap + rng.integers(0, width, (n, k))).astype(np.int32), axis=1)
                    M = oracle.Crs(n, gap * k + 1, np.arange(n + 1, dtype=np.int64) * k, cols.reshape(-1), 1 + rng.random(n * k))
                elif sub == 2:     # stencils
                    nd = int(rng.integers(2, 4)); dims = tuple(int(rng.integers(3, 400 if nd == 2 else 60)) for _ in range(nd))
                    if rng.random() < 0.5: dims = (int(rng.integers(150, 700)),) + tuple(int(rng.integers(3, 12)) for _ in range(nd - 1))   # long grid lines
                    st = "FE" if rng.random() < 0.5 else "FD"
                    M = oracle.laplace2d(st, *dims) if nd == 2 else oracle.laplace3d(st, *dims)
                elif sub == 4:     # block structure: a stencil with 2 .. 4 degrees of freedom per node, or dense diagonal blocks (the matrix-core rank-2 kernel)
                    if rng.random() < 0.5:
                        M = pc.multi_dof(oracle.laplace2d("FE" if rng.random() < 0.5 else "FD", int(rng.integers(3, 40)), int(rng.integers(3, 30))), int(rng.integers(2, 5)), seed=case)
                    else:
                        bs = int(rng.integers(2, 70)); M = pc._crop_rows(pc.block_diagonal(int(rng.integers(2, 60)), bs, seed=case), int(rng.integers(bs, 2 * bs + 1)) if rng.random() < 0.2 else bs * 2)
                else:              # banded random with duplicates / unsorted rows
                    n = int(rng.integers(200, 20000))
                    M = oracle.random_crs(n, n + int(rng.integers(0, 50)), int(rng.integers(1, 40)), variance=int(rng.integers(0, 10)), seed=int(rng.integers(1, 1 << 30)),
                                          bandwidth=int(rng.integers(5, 3000)))
                knobs = {"window_codes_min_knnz": 0, "window_codes": int(rng.integers(1, 3)), "nnz_per_thread": int(rng.choice([0, 4, 8, 16])),
                         "xcd_remap": int(rng.choice([0, 1, 2, 16])), "pattern_codes": int(rng.choice([0, 1, 2, 2])), "pattern_codes_min_knnz": 0,
                         "window_codes_min_pct": int(rng.choice([0, 10, 25, 60]))}
                if rng.random() < 0.5:     # rank 2 on the same matrix: plane-marching (where it applies), LDS-staged tiles, wave-private kernel, every tile order
                    pc.check_spmv_mv(be, M, int(rng.choice([8, 16, 16, 32, 24, 5])), "N", float(rng.integers(-3, 4)), float(rng.integers(-1, 2)), str(rng.choice(["C", "F"])), str(rng.choice(["C", "F"])),
                                     algo="SPMV_DEFAULT", seed=case, max_val=50.0, nans=bool(rng.random() < 0.3), offset_dtype=odt,
                                     knobs={"mv_kernel": int(rng.choice([0, 0, 2, 3, 5])), "mv_order": int(rng.integers(0, 3)), "mv_strip_min_kb": 50, "mv_strip_l2_kb": int(rng.choice([64, 512])),
                                            "mv4_wg_per_cu": int(rng.choice([1, 8, 64])), "mv5": int(rng.choice([1, 1, 2, 0])), "mv5_min_fill_pct": int(rng.choice([25, 5, 60]))})
                for beta in (0.0, float(rng.integers(-2, 3))):
                    pc.check_spmv(be, M, "N", float(rng.integers(-3, 4)), beta, algo="SPMV_DEFAULT", offset_dtype=odt, max_val=50.0, seed=case, knobs=knobs,
                                  nans=(beta == 0.0), value_dtype=(vdt if vdt == np.float32 and rng.random() < 0.5 else None))
            else:              # spmv_struct, random grids
                nd = int(rng.integers(1, 4)); dims = tuple(int(rng.integers(3, 300 if nd == 1 else (150 if nd == 2 else 40))) for _ in range(nd))
                pc.check_spmv_struct(be, dims, 1 if nd == 1 else int(rng.integers(1, 3)), offset_dtype=odt, seed=case)
            n_ok += 1; per_kind[kind] += 1
        except Exception as ex:
            print("FAILED case %d (seed %d, kind %d): %r" % (case - 1, seed0 + case - 1, kind, ex), flush=True)
            raise
        finally:
            if kind in (0, 1, 2, 6):
                for k_, v_ in ((b"spgemm_block_w", 16384), (b"spgemm_item_cap", 6144), (b"spgemm_items", 1), (b"spgemm_block_min_pct", 4), (b"spgemm_unit_bits", 18)):
                    kk._capi.check(be.lib, be.lib.kkamd_set_default(k_, v_))
    return n_ok, per_kind, seed0 + case - 1


KINDS = ("spgemm skewed", "spgemm rmat + options", "spgemm unsorted", "spmv irregular", "sort/merge/transpose", "spmv_struct", "spgemm long rows",
         "spmv plan modes + rank 2")
