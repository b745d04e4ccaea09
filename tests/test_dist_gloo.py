"""world_size-2 test of the multi-GPU path on CPU: the C implementation (csrc/kk_dist.hip, built under the SIMT emulator of
tests/emu) with a kkamd_transport_t whose two callbacks run gloo collectives (dist._GlooTransport) in place of RCCL; the
local SpMV runs the real kernel sources.  Checks the 1-D row partition (local row_map, global columns), equal and ragged
slabs, the halo lists, the interior / boundary split and the zero-copy x window."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ragged, exchange, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import kk_loader
    import oracle
    from emu import emu_backend
    kk = kk_loader.load()
    from kokkos_kernels_amd.dist import DistSpmv, slab_offsets
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = emu_backend.backend()
    # the slab plans (whole slab, interior and boundary views) take the 16-bit column codes even at this size
    kk._capi.check(be.lib, be.lib.kkamd_set_default(b"window_codes_min_knnz", 0))
    nx, ny, nz = 9, 8, 7
    A0 = oracle.laplace3d("FE", nx, ny, nz)
    n = A0.nrows
    offs = slab_offsets(n, world, align=1 if ragged else nx * ny)
    if ragged:
        offs = [0, n // 3 + 5, n][: world + 1]
    r0, r1 = offs[rank], offs[rank + 1]
    # local slab: rows [r0, r1), row_map rebased to 0, global column indices kept
    rm = A0.row_map[r0:r1 + 1] - A0.row_map[r0]
    sl = slice(A0.row_map[r0], A0.row_map[r1])
    A = kk.CrsMatrix.from_host(r1 - r0, n, rm, A0.entries[sl], A0.values[sl], backend=be)
    # the device slab generator must produce exactly this slab
    G = kk.laplace_matrix("FE", nx, ny, nz, backend=be, rows=(r0, r1 - r0))
    g_rm, g_ent, g_val = G.to_host()
    ok_gen = np.array_equal(g_rm, rm) and np.array_equal(g_ent, A0.entries[sl]) and np.array_equal(g_val, A0.values[sl])
    rng = np.random.default_rng(0)
    x = rng.random(n); y0 = rng.random(n)
    op = DistSpmv(A, offs, rank, to_backend=lambda t: t.numpy(), exchange=exchange)
    xs = torch.from_numpy(x[r0:r1].copy()); ys = torch.from_numpy(y0[r0:r1].copy())
    op.apply(2.0, xs, 0.5, ys)
    op.apply(1.0, xs, 0.0, ys.clone())        # second call reuses the plans and the x buffer
    exp = oracle.spmv_serial("N", A0, 2.0, x, 0.5, y0.copy())[r0:r1]
    err = float(np.abs(ys.numpy() - exp).max())
    # x kept in the operator's own window (no copy), NaN-seeded y overwritten with beta = 0
    xl = op.x_local(); xl.copy_(torch.from_numpy(3.0 * x[r0:r1]))
    y2 = torch.full((r1 - r0,), float("nan"), dtype=torch.float64)
    op.apply(1.0, xl, 0.0, y2)
    err = max(err, float(np.abs(y2.numpy() - oracle.spmv_serial("N", A0, 3.0, x, 0.0, np.zeros(n))[r0:r1]).max()))
    tol = oracle.spmv_max_error(A0, 3.0, 0.5, max_val=32.0)
    ret[rank] = (ok_gen, err, tol, op.exchange_mode, op.exchange_bytes, op.interior_rows, op.query("parts"), op.allgather_form, op.allgather_us, all(offs[i + 1] - offs[i] == offs[1] - offs[0] for i in range(world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["auto", "allgather", "allgather_collective"])
@pytest.mark.parametrize("ragged", [False, True])
def test_row_partitioned_spmv_world2(ragged, exchange):
    import torch.multiprocessing as mp
    world = 2
    port = 29500 + (os.getpid() % 2000) + (7 if ragged else 0) + (13 if exchange == "auto" else (29 if exchange == "allgather" else 0))
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ragged, exchange, ret), nprocs=world, join=True)
        assert len(ret) == world
        for r in range(world):
            ok_gen, err, tol, mode, nbytes, interior, parts, ag_form, ag_us, equal = ret[r]
            assert ok_gen, "slab generator mismatch on rank %d" % r
            assert err <= tol, "rank %d: %g > %g" % (r, err, tol)
            if exchange == "auto" and not ragged:
                # 9x8x7 grid, slabs of 4 and 3 planes: each rank needs one 72-node plane of its neighbour
                assert mode == "halo" and nbytes == 72 * 8, (mode, nbytes)
                # interior = every plane that does not touch the neighbour's plane, computed while the halo is in flight
                planes = 4 if r == 0 else 3
                # (boundaries may move a few rows inward so the views start on 16-byte aligned entries)
                assert parts == 2 and (planes - 1) * 72 - 16 <= interior <= (planes - 1) * 72, (parts, interior)
            if exchange == "allgather":
                # the operator timed the forms it has (no mapped memory between processes under the emulator) and every rank kept the same one:
                # equal shards -> the collective or the point-to-point form; unequal shards (both partitions of this test) have no collective form
                assert mode == "allgather" and ag_us is not None and ag_us["p2p"] == -1 and ag_us["sendrecv"] >= 0, (mode, ag_us)
                assert (ag_us["collective"] == -1) == (not equal) and ag_form in ((0, 1) if equal else (1,)), (equal, ag_form, ag_us)
                assert ag_form == ret[0][7] and ag_us == ret[0][8]
            if exchange == "allgather_collective":
                assert mode == "allgather" and ag_us is None and ag_form == (0 if equal else 1), (mode, ag_form, ag_us)


def _scattered_matrix(n, seed=5):
    """a band of +-2 plus, in every tenth row, two far columns anywhere: the column RANGE of a slab is everything, the column SET small"""
    import oracle
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for r in range(n):
        c = {max(0, r - 2), max(0, r - 1), r, min(n - 1, r + 1), min(n - 1, r + 2)}
        if r % 10 == 3: c |= {int(v) for v in rng.integers(0, n, size=2)}
        for v in sorted(c): rows.append(r); cols.append(v)
    rows = np.array(rows); cols = np.array(cols, dtype=np.int32)
    rm = np.zeros(n + 1, dtype=np.int64); np.add.at(rm, rows + 1, 1); rm = np.cumsum(rm)
    return oracle.Crs(n, n, rm, cols, 0.5 + rng.random(cols.size))


def _worker_set(rank, world, port, exchange, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import kk_loader
    import oracle
    from emu import emu_backend
    kk = kk_loader.load()
    from kokkos_kernels_amd.dist import DistSpmv
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = emu_backend.backend()
    n = 700
    A0 = _scattered_matrix(n)
    offs = [0, 230, 470, n] if world == 3 else [0, 330, n]
    r0, r1 = offs[rank], offs[rank + 1]
    sl = slice(A0.row_map[r0], A0.row_map[r1])
    A = kk.CrsMatrix.from_host(r1 - r0, n, A0.row_map[r0:r1 + 1] - A0.row_map[r0], A0.entries[sl], A0.values[sl], backend=be)
    rng = np.random.default_rng(1)
    x = rng.random(n); y0 = rng.random(n)
    op = DistSpmv(A, offs, rank, to_backend=lambda t: t.numpy(), exchange=exchange)
    err = 0.0
    for scale, alpha, beta in ((1.0, 2.0, 0.5), (3.0, 1.0, 0.0), (-1.0, 1.0, 1.0)):      # x changes between the calls: stale halo entries would show
        xs = torch.from_numpy(scale * x[r0:r1]); ys = torch.from_numpy(y0[r0:r1].copy())
        op.apply(alpha, xs, beta, ys)
        exp = oracle.spmv_serial("N", A0, alpha, scale * x, beta, y0.copy())[r0:r1]
        err = max(err, float(np.abs(ys.numpy() - exp).max()))
    # what the importer must move: the distinct off-slab columns of the slab
    want = np.unique(A0.entries[sl]); want = want[(want < r0) | (want >= r1)]
    ret[rank] = (err, oracle.spmv_max_error(A0, 3.0, 1.0, max_val=2.0), op.exchange_mode, op.exchange_bytes, int(want.size) * 8, op.query("recvs"), op.query("parts"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("exchange", ["auto", "halo_set"])
def test_column_set_halo(world, exchange):
    """the general importer.  A slab whose columns are scattered gets the entries of x its SET of off-slab columns
    names (per-peer index lists agreed once, pack kernel, point-to-point pieces, scatter kernel) -- auto picks it because the
    column range would be the whole vector; results against the serial oracle over three calls with changing x"""
    import torch.multiprocessing as mp
    port = 30500 + (os.getpid() % 1500) + 31 * world + (5 if exchange == "auto" else 0)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_set, args=(world, port, exchange, ret), nprocs=world, join=True)
        assert len(ret) == world
        for r in range(world):
            err, tol, mode, nbytes, want_bytes, recvs, parts = ret[r]
            assert err <= tol, "rank %d: %g > %g" % (r, err, tol)
            assert mode == "halo_set", (r, mode)
            assert nbytes == want_bytes and nbytes < 0.25 * 8 * 700, (r, nbytes, want_bytes)
            assert 1 <= recvs <= world - 1


def _check_spgemm_slabs(be, kk, oracle, R, world):
    """the slabs of C computed independently through the C ABI (kkamd_dist_spgemm_*) concatenate to the full product; the
    partition balances multiplications"""
    from kokkos_kernels_amd.dist import DistSpgemm, work_balanced_offsets
    lenB = np.diff(R.row_map)
    flops = np.array([lenB[R.entries[R.row_map[i]:R.row_map[i + 1]]].sum() for i in range(R.nrows)])
    B = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, backend=be)
    offs, mults = DistSpgemm.partition(B, B, world)
    assert offs[0] == 0 and offs[-1] == R.nrows and all(a <= b for a, b in zip(offs, offs[1:]))
    assert mults == [int(flops[a:b].sum()) for a, b in zip(offs, offs[1:])] and sum(mults) == int(flops.sum())
    assert max(mults) <= flops.sum() / world + flops.max()
    ref = work_balanced_offsets(flops, world)                  # the host helper cuts at the same places, give or take a row
    assert all(abs(a - b) <= 1 for a, b in zip(offs, ref)), (offs, ref)
    gold = oracle.spgemm(R, R)
    rm_all, ent_all, val_all = [0], [], []
    for rank, (a, b) in enumerate(zip(offs, offs[1:])):
        sl = slice(R.row_map[a], R.row_map[b])
        A_slab = kk.CrsMatrix.from_host(b - a, R.ncols, R.row_map[a:b + 1] - R.row_map[a], R.entries[sl], R.values[sl], backend=be)
        op = DistSpgemm(offs, rank, be)
        Cs = op.symbolic(A_slab, B)
        assert op.query("rows_local") == b - a and op.query("row0") == a and op.query("rows_global") == R.nrows
        assert op.query("c_nnz_local") == Cs.nnz() and op.query("mults_local") == mults[rank]
        op.numeric(A_slab, B, Cs)
        op.numeric(A_slab, B, Cs)                              # numeric reuse on the slab
        r, e, v = Cs.to_host()
        rm_all += list(np.asarray(r[1:], dtype=np.int64) + rm_all[-1]); ent_all.append(e); val_all.append(v)
        if b - a > 0:
            with pytest.raises(kk.KkamdError):                 # a slab of the wrong height is refused
                wrong = kk.CrsMatrix.from_host(b - a - 1, R.ncols, (R.row_map[a:b] - R.row_map[a]), R.entries[R.row_map[a]:R.row_map[b - 1]],
                                               R.values[R.row_map[a]:R.row_map[b - 1]], backend=be)
                op.symbolic(wrong, B)
    got = oracle.Crs(R.nrows, R.ncols, np.array(rm_all), np.concatenate(ent_all), np.concatenate(val_all))
    ok, msg = oracle.is_same_matrix(got, gold)
    assert ok, msg


def test_row_partitioned_spgemm_slabs():
    """SpGEMM shards by rows of A with B replicated and no exchange at all (kkamd_dist_spgemm_*, here under the emulator, one rank
    after the other)"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import kk_loader
    import oracle
    from emu import emu_backend
    kk = kk_loader.load()
    _check_spgemm_slabs(emu_backend.backend(), kk, oracle, oracle.rmat(9, 8), 4)


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
def test_bench_two_process_flow_emulated(launcher):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank) and started plainly
    (`python bench.py --gpus 2`: no WORLD_SIZE in the environment, bench.py launches its own ranks), with
    --emulate: gloo instead of RCCL and the kernels under the SIMT emulator.  Checks the whole multi-process control
    flow -- slab generation, halo plan, interior/boundary overlap, barriers, max-over-ranks timing, the A*1 self-check,
    the single JSON line from rank 0 -- without a GPU."""
    import json
    import subprocess
    port = 29900 + (os.getpid() % 500)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--emulate", "--grid-edge", "32"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["unit"] == "GFLOP/s"
    assert d["config"]["rows"] == 32 * 32 * 8 and d["config"]["rows_per_gpu"] == 32 * 32 * 4
    assert "halo" in d["config"]["partition"] and "overlap" in d["config"]["partition"], d["config"]["partition"]
    assert "%d bytes" % (32 * 32 * 8) in d["config"]["partition"]            # one 32x32 plane of doubles from the neighbour
    # the N > 1 line is as complete as the N = 1 line: roofline (quoted on the step's own clock, the event-clock figure beside it),
    # the host baseline of the same run, every exchange the library has, the partition
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "traffic" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["frac_kernel_events"] >= 0      # (the emulator runs at MB/s: both round to 0)
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / d["ms_per_step"] / 1e6) <= 0.06 + 1e-3 * rf["achieved"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "GFLOP/s" and cb["value"] > 0 and cb["cores"] >= 1 and "27-pt FE Laplacian" in cb["sample"]
    assert set(d["exchange"]) >= {"halo", "halo_set", "allgather"}
    for x in ("halo", "halo_set", "allgather"):
        assert d["exchange"][x]["step_ms"] > 0 and d["exchange"][x]["bytes_received_per_gpu"] > 0
    # the all-gather operator chose its form itself from what it timed at its creation (equal shards: the collective and the point-to-point form; no
    # mapped memory between processes under the emulator)
    ag = d["exchange"]["allgather"]
    assert ag["form_chosen"] in ("collective", "send_receive") and ag["form_us_at_creation"]["collective"] >= 0 and ag["form_us_at_creation"]["sendrecv"] >= 0 \
        and ag["form_us_at_creation"]["p2p"] == -1, ag


def test_slab_offsets():
    sys.path.insert(0, ROOT)
    import kk_loader
    kk_loader.load()
    from kokkos_kernels_amd.dist import slab_offsets
    assert slab_offsets(600 * 600 * 600, 8, align=600 * 600) == [i * 27_000_000 for i in range(9)]
    o = slab_offsets(103, 4)
    assert o[0] == 0 and o[-1] == 103 and all(b > a for a, b in zip(o, o[1:]))
