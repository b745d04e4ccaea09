"""Two PROCESSES on the one GPU of the box, each a rank of the row-partitioned SpMV (kkamd_dist_spmv_*) with its own device
buffers: every exchange mode of the C implementation on real device memory between real processes -- the column-range halo,
the column-set halo (pack / scatter kernels, index lists agreed at creation), the all-gather through the collective, and the
peer-to-peer all-gather (hipIpc handles of the x buffers exchanged, concurrent pulls between two barrier collectives).  RCCL
refuses a communicator of two ranks on one device, so the collectives run through a gloo-backed kkamd_transport_t that stages
the (small) messages through the host; the p2p pulls move the shards device to device through the mapped buffers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        import kk_loader
        import oracle
        from test_dist_gloo import _scattered_matrix
        kk = kk_loader.load()
        from kokkos_kernels_amd.dist import DistSpmv, _GlooDeviceTransport
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        be = kk.torch_backend()
        out = {}
        lap = oracle.laplace3d("FE", 24, 20, 16)
        cases = [("laplace", lap, [0, 24 * 20 * 9, lap.nrows], ("auto", "halo", "allgather", "allgather_collective", "allgather_p2p", "halo_set")),
                 ("scattered", _scattered_matrix(30000), [0, 14000, 30000], ("auto", "halo_set", "allgather_p2p"))]
        for name, A0, offs, modes in cases:
            n = A0.nrows
            r0, r1 = offs[rank], offs[rank + 1]
            sl = slice(A0.row_map[r0], A0.row_map[r1])
            A = kk.CrsMatrix.from_host(r1 - r0, n, A0.row_map[r0:r1 + 1] - A0.row_map[r0], A0.entries[sl], A0.values[sl], backend=be)
            rng = np.random.default_rng(2)
            x = rng.random(n); y0 = rng.random(n)
            for mode in modes:
                tr = _GlooDeviceTransport(None, world)
                op = DistSpmv(A, offs, rank, exchange=mode, transport=tr)
                err = 0.0
                for scale, alpha, beta in ((1.0, 2.0, 0.5), (3.0, 1.0, 0.0), (-1.0, 1.0, 1.0)):       # x changes between the calls: a stale halo would show
                    xs = torch.from_numpy(scale * x[r0:r1]).cuda(); ys = torch.from_numpy(y0[r0:r1].copy()).cuda()
                    op.apply(alpha, xs, beta, ys)
                    torch.cuda.synchronize()
                    exp = oracle.spmv_serial("N", A0, alpha, scale * x, beta, y0.copy())[r0:r1]
                    err = max(err, float(np.abs(ys.cpu().numpy() - exp).max()))
                out[(name, mode)] = (err, op.exchange_mode, op.exchange_bytes)
                dist.barrier()
                del op
        ret[rank] = out
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                       # surface the failure in the parent instead of a bare spawn error
        import traceback
        ret[rank] = "FAILED: " + repr(e) + "\n" + traceback.format_exc()
        raise


def test_exchange_modes_between_two_processes_on_one_gpu():
    import torch.multiprocessing as mp
    world = 2
    port = 31500 + (os.getpid() % 1500)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert len(ret) == world
        for r in range(world):
            assert not isinstance(ret[r], str), ret[r]
            got = ret[r]
            plane = 24 * 20
            for (name, mode), (err, chosen, nbytes) in got.items():
                assert err <= 1e-12, (r, name, mode, err)
                if mode == "allgather": assert chosen in ("allgather", "allgather_p2p"), (name, chosen)        # the operator's own choice among its forms
                elif mode == "allgather_collective": assert chosen == "allgather", (name, chosen)
                elif mode != "auto": assert chosen == mode, (name, mode, chosen)
            assert got[("laplace", "auto")][1] == "halo" and got[("laplace", "auto")][2] == plane * 8       # one grid plane from the neighbour
            assert got[("laplace", "halo_set")][2] == plane * 8                                              # the same plane, as a column set
            assert got[("scattered", "auto")][1] == "halo_set" and got[("scattered", "auto")][2] < 0.25 * 8 * 30000
            other = 30000 - (14000 if r == 0 else 16000)
            assert got[("scattered", "allgather_p2p")][2] == other * 8
