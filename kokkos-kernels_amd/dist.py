"""1-D row-partitioned SpMV across the GPUs of one node -- a thin caller of the C ABI (kkamd_dist_spmv_*, include/kkamd.h,
csrc/kk_dist.hip).  Rank r owns the contiguous row slab [offsets[r], offsets[r+1]) of A (local row_map, GLOBAL column
indices), the matching slab of y and shard of x; the library exchanges the x entries a slab references (halo: grouped
ncclSend / ncclRecv of the column range each slab touches; halo_set: of the column SET it touches, packed / scattered by two
small kernels -- the general importer for slabs with scattered columns; all-gather: every shard to every rank, through the
collective or, allgather_p2p, by world - 1 concurrent peer-to-peer pulls between two barriers), overlaps the interior
rows with the exchange, and runs the planned local SpMV.  What happens here:

  * GPU (torch backend, process group "nccl"): the library's built-in RCCL transport; rank 0 obtains the 128-byte RCCL id from
    the library and torch.distributed broadcasts it -- the only thing torch.distributed does on the data path's behalf;
  * CPU tests (emulator backend, process group "gloo"): the same library code with a kkamd_transport_t whose two callbacks run
    the gloo collectives -- so the world-2 tests exercise the C implementation's exchange lists, overlap split and sequencing.

SpGEMM shards by rows of A with B replicated: no data-path communication (spgemm_row_slab)."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check
from .sparse import _ALGOS, _scalar_type

# "allgather": the operator times the forms it has at its creation -- the transport's collective, every shard to every peer point to point, peer-to-peer
# pulls of mapped memory -- and keeps the fastest; "allgather_collective" / "allgather_p2p" force one form (measurement)
_EXCHANGE = {"auto": 0, "halo": 1, "allgather_collective": 2, "allgather_p2p": 3, "halo_set": 4, "allgather": 5}


def slab_offsets(nrows, world, align=1):
    """contiguous, near-equal row slabs; `align` keeps slab boundaries on multiples (e.g. a grid plane)."""
    units = nrows // align
    base, rem = divmod(units, world)
    offs = [0]
    for r in range(world):
        offs.append(offs[-1] + (base + (1 if r < rem else 0)) * align)
    offs[-1] = nrows
    return offs


def work_balanced_offsets(work_per_row, world):
    """contiguous row slabs of near-equal total work (e.g. SpGEMM multiplications per row of A)"""
    w = np.asarray(work_per_row, dtype=np.float64)
    cs = np.concatenate([[0.0], np.cumsum(w)])
    targets = cs[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(cs, targets, side="left")
    offs = [0] + [int(c) for c in cuts] + [len(w)]
    for i in range(1, len(offs)):
        offs[i] = max(offs[i], offs[i - 1])
    return offs


class DistSpgemm:
    """Row-partitioned SpGEMM behind the C ABI (kkamd_dist_spgemm_*): rank r owns a contiguous row slab of A (local row_map) and
    the matching row slab of C = A * B; B is replicated, so there is NO data-path communication (partition where the
    path shards).  With slabs balanced by multiplications (partition) this is what makes BASELINE config 4 as specified (R-MAT
    scale 22: nnz(C) = 7.2e10 = 863 GB) fit: ~108 GB of C per GPU on eight GPUs."""

    @staticmethod
    def partition(A, B, world):
        """contiguous row slabs of A of near-equal multiplications, from the full A and B on this device: (offsets, mults per rank)"""
        be, lib = A.backend, A.backend.lib
        offs = (C.c_int64 * (world + 1))(); mults = (C.c_int64 * world)()
        da, db = A.desc(), B.desc()
        check(lib, lib.kkamd_dist_spgemm_partition(A.numRows(), da.d_row_map, da.d_entries, db.d_row_map, da.offset_type, world, offs, mults, be.stream()))
        return [int(v) for v in offs], [int(v) for v in mults]

    def __init__(self, offsets, rank, backend):
        self.be, self.lib = backend, backend.lib
        self.offsets, self.rank, self.world = [int(o) for o in offsets], int(rank), len(offsets) - 1
        op = C.c_void_p()
        check(self.lib, self.lib.kkamd_dist_spgemm_create(C.byref(op), self.world, self.rank, (C.c_int64 * (self.world + 1))(*self.offsets)))
        self._op = op

    def set(self, key, value):
        check(self.lib, self.lib.kkamd_spgemm_set(C.c_void_p(self.lib.kkamd_dist_spgemm_handle(self._op)), key.encode(), float(value)))

    def query(self, key):
        v = C.c_int64(0)
        check(self.lib, self.lib.kkamd_dist_spgemm_query(self._op, key.encode(), C.byref(v)))
        return int(v.value)

    def symbolic(self, A_slab, B):
        """row_map of the rank's slab of C; entries / values are allocated from the slab's nnz"""
        from .sparse import CrsMatrix, _np_dtype
        be = self.be
        da, db = A_slab.desc(), B.desc()
        m, n, k = A_slab.numRows(), A_slab.numCols(), B.numCols()
        if n != B.numRows():
            raise RuntimeError("DistSpgemm: A has %d columns, B %d rows" % (n, B.numRows()))
        odt = _np_dtype(A_slab.graph.row_map)
        rmC = be.empty(m + 1, odt)
        nnz = C.c_int64(0)
        check(self.lib, self.lib.kkamd_dist_spgemm_symbolic(self._op, m, n, k, da.d_row_map, da.d_entries, db.d_row_map, db.d_entries, be.ptr(rmC),
                                                            da.offset_type, C.byref(nnz), be.stream()))
        vdt = _np_dtype(A_slab.values)
        return CrsMatrix(m, k, rmC, be.empty(max(nnz.value, 1), np.int32)[:nnz.value], be.empty(max(nnz.value, 1), vdt)[:nnz.value], backend=be)

    def numeric(self, A_slab, B, C_slab):
        be = self.be
        da, db, dc = A_slab.desc(), B.desc(), C_slab.desc()
        seen = (id(self), id(C_slab.graph.entries), be.ptr(C_slab.graph.entries))     # see sparse.spgemm_numeric: arrays not yet seen are new
        if getattr(C_slab, "_kk_numeric_seen", None) != seen:
            self.set("entries_computed", 0)
        C_slab._kk_numeric_seen = seen
        check(self.lib, self.lib.kkamd_dist_spgemm_numeric(self._op, A_slab.numRows(), A_slab.numCols(), B.numCols(), da.d_row_map, da.d_entries, da.d_values,
                                                           db.d_row_map, db.d_entries, db.d_values, dc.d_row_map, dc.d_entries, dc.d_values,
                                                           da.offset_type, da.value_type, be.stream()))
        return C_slab

    def __del__(self):
        try:
            if self._op:
                self.lib.kkamd_dist_spgemm_destroy(self._op)
        except Exception:
            pass
        self._op = None


def spgemm_row_slab(A_slab, B, offsets=None, rank=0):
    """one rank's slab of C = A * B through the row-partitioned operator (symbolic + numeric); without a partition the slab is
    taken as the whole of a one-rank job"""
    offs = offsets if offsets is not None else [0, A_slab.numRows()]
    op = DistSpgemm(offs, rank, A_slab.backend)
    Cs = op.symbolic(A_slab, B)
    op.numeric(A_slab, B, Cs)
    return Cs


class _GlooTransport:
    """TEST transport: kkamd_transport_t over a torch.distributed CPU process group (gloo).  "Device" pointers are host
    pointers under the emulator backend."""

    def __init__(self, group, world):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.world = torch, dist, group, world

        def view(ptr, nbytes):
            return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),)))

        def all_gather(ctx, d_send, d_recv, nbytes, stream):
            try:
                outs = [torch.empty(int(nbytes), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(outs, view(d_send, nbytes).clone(), group=group)
                view(d_recv, nbytes * world).copy_(torch.cat(outs))
                return 0
            except Exception as e:             # never raise through the C frame
                print("gloo transport all_gather failed:", e, flush=True)
                return 3

        def exchange(ctx, nsend, d_send, send_bytes, send_peer, nrecv, d_recv, recv_bytes, recv_peer, stream):
            try:
                ops = [dist.P2POp(dist.isend, view(d_send[i], send_bytes[i]).clone(), int(send_peer[i]), group=group) for i in range(nsend)]
                bufs = [torch.empty(int(recv_bytes[i]), dtype=torch.uint8) for i in range(nrecv)]
                ops += [dist.P2POp(dist.irecv, bufs[i], int(recv_peer[i]), group=group) for i in range(nrecv)]
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                for i in range(nrecv):
                    view(d_recv[i], recv_bytes[i]).copy_(bufs[i])
                return 0
            except Exception as e:
                print("gloo transport exchange failed:", e, flush=True)
                return 3

        self._ag, self._ex = _capi.ALL_GATHER_FN(all_gather), _capi.EXCHANGE_FN(exchange)     # keep the callbacks alive
        self.struct = _capi.Transport(None, self._ag, self._ex)


class _GlooDeviceTransport(_GlooTransport):
    """kkamd_transport_t over a CPU process group for DEVICE buffers (staged through the host, device-synchronous): for ranks that
    share one GPU -- where RCCL refuses to form a communicator -- e.g. the two-process GPU tests of the exchange modes."""

    def __init__(self, group, world):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.world = torch, dist, group, world

        def view(ptr, nbytes):
            return torch.as_tensor(_DeviceView(ptr, nbytes, "|u1"), device="cuda")

        def all_gather(ctx, d_send, d_recv, nbytes, stream):
            try:
                torch.cuda.synchronize()
                outs = [torch.empty(int(nbytes), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(outs, view(d_send, nbytes).cpu(), group=group)
                view(d_recv, nbytes * world).copy_(torch.cat(outs))
                torch.cuda.synchronize()
                return 0
            except Exception as e:             # never raise through the C frame
                print("gloo device transport all_gather failed:", e, flush=True)
                return 3

        def exchange(ctx, nsend, d_send, send_bytes, send_peer, nrecv, d_recv, recv_bytes, recv_peer, stream):
            try:
                torch.cuda.synchronize()
                ops = [dist.P2POp(dist.isend, view(d_send[i], send_bytes[i]).cpu(), int(send_peer[i]), group=group) for i in range(nsend)]
                bufs = [torch.empty(int(recv_bytes[i]), dtype=torch.uint8) for i in range(nrecv)]
                ops += [dist.P2POp(dist.irecv, bufs[i], int(recv_peer[i]), group=group) for i in range(nrecv)]
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                for i in range(nrecv):
                    view(d_recv[i], recv_bytes[i]).copy_(bufs[i])
                torch.cuda.synchronize()
                return 0
            except Exception as e:
                print("gloo device transport exchange failed:", e, flush=True)
                return 3

        self._ag, self._ex = _capi.ALL_GATHER_FN(all_gather), _capi.EXCHANGE_FN(exchange)
        self.struct = _capi.Transport(None, self._ag, self._ex)


class _DeviceView:
    """a library-owned device buffer as something torch.as_tensor understands"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def transport_selftest(backend, nbytes=1 << 20):
    """preflight of the library's built-in RCCL transport on this rank alone (kkamd_dist_transport_selftest)"""
    check(backend.lib, backend.lib.kkamd_dist_transport_selftest(int(nbytes), backend.stream()))


class DistSpmv:
    def __init__(self, A_local, offsets, rank, group=None, algo="SPMV_DEFAULT", to_backend=None, exchange="auto",
                 overlap=True, dtype=np.float64, transport=None):
        """A_local: CrsMatrix slab (numRows = offsets[rank+1]-offsets[rank], numCols = global).
        to_backend: converts a torch tensor to what the backend's ptr() accepts (identity for HBM tensors;
        tests on CPU/gloo pass `lambda t: t.numpy()` for the emulator backend)."""
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.A, self.offsets, self.rank, self.group = A_local, [int(o) for o in offsets], int(rank), group
        self.world = len(offsets) - 1
        assert A_local.numRows() == offsets[rank + 1] - offsets[rank]
        assert A_local.numCols() == offsets[-1], "column indices must be global"
        self.be = A_local.backend
        self.lib = self.be.lib
        self.to_backend = to_backend or (lambda t: t)
        self.dtype = np.dtype(dtype)
        self._transport = None
        id_buf = None
        tr_ptr = None
        if transport == "rccl" and self.world == 1:
            # one rank, the library's own RCCL transport all the same (with a forced exchange the one-rank communicator runs the calls
            # an N-rank job makes: the N-rank code path on one GPU)
            raw = (C.c_char * 128)()
            check(self.lib, self.lib.kkamd_dist_unique_id(raw))
            id_buf = (C.c_char * 128).from_buffer_copy(raw.raw)
        elif transport == "rccl":
            transport = None
        if self.world > 1 and transport is not None:
            self._transport = transport            # a caller-supplied kkamd_transport_t holder (object with a .struct)
            tr_ptr = C.byref(transport.struct)
        elif self.world > 1:
            if self.be.name == "torch":
                # the library's RCCL transport: its id comes from rank 0 through the process group
                ident = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    raw = (C.c_char * 128)()
                    check(self.lib, self.lib.kkamd_dist_unique_id(raw))
                    ident = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).cuda()
                dist.broadcast(ident, src=0, group=group)
                id_buf = (C.c_char * 128).from_buffer_copy(bytes(ident.cpu().numpy().tobytes()))
            else:
                self._transport = _GlooTransport(group, self.world)
                tr_ptr = C.byref(self._transport.struct)
        offs = (C.c_int64 * (self.world + 1))(*self.offsets)
        d = A_local.desc()
        op = C.c_void_p()
        vt = _capi.F64 if self.dtype == np.dtype(np.float64) else _capi.F32
        check(self.lib, self.lib.kkamd_dist_spmv_create(C.byref(op), C.byref(d), offs, self.world, self.rank, id_buf, tr_ptr,
                                                        _ALGOS[algo], _EXCHANGE[exchange], 1 if overlap else 0, vt, self.be.stream()))
        self._op = op
        self.exchange_mode = ("local", "halo", "allgather", "allgather_p2p", "halo_set")[self.query("exchange")]
        self.exchange_bytes = self.query("exchange_bytes")
        # all-gather: the form in use (0 collective, 1 send / receive, 2 peer-to-peer pulls; -1: no all-gather) and, when the operator chose it itself,
        # what one exchange of every form took at creation (microseconds, maximum over the ranks; -1: form not available)
        self.allgather_form = self.query("allgather_form")
        self.allgather_us = {k: self.query("allgather_us_" + k) for k in ("collective", "sendrecv", "p2p")} if self.query("allgather_selected") else None
        self.interior_rows = self.query("interior_rows")

    def query(self, key):
        v = C.c_int64(0)
        check(self.lib, self.lib.kkamd_dist_spmv_query(self._op, key.encode(), C.byref(v)))
        return int(v.value)

    def x_local(self):
        """the rank's own window of the operator's full-length x buffer: an x kept here is never copied by apply()"""
        p = C.c_void_p()
        check(self.lib, self.lib.kkamd_dist_spmv_x_local(self._op, C.byref(p), None))
        n = self.offsets[self.rank + 1] - self.offsets[self.rank]
        if self.be.name == "torch":
            return self.torch.as_tensor(_DeviceView(p.value, n, "<f8" if self.dtype.itemsize == 8 else "<f4"), device="cuda")
        ct = C.c_double if self.dtype.itemsize == 8 else C.c_float
        return self.torch.from_numpy(np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(n,)))

    def apply(self, alpha, x_shard, beta, y_shard, events=None, what=0):
        """y_shard := alpha * A_local * exchanged(x) + beta * y_shard.  events = (start, end) are recorded around the step when
        given; what = 1 runs the exchange only, 2 the local SpMV only (measurement)."""
        if events:
            events[0].record()
        check(self.lib, self.lib.kkamd_dist_spmv_apply(self._op, float(alpha), self.be.ptr(self.to_backend(x_shard)), float(beta),
                                                       self.be.ptr(self.to_backend(y_shard)), int(what), self.be.stream()))
        if events:
            events[1].record()
        return y_shard

    def __del__(self):
        try:
            if self._op:
                self.lib.kkamd_dist_spmv_destroy(self._op)
        except Exception:
            pass
        self._op = None
