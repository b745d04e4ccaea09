"""1-D row-partitioned SpMV across the GPUs of one node: rank r owns the contiguous row slab
[offsets[r], offsets[r+1]) of A (local row_map, GLOBAL column indices), the matching slab of y and the
matching shard of x.  One exchange step per SpMV: an all-gather of the x shards (RCCL over xGMI when the
process group is "nccl"), then the local planned SpMV.  The reference has no distributed layer at all
(SURVEY F2); this is the multi-GPU row of the scope table.

Exchange step.  "allgather": every rank receives every shard (what BASELINE.json's north star names).
"halo" (default when it moves less than half of that): a rank only needs the x entries its slab's column
indices reference -- for the column RANGE [cmin, cmax] of the slab it receives, from each peer, the piece of that
range the peer owns (point-to-point RCCL send/recv, one batch per SpMV).  For a 1-D slab of a 3-D stencil that is
one grid plane from each neighbour (2 x 600^2 x 8 B = 5.8 MB per rank at 600^3) instead of 1.5 GB; for a matrix
whose slab touches every column it degenerates to the all-gather.  This is row N1 of SURVEY 8(f) (an importer
built from the column set of each slab).

Overlap (halo mode): rows whose columns all lie in the rank's own x range form the INTERIOR -- the longest contiguous
run of such rows; the rows before and after it (one grid plane each for a stencil slab) are the boundary.  Interior,
head and tail are zero-copy row-range views of the slab (rebased row_map, offset entries/values) with their own SpMV
plans.  Per SpMV: start the point-to-point exchange, run the interior SpMV while the halo is in flight, wait, run the
two boundary SpMVs.

One process per GPU (torch.distributed); no other data-path collective.
"""
import numpy as np

from .sparse import SPMVHandle, spmv


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


def spgemm_row_slab(A_slab, B):
    """Row-partitioned SpGEMM: rank r owns a contiguous row slab of A (local row_map) and the matching row slab of
    C = A * B; B is replicated, so there is NO data-path communication (partition where the path shards).
    With slabs balanced by multiplications (work_balanced_offsets) this is what makes BASELINE config C4 as specified
    (R-MAT scale 22: nnz(C) = 7.2e10 = 863 GB) fit: 108 GB of C per GPU on eight GPUs."""
    from .sparse import spgemm
    return spgemm(A_slab, False, B, False)


class DistSpmv:
    def __init__(self, A_local, offsets, rank, group=None, algo="SPMV_DEFAULT", to_backend=None, exchange="auto",
                 overlap=True):
        """A_local: CrsMatrix slab (numRows = offsets[rank+1]-offsets[rank], numCols = global).
        to_backend: converts a torch tensor to what the backend's ptr() accepts (identity for HBM tensors;
        tests on CPU/gloo pass `lambda t: t.numpy()` for the emulator backend)."""
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.A, self.offsets, self.rank, self.group = A_local, list(offsets), rank, group
        self.world = len(offsets) - 1
        assert A_local.numRows() == offsets[rank + 1] - offsets[rank]
        assert A_local.numCols() == offsets[-1], "column indices must be global"
        self.handle = SPMVHandle(algo)
        self.to_backend = to_backend or (lambda t: t)
        sizes = np.diff(self.offsets)
        self.equal = bool((sizes == sizes[0]).all())
        self.max_shard = int(sizes.max())
        self.x_full = None
        self._pad = None
        self.exchange = exchange          # "auto" | "halo" | "allgather"
        self._plan = None                 # (mode, send list, recv list)
        self.exchange_bytes = None        # bytes this rank receives per SpMV
        self.algo = algo
        self.overlap = overlap            # halo mode: interior rows computed while the halo is in flight
        self._split = None                # [(handle, sub-matrix, row_begin, row_end)]: interior first, then boundary parts

    def _entries_minmax(self):
        ent = self.A.graph.entries
        if self.A.nnz() == 0:
            return 0, -1
        if hasattr(ent, "min") and not isinstance(ent, np.ndarray):
            return int(ent.min().item()), int(ent.max().item())
        return int(ent.min()), int(ent.max())

    def _setup_exchange(self, like):
        """decide halo vs all-gather and build the per-peer segment lists (once per operator)"""
        torch, dist = self.torch, self.dist
        n, me, offs = self.offsets[-1], self.rank, self.offsets
        item = like.element_size()
        full_bytes = (n - (offs[me + 1] - offs[me])) * item
        if self.world == 1:
            self._plan = ("local", [], []); self.exchange_bytes = 0
            return
        cmin, cmax = self._entries_minmax()
        mine = torch.tensor([cmin, cmax], dtype=torch.int64, device=like.device)
        allr = torch.empty(2 * self.world, dtype=torch.int64, device=like.device)
        dist.all_gather_into_tensor(allr, mine, group=self.group)
        allr = allr.cpu().tolist()
        recv, send = [], []
        for p in range(self.world):
            if p == me:
                continue
            lo, hi = max(cmin, offs[p]), min(cmax + 1, offs[p + 1])          # what I need from p
            if hi > lo:
                recv.append((p, lo, hi))
            plo, phi = allr[2 * p], allr[2 * p + 1]
            lo, hi = max(plo, offs[me]), min(phi + 1, offs[me + 1])          # what p needs from me
            if hi > lo:
                send.append((p, lo - offs[me], hi - offs[me]))
        halo_bytes = sum(hi - lo for _, lo, hi in recv) * item
        # every rank must take the same decision: all-reduce the largest halo fraction
        frac = torch.tensor([halo_bytes / max(full_bytes, 1)], dtype=torch.float64, device=like.device)
        dist.all_reduce(frac, op=dist.ReduceOp.MAX, group=self.group)
        use_halo = self.exchange == "halo" or (self.exchange == "auto" and frac.item() < 0.5)
        if use_halo:
            self._plan = ("halo", send, recv); self.exchange_bytes = halo_bytes
            if self.overlap:
                self._setup_overlap()
        else:
            self._plan = ("allgather", [], []); self.exchange_bytes = full_bytes

    def _setup_overlap(self):
        """interior = longest contiguous run of rows that reference only this rank's own x entries"""
        from .sparse import CrsMatrix
        A, me0, me1 = self.A, self.offsets[self.rank], self.offsets[self.rank + 1]
        m = A.numRows()
        rm, ent = A.graph.row_map, A.graph.entries
        is_np = isinstance(ent, np.ndarray)
        if m == 0 or A.nnz() == 0:
            return
        outside = (ent < me0) | (ent >= me1)
        if is_np:
            cs = np.concatenate([[0], np.cumsum(outside, dtype=np.int64)])
            per_row = cs[np.asarray(rm[1:], dtype=np.int64)] - cs[np.asarray(rm[:-1], dtype=np.int64)]
            bad = np.nonzero(per_row)[0]
        else:
            torch = self.torch
            cs = torch.zeros(ent.numel() + 1, dtype=torch.int32, device=ent.device)
            torch.cumsum(outside, 0, dtype=torch.int32, out=cs[1:])
            rml = rm.long()
            per_row = cs[rml[1:]] - cs[rml[:-1]]
            bad = torch.nonzero(per_row).flatten().cpu().numpy()
            del cs, outside, per_row
        if bad.size == 0:
            return                                   # nothing depends on the halo
        edges = np.concatenate([[-1], bad, [m]])
        gaps = np.diff(edges) - 1
        g = int(np.argmax(gaps))
        r_lo, r_hi = int(edges[g]) + 1, int(edges[g + 1])
        # the planned kernel wants 16-byte aligned entries / values: start the interior and the tail on rows whose
        # first entry sits at a multiple of 4 (a misaligned view still works, through the slower no-analysis kernel)
        def offset_of(r):
            return int(rm[r]) if is_np else int(rm[r].item())
        for _ in range(64):
            if r_lo < r_hi and offset_of(r_lo) % 4:
                r_lo += 1
        for _ in range(64):
            if r_hi > r_lo and offset_of(r_hi) % 4:
                r_hi -= 1
        if r_hi - r_lo < m // 2:
            return                                   # not worth splitting
        parts = []
        for a, b in ((r_lo, r_hi), (0, r_lo), (r_hi, m)):
            if b <= a:
                continue
            if is_np:
                p0, p1 = int(rm[a]), int(rm[b])
                sub_rm = (rm[a:b + 1] - rm[a]).astype(rm.dtype)
            else:
                p0, p1 = int(rm[a].item()), int(rm[b].item())
                sub_rm = (rm[a:b + 1] - rm[a]).contiguous()
            sub = CrsMatrix(b - a, A.numCols(), sub_rm, ent[p0:p1], A.values[p0:p1], backend=A.backend)
            parts.append((SPMVHandle(self.algo), sub, a, b))
        self._split = parts

    def _buffers(self, like):
        if self.x_full is None:
            n = self.offsets[-1]
            # zero-filled once: entries outside the slab's column range are never read, but must not be garbage NaNs
            self.x_full = self.torch.zeros(n, dtype=like.dtype, device=like.device)
            if not self.equal:
                self._pad = self.torch.empty(self.world * self.max_shard, dtype=like.dtype, device=like.device)
        return self.x_full

    def gather_x(self, x_shard):
        """make the x entries this rank's slab references available in the full-length buffer"""
        x_full = self._buffers(x_shard)
        if self._plan is None:
            self._setup_exchange(x_shard)
        mode, send, recv = self._plan
        me0, me1 = self.offsets[self.rank], self.offsets[self.rank + 1]
        if mode == "local":
            x_full.copy_(x_shard)
        elif mode == "halo":
            x_full[me0:me1].copy_(x_shard)
            dist = self.dist
            ops = [dist.P2POp(dist.isend, x_shard[lo:hi], p, group=self.group) for p, lo, hi in send]
            ops += [dist.P2POp(dist.irecv, x_full[lo:hi], p, group=self.group) for p, lo, hi in recv]
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        elif self.equal:
            self.dist.all_gather_into_tensor(x_full, x_shard, group=self.group)
        else:
            mine = self.torch.zeros(self.max_shard, dtype=x_shard.dtype, device=x_shard.device)
            mine[: x_shard.numel()] = x_shard
            self.dist.all_gather_into_tensor(self._pad, mine, group=self.group)
            for r in range(self.world):
                n = self.offsets[r + 1] - self.offsets[r]
                x_full[self.offsets[r]: self.offsets[r + 1]] = self._pad[r * self.max_shard: r * self.max_shard + n]
        return x_full

    def apply(self, alpha, x_shard, beta, y_shard, events=None):
        """y_shard := alpha * A_local * exchanged(x) + beta * y_shard.  events = (start, end): recorded around the
        compute part (local SpMV kernels) when given."""
        if self._plan is None:
            self._buffers(x_shard)
            self._setup_exchange(x_shard)
        if self._plan[0] == "halo" and self._split:
            x_full = self._buffers(x_shard)
            _, send, recv = self._plan
            me0, me1 = self.offsets[self.rank], self.offsets[self.rank + 1]
            x_full[me0:me1].copy_(x_shard)
            dist = self.dist
            ops = [dist.P2POp(dist.isend, x_shard[lo:hi], p, group=self.group) for p, lo, hi in send]
            ops += [dist.P2POp(dist.irecv, x_full[lo:hi], p, group=self.group) for p, lo, hi in recv]
            reqs = dist.batch_isend_irecv(ops) if ops else []
            xb = self.to_backend(x_full)
            if events:
                events[0].record()
            h, sub, a, b = self._split[0]                        # interior: needs no halo entry
            spmv(h, "N", alpha, sub, xb, beta, self.to_backend(y_shard[a:b]))
            for req in reqs:
                req.wait()
            for h, sub, a, b in self._split[1:]:
                spmv(h, "N", alpha, sub, xb, beta, self.to_backend(y_shard[a:b]))
            if events:
                events[1].record()
            return y_shard
        x_full = self.gather_x(x_shard)
        if events:
            events[0].record()
        spmv(self.handle, "N", alpha, self.A, self.to_backend(x_full), beta, self.to_backend(y_shard))
        if events:
            events[1].record()
        return y_shard
