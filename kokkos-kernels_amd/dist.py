"""1-D row-partitioned SpMV across the GPUs of one node: rank r owns the contiguous row slab
[offsets[r], offsets[r+1]) of A (local row_map, GLOBAL column indices), the matching slab of y and the
matching shard of x.  One exchange step per SpMV: an all-gather of the x shards (RCCL over xGMI when the
process group is "nccl"), then the local planned SpMV.  The reference has no distributed layer at all
(SURVEY F2); this is the multi-GPU row of the scope table.

One process per GPU (torch.distributed); no data-path collective other than the x all-gather.
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


class DistSpmv:
    def __init__(self, A_local, offsets, rank, group=None, algo="SPMV_DEFAULT", to_backend=None):
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

    def _buffers(self, like):
        if self.x_full is None:
            n = self.offsets[-1]
            if self.equal:
                self.x_full = self.torch.empty(n, dtype=like.dtype, device=like.device)
            else:
                self._pad = self.torch.empty(self.world * self.max_shard, dtype=like.dtype, device=like.device)
                self.x_full = self.torch.empty(n, dtype=like.dtype, device=like.device)
        return self.x_full

    def gather_x(self, x_shard):
        """all-gather the shards of x into the full vector every rank needs for its slab"""
        x_full = self._buffers(x_shard)
        if self.world == 1:
            x_full.copy_(x_shard)
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

    def apply(self, alpha, x_shard, beta, y_shard):
        """y_shard := alpha * A_local * allgather(x) + beta * y_shard"""
        x_full = self.gather_x(x_shard)
        spmv(self.handle, "N", alpha, self.A, self.to_backend(x_full), beta, self.to_backend(y_shard))
        return y_shard
