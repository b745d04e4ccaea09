"""A loop-back kkamd_transport_t for one-process tests of the multi-GPU SpMV operator on a real device: the collectives are answered
from the test's own copy of the GLOBAL x (what the peers would have sent), so the halo lists, the interior / boundary split, the
communication stream + events and the zero-copy x window all run on device pointers.  The protocol BETWEEN processes is covered by
tests/test_dist_gloo.py (CPU, gloo) and tests/test_gpu_dist_two_process.py."""
import torch

from kokkos_kernels_amd import _capi
from kokkos_kernels_amd.dist import _DeviceView


def view(ptr, nbytes):
    return torch.as_tensor(_DeviceView(ptr, nbytes, "|u1"), device="cuda")


class Loopback:
    """x: the global vector (device, float64); offsets: row offsets of the ranks; ranges[p]: (first, last) column of rank p's slab"""

    def __init__(self, x, offsets, rank, ranges):
        self.calls, self.base, self.log = 0, None, []
        world = len(offsets) - 1
        shard_bytes = 8 * (offsets[rank + 1] - offsets[rank])

        def all_gather(ctx, d_send, d_recv, nbytes, stream):
            torch.cuda.synchronize()
            mine = view(d_send, nbytes).clone(); out = view(d_recv, nbytes * world)
            for p in range(world):
                if p == rank: out[p * nbytes:(p + 1) * nbytes] = mine
                elif nbytes == shard_bytes and nbytes > 16:                  # the data all-gather: every rank's x shard
                    out[p * nbytes:(p + 1) * nbytes] = x[offsets[p]:offsets[p + 1]].view(torch.uint8)
                elif self.calls == 0: out[p * nbytes:(p + 1) * nbytes] = torch.tensor(ranges[p], dtype=torch.int64, device="cuda").view(torch.uint8)
                else: out[p * nbytes:(p + 1) * nbytes] = torch.tensor([0.01, 0.0], dtype=torch.float64, device="cuda").view(torch.uint8)
            self.calls += 1
            torch.cuda.synchronize()
            return 0

        def exchange(ctx, nsend, d_send, send_bytes, send_peer, nrecv, d_recv, recv_bytes, recv_peer, stream):
            torch.cuda.synchronize()
            for i in range(nrecv):                       # serve every receive from the global x (what the peer would have sent)
                lo = (int(d_recv[i]) - self.base) // 8; cnt = int(recv_bytes[i]) // 8
                view(d_recv[i], recv_bytes[i]).copy_(x[lo:lo + cnt].view(torch.uint8))
                self.log.append(("recv", int(recv_peer[i]), lo, cnt))
            for i in range(nsend):
                self.log.append(("send", int(send_peer[i]), (int(d_send[i]) - self.base) // 8, int(send_bytes[i]) // 8))
            torch.cuda.synchronize()
            return 0
        self._ag, self._ex = _capi.ALL_GATHER_FN(all_gather), _capi.EXCHANGE_FN(exchange)
        self.struct = _capi.Transport(None, self._ag, self._ex)
