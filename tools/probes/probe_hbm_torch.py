#!/usr/bin/env python3
"""HBM ceilings seen from a kernel's point of view: fill (write-only), copy (1 read : 1 write), sum (read-only), axpy-like
(2 reads : 1 write) on 3.456 GB fp64 arrays (the size of one 27e6 x 16 right-hand-side block), HIP-event timed."""
import json, torch
n = 27_000_000 * 16
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x); z = torch.empty_like(x)


def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it


gb = n * 8 / 1e9
for name, fn, moved in (("fill (write)", lambda: y.fill_(1.5), gb), ("copy (1r:1w)", lambda: y.copy_(x), 2 * gb),
                        ("sum (read)", lambda: x.sum(), gb), ("add (2r:1w)", lambda: torch.add(x, y, out=z), 3 * gb),
                        ("mul scalar (1r:1w)", lambda: torch.mul(x, 2.0, out=z), 2 * gb)):
    ms = t(fn)
    print(json.dumps({"op": name, "ms": round(ms, 4), "GB": round(moved, 3), "TBps": round(moved / ms, 3)}), flush=True)
