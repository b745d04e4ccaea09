#!/usr/bin/env python3
"""BASELINE config 4 at its stated scale on one GPU: slabs of the 8-GPU row partition of R-MAT scale 22 through kkamd_dist_spgemm_* (bench.py's
c4_slab_of_8 section on its own).  Usage: python tools/bench_c4_slab.py [rank,rank,...] ; KK_VERBOSE=2 prints the library's stage times of the last repetition."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, kk_loader, bench
kk = kk_loader.load()
ranks = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else None
print(json.dumps(bench._spgemm_c4_slabs(kk, torch, time.perf_counter(), 600.0, verbose=int(os.environ.get("KK_VERBOSE", "0")), ranks=ranks)))
