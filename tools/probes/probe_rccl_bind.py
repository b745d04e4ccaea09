#!/usr/bin/env python3
"""One-GPU check of the library's RCCL binding (the multi-rank path itself needs a second GPU): kkamd_dist_unique_id dlopens
RCCL -- the copy torch already loaded, matched by soname -- resolves the nine entry points and asks it for a communicator id."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, kk_loader
kk = kk_loader.load()
lib = kk.torch_backend().lib
raw = (C.c_char * 128)()
rc = lib.kkamd_dist_unique_id(raw)
maps = [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l]
print("kkamd_dist_unique_id rc", rc, "non-zero id bytes", sum(1 for b in raw.raw if b != 0), "rccl mapped from", sorted(set(maps)))
