#!/usr/bin/env python3
"""Measurement (libkkamd_ablate.so): what the numeric phase's entry emission for the rows without a stored bitmap consists of.
   One symbolic + numeric, then the numeric call repeated into the same (already correct) C arrays with parts of
   spgemm_dense_cols_kernel<true> switched off: bit 1 = no stores of entries(C), bit 4 = no count / emit walk at all.
   KKAMD_LIBRARY=libkkamd_ablate.so python tools/proto/spgemm_emit_split.py [scale]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, kk_loader, oracle
kk = kk_loader.load(); lib = kk.torch_backend().lib
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = oracle.rmat(scale, 16)
M = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
Cm = kk.spgemm_symbolic(kh, M, False, M, False)
kk.spgemm_numeric(kh, M, False, M, False, Cm); torch.cuda.synchronize()
sh = kh.get_spgemm_handle()
out = {}
for dbg in (0, 1, 4, 0):
    kk._capi.check(lib, lib.kkamd_set_default(b"spgemm_debug", dbg))
    sh.set("entries_computed", 0)                      # write entries(C) again (same arrays: the value kernels find correct entries whatever this call leaves out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    kk.spgemm_numeric(kh, M, False, M, False, Cm)
    torch.cuda.synchronize(); out.setdefault("debug_%d_ms" % dbg, []).append(round((time.perf_counter() - t0) * 1e3, 2))
print(json.dumps(out))
