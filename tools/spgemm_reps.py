import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, kk_loader, oracle
kk = kk_loader.load()
for kv in os.environ.get("KK_DEFAULTS", "").split(","):
    if kv: kk._capi.check(kk.torch_backend().lib, kk.torch_backend().lib.kkamd_set_default(kv.split("=")[0].encode(), int(kv.split("=")[1])))
R = oracle.rmat(20, 16)
M = kk.CrsMatrix.from_host(R.nrows, R.ncols, R.row_map, R.entries, R.values, offset_dtype=np.int64)
sync = torch.cuda.synchronize
for rep in range(6):
    kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
    sync(); t0 = time.perf_counter()
    Cm = kk.spgemm_symbolic(kh, M, False, M, False)
    sync(); t1 = time.perf_counter()
    kk.spgemm_numeric(kh, M, False, M, False, Cm)
    sync(); t2 = time.perf_counter()
    kk.spgemm_numeric(kh, M, False, M, False, Cm)
    sync(); t3 = time.perf_counter()
    print("rep %d: symbolic %.1f numeric %.1f reuse %.1f ms  free %.1f GB" % (rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, torch.cuda.mem_get_info()[0]/2**30), flush=True)
    kh.destroy_spgemm_handle(); del Cm
