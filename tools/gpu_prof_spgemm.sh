#!/bin/bash
# per-kernel breakdown of one symbolic+numeric SpGEMM (C = A*A); usage: gpu_prof_spgemm.sh [laplace|rmatNN]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CASE=${1:-laplace}
cat > /tmp/sg.py <<'PY'
import sys, time; sys.path.insert(0, sys.argv[1])
import numpy as np, torch, kk_loader, oracle
kk = kk_loader.load()
import os
for kv in os.environ.get("KK_SPGEMM_KNOBS", "").split(","):
    if kv: kk._capi.check(kk.torch_backend().lib, kk.torch_backend().lib.kkamd_set_default(("spgemm_" + kv.split("=")[0]).encode(), int(kv.split("=")[1])))
case = sys.argv[2]
if case == "laplace":
    M = kk.laplace_matrix("FE", 100, 100, 100)
else:
    Rm = oracle.rmat(int(case[4:]), 16)
    M = kk.CrsMatrix.from_host(Rm.nrows, Rm.ncols, Rm.row_map, Rm.entries, Rm.values, offset_dtype=np.int64)
for rep in range(2):
    kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    C = kk.spgemm_symbolic(kh, M, False, M, False); torch.cuda.synchronize(); t1 = time.perf_counter()
    kk.spgemm_numeric(kh, M, False, M, False, C); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s sym %.2f ms num %.2f ms" % (case, (t1-t0)*1e3, (t2-t1)*1e3))
    kh.destroy_spgemm_handle(); del C
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_spgemm_$CASE -o sg -- python /tmp/sg.py $R $CASE > $OUT/prof_spgemm_$CASE.log 2>&1
echo rc=$?; grep " sym " $OUT/prof_spgemm_$CASE.log
python3 - $OUT/prof_spgemm_$CASE/sg_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'], "%.3f ms" % (float(r['AverageNs'])/1e6), r['Percentage'])
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
