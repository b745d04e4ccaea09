#!/bin/bash
# SQ/LDS counters of the SpGEMM kernels on one case; usage: gpu_pmc_spgemm.sh rmat18
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
CASE=${1:-rmat18}
cat > /tmp/sg1.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import numpy as np, torch, kk_loader, oracle
kk = kk_loader.load()
if sys.argv[2] == "laplace":
    M = kk.laplace_matrix("FE", 100, 100, 100)
else:
    Rm = oracle.rmat(int(sys.argv[2][4:]), 16)
    M = kk.CrsMatrix.from_host(Rm.nrows, Rm.ncols, Rm.row_map, Rm.entries, Rm.values, offset_dtype=np.int64)
kh = kk.KokkosKernelsHandle(); kh.create_spgemm_handle()
C = kk.spgemm_symbolic(kh, M, False, M, False)
kk.spgemm_numeric(kh, M, False, M, False, C); torch.cuda.synchronize()
PY
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmcsg_$i -o b -- python /tmp/sg1.py $R $CASE > $OUT/pmcsg_$i.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmcsg_*/b_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spgemm" not in k: continue
        k = k.split("(")[0].replace("void kk::", "")[:48]
        agg.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], 0.0)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k, " ".join("%s=%.3g" % (c, v) for c, v in d.items()))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
