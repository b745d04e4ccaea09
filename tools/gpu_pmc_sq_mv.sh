#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/sqmv.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
A = kk.laplace_matrix("FE", 300, 300, 300)
X = torch.rand(A.numCols(), 16, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numRows(), 16, dtype=torch.float64, device="cuda")
h = kk.SPMVHandle("SPMV_DEFAULT")
for _ in range(3): kk.spmv(h, "N", 1.0, A, X, 0.0, Y)
torch.cuda.synchronize()
PY
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmcsqmv_$i -o b -- python /tmp/sqmv.py $R > $OUT/pmcsqmv_$i.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/pmcsqmv_*/b_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmv_mv2" not in k: continue
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("spmv_mv2_kernel<4,4> " + " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in agg.items()))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
