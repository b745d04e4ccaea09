#!/bin/bash
# SQ instruction-mix / wait counters of the headline SpMV kernel (and spmv_struct) -- separate --pmc passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/sq.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
A = kk.laplace_matrix("FE", 300, 300, 300)
x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
h = kk.SPMVHandle("SPMV_DEFAULT")
for _ in range(4): kk.spmv(h, "N", 1.0, A, x, 0.0, y)
for _ in range(4): kk.spmv_struct("N", 2, (300, 300, 300), 1.0, A, x, 0.0, y)
torch.cuda.synchronize()
PY
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmcsq_$i -o b -- python /tmp/sq.py $R > $OUT/pmcsq_$i.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict(); cnt = collections.Counter()
for f in sorted(glob.glob("gpurun_out/pmcsq_*/b_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmv_stream3" not in k and "struct_interior" not in k: continue
        k = k.split("(")[0].replace("void kk::", "")[:60]
        agg.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in d.items()))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
