#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the spmv_struct and SpMV_MV kernels on C2 / C3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/hbmx.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
A = kk.laplace_matrix("FE", 300, 300, 300)
x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
X = torch.rand(A.numCols(), 16, dtype=torch.float64, device="cuda"); Y = torch.zeros(A.numRows(), 16, dtype=torch.float64, device="cuda")
h = kk.SPMVHandle("SPMV_DEFAULT")
for _ in range(3): kk.spmv_struct("N", 2, (300, 300, 300), 1.0, A, x, 0.0, y)
for _ in range(3): kk.spmv(h, "N", 1.0, A, X, 0.0, Y)
torch.cuda.synchronize()
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmchbmx_$c -o b -- python /tmp/hbmx.py $R > $OUT/pmchbmx_$c.log 2>&1
  echo "pmc $c rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for f in sorted(glob.glob("gpurun_out/pmchbmx_*/b_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "struct_interior" in k or "spmv_mv2" in k:
            agg[(k.split("(")[0].replace("void kk::", "")[:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        res["%s | %s" % (k, c)] = {"launches": len(v), "mean_KB": round(sum(v) / len(v), 1)}
print(json.dumps(res, indent=1))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
