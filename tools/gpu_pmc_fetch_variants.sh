#!/bin/bash
# memory-side read traffic (FETCH_SIZE) and time of the C2 SpMV for tile orders / non-temporal loads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cat > /tmp/fv.py <<'PY'
import sys; sys.path.insert(0, sys.argv[1])
import torch, kk_loader
kk = kk_loader.load()
nt, g, it = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
A = kk.laplace_matrix("FE", 300, 300, 300)
x = torch.rand(A.numCols(), dtype=torch.float64, device="cuda"); y = torch.zeros(A.numRows(), dtype=torch.float64, device="cuda")
h = kk.SPMVHandle("SPMV_DEFAULT"); h.set("nontemporal", nt); h.set("xcd_remap", g)
for _ in range(3): kk.spmv(h, "N", 1.0, A, x, 0.0, y)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(it): kk.spmv(h, "N", 1.0, A, x, 0.0, y)
b.record(); torch.cuda.synchronize()
print("nt=%d group=%d: %.4f ms" % (nt, g, a.elapsed_time(b) / it))
PY
for v in "0 0" "0 16" "1 16" "1 0"; do
  python /tmp/fv.py $R $v 30
done
cd /tmp
for v in "0 0" "0 16" "1 16"; do
  tag=$(echo $v | tr ' ' '_')
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcfv_$tag -o b -- python /tmp/fv.py $R $v 3 > $OUT/pmcfv_$tag.log 2>&1
  echo "pmc $v rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmcfv_*/b_counter_collection.csv")):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "spmv_stream3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    print(f.split("/")[1], "launches", len(v), "FETCH_SIZE mean KB %.0f -> x2 = %.2f GB" % (sum(v) / len(v), sum(v) / len(v) * 2048 / 1e9))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
