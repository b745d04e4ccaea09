#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmcmvb_$i -o b -- python $R/tools/sweep_spmv.py --what mv --iters 4 > $OUT/pmcmvb_$i.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcmvb_*/b_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    # dispatch order: per layout (right,left) x (mvk,rm) in ((0,1),(0,0),(3,1),(3,0)) ; timeit = 1 warm + iters
    agg = collections.OrderedDict()
    prev = None; run = 0
    for r in rows:
        k = r["Kernel_Name"][:60]
        if "spmv_mv2" not in k: continue
        if k != prev: run += 1; prev = k
        agg.setdefault((run, k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f.split("/")[1], k, "mean=%.4g n=%d" % (sum(v)/len(v), len(v)))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
