#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TA_BUSY_avr TD_TD_BUSY_sum" "GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmcmv_$i -o b -- python $R/tools/sweep_spmv.py --what mv --iters 4 > $OUT/pmcmv_$i.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcmv_*/b_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmv_mv" in r["Kernel_Name"] or "pack_rows" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f.split("/")[1], k[0], k[1], "mean=%.4g" % (sum(v)/len(v)), "n=%d" % len(v))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
