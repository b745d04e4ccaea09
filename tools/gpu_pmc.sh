#!/bin/bash
# PMC passes over bench.py (separate runs per counter group, kernel-trace only -- see gpurun rules)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
KNOBS="${KNOBS:---knob nnz_per_thread=16 --knob nontemporal=0 --knob stream_variant=0}"
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $KNOBS > $OUT/pmc_$i.log 2>&1
  echo "pmc group $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_*/b_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmv_stream" in r["Kernel_Name"] and "fixup" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f.split("/")[1], k, "mean=%.4g" % (sum(v)/len(v)), "n=%d" % len(v))
PY
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
