#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for var in 3 1; do
 i=0
 for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmcv${var}_$i -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --knob nnz_per_thread=16 --knob nontemporal=0 --knob stream_variant=$var > $OUT/pmcv${var}_$i.log 2>&1
  echo "var $var pmc group $i rc=$?"
 done
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcv*/b_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmv_stream" in r["Kernel_Name"] and "fixup" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f.split("/")[1], k, "mean=%.4g" % (sum(v)/len(v)), "n=%d" % len(v))
PY
grep -h "rocprofv3\|rror" $OUT/pmcv1_3.log | head -5
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
