#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of one R-MAT SpGEMM with parts of the dense-row kernels switched off
# (measurement build libkkamd_ablate.so; spgemm_debug bits: see kk_spgemm.hip).  Usage: tools/prof_spgemm_ablate.sh SCALE "0 1 4 256 2"
R=${GRAFT_REPO_ROOT:-/root/repo}; SCALE=${1:-20}; OUT=$R/gpurun_out/spgemm_ablate.txt; : > $OUT
export TMPDIR=/tmp KKAMD_LIBRARY=libkkamd_ablate.so KK_REPS=1
cd /tmp
for d in ${2:-0}; do
  rm -rf /tmp/abl_$d
  KK_DEFAULTS=spgemm_debug=$d${KK_EXTRA:+,$KK_EXTRA} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$d -o p -- python $R/tools/bench_spgemm_quick.py $SCALE > /dev/null 2>&1
  echo "== spgemm_debug=$d" >> $OUT
  python - >> $OUT <<PY
import csv,glob
f=glob.glob("/tmp/abl_$d/**/p_kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9] if f else []: print("%-70s calls %s avg %.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6))
PY
done
cat $OUT
