#!/bin/bash
# One gpurun call: parity tests, smoke, sweep, bench, rocprof (stats + PMC).  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
WHAT=${1:-all}
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
nproc
if [[ $WHAT == all || $WHAT == *smoke* ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
fi
if [[ $WHAT == all || $WHAT == *pytest* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
fi
if [[ $WHAT == all || $WHAT == *sweep* ]]; then
  timeout 900 python tools/sweep_spmv.py --n 300 --what ${SWEEP_WHAT:-copy,spmv} > $OUT/sweep.jsonl 2> $OUT/sweep.err; echo "sweep rc=$?"; tail -3 $OUT/sweep.err; cat $OUT/sweep.jsonl
fi
if [[ $WHAT == all || $WHAT == *bench* ]]; then
  timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err; cat $OUT/bench.json
fi
if [[ $WHAT == all || $WHAT == *prof* ]]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $R/bench.py --steps 30 --no-cpu-baseline > $OUT/prof_stats.log 2>&1; echo "prof stats rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1; echo "prof fetch rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1; echo "prof write rc=$?"
  cd $R
  find $OUT -name "*.csv" | head -20; for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do head -12 $f; done
  # keep only the small CSVs (the merge-back limit is 64 MiB)
  find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
fi
echo "== done"
