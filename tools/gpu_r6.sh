#!/bin/bash
# Round-5 gpurun driver: WHAT is a comma list of steps.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
WHAT=${1:-pytest,bench}
has() { [[ ",$WHAT," == *",$1,"* ]]; }
echo "== box: $(rocminfo 2>/dev/null | grep -m1 -E 'gfx9[0-9a-z]+') cpus $(nproc) mem $(free -g | awk '/Mem:/{print $2" GB total, "$7" GB available"}')"
if has probe;  then ./tools/probes/probe_mfma_f64 > $OUT/probe_mfma_f64.txt 2>&1; echo "probe rc=$?"; head -12 $OUT/probe_mfma_f64.txt; fi
if has probelds; then ./tools/probes/probe_lds > $OUT/probe_lds.txt 2>&1; echo "probe_lds rc=$?"; cat $OUT/probe_lds.txt; fi
if has rccl; then timeout 600 python -m pytest tests -m gpu -x -q -rA -k "rccl or loopback or two_processes" > $OUT/pytest_rccl.log 2>&1; echo "rccl pytest rc=$?"; tail -8 $OUT/pytest_rccl.log
  timeout 600 python bench.py --gpus 1 --exchange allgather --steps 20 --no-cpu-baseline > $OUT/bench_n1_allgather.json 2> $OUT/bench_n1_allgather.err; echo "bench --gpus 1 --exchange allgather rc=$?"; tail -2 $OUT/bench_n1_allgather.err; cat $OUT/bench_n1_allgather.json; fi
if has abc3; then for i in 1 2 3; do for t in build/r3:round3 .:head; do timeout 300 python tools/ab_c3.py ${t%%:*} ${t##*:} >> $OUT/ab_c3.jsonl 2>> $OUT/ab_c3.err; done; done; echo "abc3 rc=$?"; cat $OUT/ab_c3.jsonl; fi
if has pytestk; then timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q -k "$PYTEST_K" > $OUT/pytest_gpu_k.log 2>&1; echo "pytest -k rc=$?"; tail -6 $OUT/pytest_gpu_k.log; fi
if has pytest; then timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q -rA > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head; fi
if has pytestfast; then KK_SKIP_HEAVY=1 timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_fast.log 2>&1; echo "pytest(fast) rc=$?"; tail -4 $OUT/pytest_gpu_fast.log; fi
if has bench;  then timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err; cat $OUT/bench.json; fi
if has mv3;    then timeout 900 python tools/bench_mv3.py 300 ${MV3_ARGS:-} > $OUT/bench_mv3.jsonl 2> $OUT/bench_mv3.err; echo "mv3 rc=$?"; tail -3 $OUT/bench_mv3.err; cat $OUT/bench_mv3.jsonl; fi
if has mv4;    then timeout 900 python tools/bench_mv4.py 300 ${MV4_ARGS:-} > $OUT/bench_mv4.jsonl 2> $OUT/bench_mv4.err; echo "mv4 rc=$?"; tail -3 $OUT/bench_mv4.err; cat $OUT/bench_mv4.jsonl; fi
if has extra;  then timeout ${EXTRA_TIMEOUT:-900} bash -c "$EXTRA_CMD" > $OUT/extra.log 2>&1; echo "extra rc=$?"; tail -${EXTRA_TAIL:-40} $OUT/extra.log; fi
prof() {   # prof <tag> <command...>: kernel stats + FETCH_SIZE and WRITE_SIZE in separate passes
  local tag=$1; shift
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${tag}_stats -o p -- "$@" > $OUT/prof_${tag}_stats.log 2>&1; echo "prof $tag stats rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/prof_${tag}_$c -o p -- "$@" > $OUT/prof_${tag}_$c.log 2>&1; echo "prof $tag $c rc=$?"
  done
  cd $R
}
if has profx; then prof ${PROF_TAG:-x} bash -c "$PROF_CMD"; fi
if has statsx; then cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${PROF_TAG:-x}_stats -o p -- bash -c "$PROF_CMD" > $OUT/prof_${PROF_TAG:-x}_stats.log 2>&1; echo "stats rc=$?"; cd $R; fi
if has pmcx; then cd /tmp; timeout 900 rocprofv3 --kernel-trace --pmc $PMC_LIST --output-format csv -d $OUT/prof_sq_${PROF_TAG:-x} -o p -- bash -c "$PROF_CMD" > $OUT/prof_sq_${PROF_TAG:-x}.log 2>&1; echo "pmcx rc=$?"; cd $R; fi
if has profbench; then prof bench python $R/bench.py --steps 30 --no-cpu-baseline --extras-seconds 0; fi
if has profmv;    then prof mv python $R/tools/bench_mv3.py 300 quick; fi
if has profmv4;   then prof mv4 python $R/tools/bench_mv4.py 300 quick; fi
if has profspgemm; then prof spgemm python $R/tools/bench_spgemm_quick.py ${SPGEMM_SCALE:-20}; fi
if has spgemm; then timeout 600 python tools/bench_spgemm_quick.py ${SPGEMM_SCALES:-18,20} > $OUT/spgemm_quick.jsonl 2> $OUT/spgemm_quick.err; echo "spgemm rc=$?"; cat $OUT/spgemm_quick.jsonl; fi
if has sq; then
  cd /tmp
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
    tag=$(echo $grp | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/prof_sq_$tag -o p -- ${SQ_CMD:-python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --extras-seconds 0} > $OUT/prof_sq_$tag.log 2>&1; echo "sq [$grp] rc=$?"
  done
  cd $R
fi
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
python tools/extract_profiles.py round6 2>&1 | tail -3
echo "== done"
