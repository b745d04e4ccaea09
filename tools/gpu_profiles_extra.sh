#!/bin/bash
# rocprofv3 kernel-stats summaries for the rows other than the headline bench: SpGEMM (R-MAT s20, 27-pt 100^3) and the
# C2 / C3 / spmv_struct kernels; copied into profiles/<round>/ by the caller.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_prof_spgemm.sh rmat20 > $OUT/extra_rmat20.log 2>&1
bash tools/gpu_prof_spgemm.sh laplace > $OUT/extra_laplace.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2c3 -o c2 -- python $R/tools/run_configs.py c2 > $OUT/prof_c2c3.log 2>&1
echo "c2c3 rc=$?"
cd $R
for f in $OUT/prof_spgemm_rmat20/sg_kernel_stats.csv $OUT/prof_spgemm_laplace/sg_kernel_stats.csv $OUT/prof_c2c3/c2_kernel_stats.csv; do echo "== $f"; head -8 $f | cut -c1-160; done
find $OUT -name "*.db" -delete; find $OUT -size +20M -delete
