import os, sys, time
sys.path.insert(0, sys.argv[1])
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|NUMA node\\(s\\)|Thread|Core' | head -6")
os.system("free -g | head -2")
import subprocess
code = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("OMP_PROC_BIND", "spread"); os.environ.setdefault("OMP_PLACES", "threads")
import numpy as np, oracle
A = oracle.laplace3d("FE", 160, 160, 160)
ft = oracle.first_touch
rm32 = ft(A.row_map.astype(np.int32)); ent = ft(A.entries); val = ft(A.values)
x = ft(np.random.default_rng(1).random(A.ncols)); y = ft(np.zeros(A.nrows))
oracle.spmv_omp(rm32, ent, val, 1.0, x, 0.0, y)
t0 = time.perf_counter(); it = 0
while time.perf_counter() - t0 < 2.0:
    oracle.spmv_omp(rm32, ent, val, 1.0, x, 0.0, y); it += 1
el = time.perf_counter() - t0
print("threads %s: %.2f GFLOP/s" % (os.environ.get("OMP_NUM_THREADS"), 2.0 * A.nnz * it / el / 1e9), flush=True)
'''
open("/tmp/cpu_one.py", "w").write(code)
for nt in (4, 8, 16, 32, 64, 128, 256):
    env = dict(os.environ, OMP_NUM_THREADS=str(nt))
    subprocess.run([sys.executable, "/tmp/cpu_one.py", sys.argv[1]], env=env)
