#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs a tools/gpu_r2.sh call left under gpurun_out/prof_<tag>_{stats,FETCH_SIZE,WRITE_SIZE}
into small summaries under gpurun_out/summary/<round>/ (copy the ones to keep into profiles/<round>/): kernel stats with
trimmed names, and per-kernel memory-side counter means stamped with the hash of the kernel sources they were measured on."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rnd = sys.argv[1] if len(sys.argv) > 1 else "round2"
out = os.path.join(ROOT, "gpurun_out", "summary", rnd); os.makedirs(out, exist_ok=True)
import bench
# the translation unit a profile tag measures, by the tag's first word(s): every file is stamped with THAT unit's hash
# (round 5 stamped "spgemm_s20" and "bench_full" with kk_spmv.hip's: the tags were looked up whole)
UNIT = (("spgemm", "kk_spgemm.hip"), ("mv4", "kk_spmv_mv4.h"), ("mv5", "kk_spmv_mvblk.hip"), ("mv6", "kk_spmv_mvnnz.hip"), ("mv", "kk_spmv_mv.hip"),
        ("struct", "kk_spmv_struct.hip"), ("colslab", "kk_spmv_colslab.hip"), ("dist", "kk_dist.hip"), ("bench_full", None), ("bench", "kk_spmv.hip"))


def unit_of_tag(tag):
    for prefix, unit in UNIT:
        if tag.startswith(prefix): return unit
    return None            # several units (or unknown): stamped with every unit's hash


def stamp(tag):
    u = unit_of_tag(tag)
    return (bench.kernel_source_sha(u), u) if u else (json.dumps(bench.all_unit_shas(), sort_keys=True), "all units")


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


for stats_dir in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_*_stats")):
    tag = os.path.basename(stats_dir)[5:-6]
    sha, unit = stamp(tag)
    f = find(stats_dir, "kernel_stats.csv")
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as g:
            g.write("# kernel_source_sha %s (%s)\n" % (sha, unit))
            w = csv.writer(g); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    res = {"kernel_source_sha": sha, "kernel_source_unit": unit,
           "note": "FETCH_SIZE/WRITE_SIZE in KB per launch, separate rocprofv3 --pmc passes; gfx950 FETCH_SIZE counts 64 B per 128 B request on "
                   "wide coalesced reads (MI355X_MICROARCH.md, HBM section): corrected_read_bytes = FETCH_SIZE*1024*2", "counters": {}}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = find(os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, c)), "counter_collection.csv")
        if not f: continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:110], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "kk::" in k[0]:
                res["counters"]["%s | %s" % k] = {"launches": len(v), "mean_KB": round(sum(v) / len(v), 2)}
    if res["counters"]:
        json.dump(res, open(os.path.join(out, ("bench_n1" if tag == "bench" else tag) + "_pmc_hbm.json"), "w"), indent=1)
for sq_dir in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_sq_*")):
    if not os.path.isdir(sq_dir): continue
    f = find(sq_dir, "counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:110], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, "sq_counters.txt"), "a") as g:
        g.write("# kernel_source_sha %s (%s; %s)\n" % (stamp(os.path.basename(sq_dir)[8:]) + (os.path.basename(sq_dir),)))
        for k, v in sorted(agg.items()):
            if "kk::" in k[0]: g.write("%-110s %-32s launches %3d mean %.4g\n" % (k[0], k[1], len(v), sum(v) / len(v)))
for name in ("bench.json", "bench_mv3.jsonl", "bench_mv4.jsonl", "probe_mfma_f64.txt"):
    p = os.path.join(ROOT, "gpurun_out", name)
    if os.path.exists(p): open(os.path.join(out, name), "w").write(open(p).read())
print("wrote", out, "sha", bench.kernel_source_sha())
