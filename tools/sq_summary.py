#!/usr/bin/env python3
"""Per-kernel summary of the SQ / TCC counter passes tools/gpu_r5.sh `sq` left under gpurun_out/prof_sq_*/ (one rocprofv3 --pmc pass per
counter group): mean counter values per launch and what follows from them -- vector-ALU and LDS busy fractions, vector instructions, L2
misses in bytes.  Usage: python tools/sq_summary.py <kernel name filter> [products per launch, "name:count,name:count"] > profiles/roundN/<file>"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else "spgemm"
prods = {}
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
    k, v = kv.rsplit(":", 1); prods[k] = float(v)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_sq_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if flt in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc <group> (one pass per group, tools/gpu_r5.sh sq); values are means per launch, summed over the 8 XCDs.")
print("# SQ_ACTIVE_INST_* count in units of 4 cycles; GRBM_GUI_ACTIVE / 8 = cycles of the launch.")
for name, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    n = max(len(v) for v in cs.values())
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc < 1e5: continue
    print("\n== %s   (%d launches per pass, %.2f ms per launch at 2.4 GHz)" % (name, n, cyc / 2.4e6))
    for k in sorted(m): print("   %-34s %.4g" % (k, m[k]))
    line = []
    if "SQ_ACTIVE_INST_VALU" in m: line.append("vector ALU busy %.0f %%" % (100 * m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc))
    if "SQ_ACTIVE_INST_LDS" in m: line.append("LDS busy %.0f %%" % (100 * m["SQ_ACTIVE_INST_LDS"] * 4 / 256 / cyc))
    if "SQ_LDS_BANK_CONFLICT" in m: line.append("LDS bank-conflict cycles %.0f %% of the launch" % (100 * m["SQ_LDS_BANK_CONFLICT"] / 256 / cyc))
    if line: print("   -> " + ", ".join(line))
    if "TCC_MISS_sum" in m: print("   -> L2 misses %.3g x 128 B = %.1f GB per launch = %.2f TB/s" % (m["TCC_MISS_sum"], m["TCC_MISS_sum"] * 128 / 1e9, m["TCC_MISS_sum"] * 128 / (cyc / 2.4e9) / 1e12))
    p = next((v for k, v in prods.items() if k in name), None)
    if p and "SQ_INSTS_VALU" in m:
        print("   -> %.3g products per launch: %.0f vector instructions (lanes) per product, L2-miss bytes per product %.1f (algorithmic: 12)"
              % (p, m["SQ_INSTS_VALU"] * 64 / p, m.get("TCC_MISS_sum", 0) * 128 / p))
# aggregate over the kernels whose name contains KK_SQ_AGG's filter ("filter:products:algorithmic bytes"), e.g. all value kernels of one numeric call
spec = os.environ.get("KK_SQ_AGG", "")
if spec:
    af, ap, ab = spec.split(":"); ap = float(ap); ab = float(ab)
    tot_valu = tot_miss = tot_cyc = 0.0; names = []
    for name, cs in agg.items():
        if af in name:
            m = {k: sum(v) / len(v) for k, v in cs.items()}
            tot_valu += m.get("SQ_INSTS_VALU", 0); tot_miss += m.get("TCC_MISS_sum", 0); tot_cyc += m.get("GRBM_GUI_ACTIVE", 0) / 8; names.append(name.split("<")[0].replace("kk::", "") + "<" + name.split("<")[1][:30] if "<" in name else name)
    print("\n== all kernels matching '%s' of one numeric call (%d kernels, %.1f ms): %.3g products" % (af, len(names), tot_cyc / 2.4e6, ap))
    print("   -> %.0f vector instructions (lanes) per product; L2 misses %.1f GB = %.2f x the algorithmic bytes of these kernels (%.1f GB)"
          % (tot_valu * 64 / ap, tot_miss * 128 / 1e9, tot_miss * 128 / ab, ab / 1e9))
