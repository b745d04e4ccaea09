#!/usr/bin/env python3
"""Timeline of the kernels of the LAST occurrence of a marker kernel in a rocprofv3 --kernel-trace CSV: start / end relative to the first kernel of
the window, stream, duration -- to see what overlaps what.  Usage: python tools/trace_timeline.py <dir with *_kernel_trace.csv> <marker kernel substring> <ms before> <ms after>"""
import csv, glob, sys
d, marker, before, after = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ks.sort()
last = max(i for i, k in enumerate(ks) if marker in k[2])
t0 = ks[last][0]
for s, e, n, q in ks:
    if t0 - before * 1e6 <= s <= t0 + after * 1e6:
        print("%9.3f .. %9.3f ms  (%7.3f)  q %s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n.replace("void kk::", "")[:70]))
