#!/usr/bin/env python3
"""Print a window of the kernel timeline from a rocprofv3 --kernel-trace CSV (start, end, duration in us, queue):
   python tools/timeline.py gpurun_out/.../*_kernel_trace.csv [n_compositor_launches]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("splat::", "")[:34], r["Queue_Id"]) for r in rows)
comp = [k for k in ks if "composite" in k[2]]
mid = comp[len(comp) // 2: len(comp) // 2 + n]
t0 = mid[0][0]
for k in ks:
    if t0 - 20000 <= k[0] <= mid[-1][1]:
        print("%9.1f %9.1f %7.1f  q%s %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2]))
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(comp[len(comp) // 4: -len(comp) // 4], comp[len(comp) // 4 + 1: -len(comp) // 4 + 1])]
durs = [(k[1] - k[0]) / 1e3 for k in comp[len(comp) // 4: -len(comp) // 4]]
print("compositor launches: mean duration %.1f us, mean gap to the next %.1f us (middle half of the run)" % (sum(durs) / len(durs), sum(gaps) / len(gaps)))
