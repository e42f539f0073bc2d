#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of an asynchronous moving-camera run: how many frames were binned twice on the device
(the overflow redo's K1 ran in full instead of leaving at once), and what the launches between K1 and the selection cost.
usage: redo_rate.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    by[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for n, v in sorted(by.items()):
    d = sorted(x[1] for x in v)
    print("%-55s launches %6d  median %8.1f us  mean %8.1f  p90 %8.1f  max %8.1f" % (n[:55], len(d), d[len(d) // 2] / 1e3, sum(d) / len(d) / 1e3, d[int(len(d) * 0.9)] / 1e3, d[-1] / 1e3))
k1 = [x[1] for x in by.get("splat::preprocess_kernel<true, false, false>", [])]
if k1:
    short = sum(1 for d in k1 if d < 60e3)
    print("K1 launches %d: %d left at once (< 60 us: a redo launch of a frame that needed none), %d ran" % (len(k1), short, len(k1) - short))
    sh = sorted(d for d in k1 if d < 60e3)
    if sh: print("   the launches that left at once: median %.1f us, mean %.1f, max %.1f" % (sh[len(sh) // 2] / 1e3, sum(sh) / len(sh) / 1e3, sh[-1] / 1e3))
sc = sorted(x[1] for x in by.get("splat::scan_bucket_kernel<256>", []))
if sc:
    print("scan launches %d: %d under 12 us (a redo launch that left at once)" % (len(sc), sum(1 for d in sc if d < 12e3)))
ly = sorted(x[1] for x in by.get("splat::layout_kernel<256>", []) + by.get("splat::layout_kernel", []))
if ly:
    print("redo layout launches %d: %d under 12 us; median %.1f us" % (len(ly), sum(1 for d in ly if d < 12e3), ly[len(ly) // 2] / 1e3))
