#!/usr/bin/env python3
"""Kernel timeline of the LAST n compositor launches of a run (a short `bench.py --steps 20` as the driver times it):
   python tools/short_run_timeline.py <kernel_trace.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("splat::", "")[:28]) for r in rows)
comp = [k for k in ks if "composite" in k[2]]
sel = comp[-n:]
t0 = sel[0][0]
prev = None
for k in sel:
    print("K4 start %8.1f end %8.1f dur %6.1f gap %6.1f" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, ((k[0] - prev) / 1e3) if prev else 0.0))
    prev = k[1]
print("span of these %d launches: %.1f us" % (n, (sel[-1][1] - sel[0][0]) / 1e3))
first_k1 = [k for k in ks if "preprocess" in k[2] and k[0] < sel[0][0]]
if first_k1:
    print("the K1 in front of the first of them started at %.1f" % ((first_k1[-1][0] - t0) / 1e3))
