#!/usr/bin/env python3
"""The compositor lane of a rocprofv3 --kernel-trace CSV (steady state, middle of the run): durations of the compositor
and of the repair launch behind it, the gaps between them, the frame period.  usage: lane_gaps.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("splat::", "")[:24]) for r in rows)
lane = [k for k in ks if "composite" in k[2]]
lane = lane[len(lane) // 4: -len(lane) // 8]
cd, rd, g1, g2 = [], [], [], []
for a, b in zip(lane, lane[1:]):
    if "exact" in a[2] and "repair" in b[2]:
        cd.append((a[1] - a[0]) / 1e3); g1.append((b[0] - a[1]) / 1e3); rd.append((b[1] - b[0]) / 1e3)
    if "exact" in a[2] and "exact" in b[2]:
        cd.append((a[1] - a[0]) / 1e3); g2.append((b[0] - a[1]) / 1e3)
    if "repair" in a[2] and "exact" in b[2]:
        g2.append((b[0] - a[1]) / 1e3)
for n, v in (("compositor", cd), ("gap to the repair", g1), ("repair launch", rd), ("gap to the next compositor", g2)):
    v = sorted(v)
    if v: print("%-27s n %4d mean %6.1f median %6.1f p10 %6.1f p90 %6.1f us" % (n, len(v), st.mean(v), st.median(v), v[len(v) // 10], v[9 * len(v) // 10]))
ex = [k for k in lane if "exact" in k[2]]
print("frame period %.1f us" % st.mean([(b[0] - a[0]) / 1e3 for a, b in zip(ex, ex[1:])]))
