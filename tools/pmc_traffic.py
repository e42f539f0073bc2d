#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes, as the MI355X guide
prescribes) into per-kernel HBM bytes per launch.

Units/corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a 16 B/lane coalesced stream (128-B requests
tallied at 64 B) -> doubled.  Calibration in this repo's own access pattern: preprocess_kernel
reads 160 B/Gaussian as coalesced float4 planes (+4 B index) = 246 MB at N = 1.5 M; FETCH_SIZE*2
gives 241 MB.  WRITE_SIZE is taken as is: composite_exact_kernel writes the 8.29 MB image and
WRITE_SIZE reads 8.13 MB... (KiB).

usage: pmc_traffic.py <dir with FETCH_SIZE_counter_collection.csv, WRITE_SIZE_counter_collection.csv> <workload> <out.json>
"""
import collections
import csv
import json
import os
import sys


def avg_by_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("splat::", "")
        agg[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    d, workload, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch = avg_by_kernel(os.path.join(d, "FETCH_SIZE_counter_collection.csv"))
    write = avg_by_kernel(os.path.join(d, "WRITE_SIZE_counter_collection.csv"))
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, 0.0) * 1024.0 * 2.0      # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        w = write.get(k, 0.0) * 1024.0
        res[k] = int(f + w)
        res[k + ":detail"] = {"FETCH_SIZE_KiB_raw": fetch.get(k, 0.0), "WRITE_SIZE_KiB_raw": write.get(k, 0.0),
                              "read_bytes_corrected": int(f), "write_bytes": int(w)}
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[workload] = res
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        if not k.endswith(":detail"):
            print("%-28s %8.1f MB / launch" % (k, v / 1e6))


if __name__ == "__main__":
    main()
