#!/usr/bin/env python3
"""Copy what tools/profile_round.sh collected (gpurun_out/prof_<tag>/) into profiles/:
  <tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats summary, as written by rocprofv3
  <tag>_pmc_{FETCH_SIZE,WRITE_SIZE,SQ}.csv   per-kernel averages of each --pmc pass
  <tag>_bench.json                  the bench.py line of the same box
  traffic.json                      HBM bytes per launch per kernel (tools/pmc_traffic.py rules)
usage: summarise_profiles.py <tag> [workload]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


def kname(full):
    return full.split("(")[0].strip()


def pmc_avg(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(kname(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else "C3"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(one(os.path.join(src, "trace", "**", "*_kernel_stats.csv")), os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench.json"))
    iso = glob.glob(os.path.join(src, "trace_isolated", "**", "*_kernel_stats.csv"), recursive=True)
    if iso:
        shutil.copy(iso[0], os.path.join(dst, tag + "_kernel_stats_isolated.csv"))
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ", "SQ2"):
        found = glob.glob(os.path.join(src, "pmc_" + c, "**", "*_counter_collection.csv"), recursive=True)
        if not found:
            if c == "SQ2":
                continue
            raise SystemExit("missing counter pass " + c)
        agg = pmc_avg(found[0])
        with open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, c)), "w") as f:
            f.write("Kernel,Counter,Dispatches,Average\n")
            for (k, cn), v in sorted(agg.items()):
                f.write('"%s",%s,%d,%g\n' % (k, cn, len(v), sum(v) / len(v)))
                per[(k, cn)] = sum(v) / len(v)
    res = {}
    for k in sorted({k for k, _ in per}):
        fe, wr = per.get((k, "FETCH_SIZE"), 0.0), per.get((k, "WRITE_SIZE"), 0.0)
        f = fe * 1024.0 * 2.0            # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
        w = wr * 1024.0
        short = k.replace("void ", "").replace("splat::", "")
        res[short] = int(f + w)
        res[short + ":detail"] = {"FETCH_SIZE_KiB_raw": fe, "WRITE_SIZE_KiB_raw": wr, "read_bytes_corrected": int(f), "write_bytes": int(w)}
        if (k, "SQ_INSTS_VALU") in per and (k, "SQ_BUSY_CYCLES") in per and per[(k, "SQ_BUSY_CYCLES")] > 0:
            # SQ_INSTS_VALU: wave64 instructions of the launch, 2 issue cycles each on a SIMD-32
            # (MI355X_MICROARCH.md); SQ_BUSY_CYCLES is summed over the chip's 32 shader engines, so
            # BUSY/32 is the launch's length in shader clocks; 256 CUs x 4 SIMDs issue in parallel.
            busy = per[(k, "SQ_BUSY_CYCLES")] / 32.0
            res[short + ":detail"]["valu_insts"] = per[(k, "SQ_INSTS_VALU")]
            res[short + ":detail"]["busy_cycles"] = busy
            res[short + ":detail"]["valu_issue_util"] = round(per[(k, "SQ_INSTS_VALU")] * 2.0 / (1024.0 * busy), 4)
    tj = os.path.join(dst, "traffic.json")
    data = json.load(open(tj)) if os.path.exists(tj) else {}
    data[workload] = res
    data[workload + ":source"] = tag
    json.dump(data, open(tj, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        if not k.endswith(":detail"):
            print("%-40s %8.1f MB / launch" % (k, v / 1e6))


if __name__ == "__main__":
    main()
