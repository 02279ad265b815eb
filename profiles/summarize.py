#!/usr/bin/env python3
"""Summarise a profiles/run_rocprof.sh output directory: per-kernel duration stats from the
kernel trace and per-kernel sums of each PMC counter (one row per dispatch and counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k_bounce", "k_intersect", "k_raygen", "k_gather", "k_shade_fake", "k_tonemap", "k_sort"):
        if k in name:
            tail = ""
            if "k_bounce" in name:
                tail = "<%s,%s>" % ("isect" if "Li1E" in name else "fused", "compact" if "Lb1E" in name else "inplace")
            return k + tail
    return name[:60]


def main(d):
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("== kernel trace:", os.path.relpath(f, d))
        print("%-34s %8s %12s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
        tot = sum(sum(v) for v in dur.values())
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            print("%-34s %8d %12.1f %12.2f %12.2f %12.2f  %5.1f%%" % (k, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3,
                                                                 min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
    for sub in sorted(glob.glob(os.path.join(d, "p*"))):
        if not os.path.isdir(sub):
            continue
        for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(int)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[(k, r["Counter_Name"])] += 1
            print("== counters:", os.path.relpath(f, d))
            for k in acc:
                for c, v in sorted(acc[k].items()):
                    n = cnt[(k, c)]
                    print("%-34s %-24s dispatches=%-6d sum=%-16.0f per_dispatch=%.1f" % (k, c, n, v, v / n))


if __name__ == "__main__":
    main(sys.argv[1])
