#!/usr/bin/env python3
"""Summarise a profiles/run_rocprof.sh output directory: per-kernel duration stats from the
kernel trace and per-kernel sums of each PMC counter (one row per dispatch and counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k_bounce", "k_mesh", "k_intersect", "k_raygen", "k_gather", "k_shade_fake", "k_tonemap", "k_sort"):
        if k in name:
            tail = ""
            if "k_bounce" in name:
                import re
                m = re.search(r"k_bounce<(\d), (true|false)(?:, (true|false|\d))?>", name)
                if m:
                    mesh = {"true": ",mesh", "1": ",mesh-tiles", "2": ",mesh-bvh", "3": ",mesh-prepass"}.get(m.group(3), "")
                    tail = "<%s,%s%s>" % ("isect" if m.group(1) == "1" else "fused",
                                          "compact" if m.group(2) == "true" else "inplace", mesh)
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


def traffic_json(d, out_path, meta):
    """bytes per k_bounce launch = 2 * FETCH_SIZE (gfx950 reports half of a coalesced read) + WRITE_SIZE,
    both in KiB, from their separate PMC passes."""
    import json
    tot = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == name and "k_bounce" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        if vals:
            tot[name] = sum(vals) / len(vals)
    valu = []
    for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "SQ_INSTS_VALU" and "k_bounce" in r["Kernel_Name"]:
                valu.append(float(r["Counter_Value"]))
    if len(tot) == 2:
        meta = dict(meta)
        if valu:
            meta["valu_wave_insts_per_launch"] = sum(valu) / len(valu)
        meta.update({"fetch_kib_per_launch_raw": tot["FETCH_SIZE"], "write_kib_per_launch": tot["WRITE_SIZE"],
                     "bytes_per_launch": int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024),
                     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 "
                             "(gfx950 counts 64 B per 128-B request; calibrated on k_gather's known bytes)"})
        json.dump(meta, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
    if len(sys.argv) > 2:
        traffic_json(sys.argv[1], sys.argv[2], {"config": "c2", "batch": 64, "flags": "compact"})   # bench.py defaults
