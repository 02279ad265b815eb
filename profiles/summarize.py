#!/usr/bin/env python3
"""Summarise a profiles/run_rocprof.sh output directory: per-kernel duration stats from the
kernel trace and per-kernel sums of each PMC counter (one row per dispatch and counter)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k_bounce", "k_iteration", "k_mesh", "k_intersect", "k_raygen", "k_gather", "k_shade_fake", "k_tonemap",
              "k_sort_hist", "k_sort_perm", "k_cache_first"):
        if k in name:
            tail = ""
            if "k_bounce" in name:
                import re
                m = re.search(r"k_bounce<(\d), (true|false), (\d), (true|false)>", name)
                if m:
                    mesh = {"1": ",mesh-tiles", "2": ",mesh-bvh", "3": ",mesh-prepass"}.get(m.group(3), "")
                    tail = "<%s,%s%s%s>" % ({"0": "fused", "1": "isect", "2": "cache0"}[m.group(1)],
                                            "compact" if m.group(2) == "true" else "inplace", mesh,
                                            "" if m.group(4) == "true" else ",global-scene")
            return k + tail
    return name[:60]


# SIMD cycles per wave-instruction by class, measured with profiles/microbench/valu_peak.hip on MI355X
# (profiles/r02/valu_peak_r02.json, >= 2 waves per SIMD): fp32 add / mul / fma and int32 add 2.45; packed, fp64, 64-bit
# integer, conversions, min / max / med3, compare + select ~4.2; transcendentals 8.2
ISSUE_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.45, "SQ_INSTS_VALU_MUL_F32": 2.45, "SQ_INSTS_VALU_FMA_F32": 2.45,
                "SQ_INSTS_VALU_TRANS_F32": 8.2, "SQ_INSTS_VALU_INT32": 2.6, "SQ_INSTS_VALU_INT64": 4.3,
                "SQ_INSTS_VALU_CVT": 4.4, "SQ_INSTS_VALU_ADD_F64": 4.3, "SQ_INSTS_VALU_MUL_F64": 4.3,
                "SQ_INSTS_VALU_FMA_F64": 4.3, "SQ_INSTS_VALU_TRANS_F64": 16.0}
OTHER_CYCLES = 4.2          # min / max, compare, select, move, bit operations, cross-lane: everything not in a class above


def valu_model(per_kernel, durations_us, clock_ghz=2.1, simds=1024):
    """Issue-cycle model of a kernel's vector instruction stream: sum over classes of count x measured cycles,
    against SIMDs x clock x time."""
    out = {}
    for k, c in per_kernel.items():
        if "SQ_INSTS_VALU" not in c or k not in durations_us:
            continue
        total = c["SQ_INSTS_VALU"]
        classed = sum(c.get(n, 0.0) for n in ISSUE_CYCLES)
        cycles = sum(c.get(n, 0.0) * w for n, w in ISSUE_CYCLES.items()) + max(0.0, total - classed) * OTHER_CYCLES
        avail = simds * clock_ghz * 1e3 * durations_us[k]          # SIMD-cycles in one average launch
        out[k] = {"valu_wave_insts": total, "classified": classed, "issue_cycles": cycles, "simd_cycles": avail,
                  "issue_utilisation": cycles / avail if avail else 0.0,
                  "mix": {n[14:]: c.get(n, 0.0) / total for n in ISSUE_CYCLES if c.get(n, 0.0)}}
    return out


def main(d):
    avg_us = {}
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        print("== kernel trace:", os.path.relpath(f, d))
        print("%-34s %8s %12s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us"))
        tot = sum(sum(v) for v in dur.values())
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            avg_us[k] = sum(v) / len(v) / 1e3
            print("%-34s %8d %12.1f %12.2f %12.2f %12.2f  %5.1f%%" % (k, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3,
                                                                 min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))
    per_kernel = defaultdict(dict)
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
                    per_kernel[k][c] = v / n
                    print("%-34s %-24s dispatches=%-6d sum=%-16.0f per_dispatch=%.1f" % (k, c, n, v, v / n))
    model = valu_model(per_kernel, avg_us)
    if model:
        print("== vector issue model (count per class x measured cycles per wave-instruction; 1024 SIMDs at 2.1 GHz)")
        for k, m in sorted(model.items(), key=lambda kv: -kv[1]["issue_cycles"]):
            print("%-34s insts/launch=%.4g  issue cycles=%.4g of %.4g SIMD-cycles = %.1f %%   mix: %s" % (
                k, m["valu_wave_insts"], m["issue_cycles"], m["simd_cycles"], 100 * m["issue_utilisation"],
                " ".join("%s=%.1f%%" % (n, 100 * f) for n, f in sorted(m["mix"].items(), key=lambda kv: -kv[1]))))
    return model


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
    model = main(sys.argv[1])
    if len(sys.argv) > 2:
        import subprocess
        meta = {"config": "c2", "batch": 64, "flags": "compact"}                  # bench.py defaults
        args = sys.argv[3:]
        for k in ("config", "batch", "flags"):
            if "--" + k in args:
                v = args[args.index("--" + k) + 1]
                meta[k] = int(v) if k == "batch" else v
        # which kernel sources these numbers belong to: bench.py drops them when the sources have changed
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        try:
            import bench
            meta["csrc_sha16"] = bench.csrc_digest()
        except Exception as e:                                                    # noqa
            meta["csrc_sha16"] = None
        dom = max(model.items(), key=lambda kv: kv[1]["issue_cycles"]) if model else None
        if dom:
            meta["valu_model"] = {"kernel": dom[0], "issue_utilisation": round(dom[1]["issue_utilisation"], 4),
                                  "wave_insts_per_launch": dom[1]["valu_wave_insts"],
                                  "issue_cycles_per_launch": dom[1]["issue_cycles"],
                                  "mix": {k: round(v, 4) for k, v in dom[1]["mix"].items()}}
        traffic_json(sys.argv[1], sys.argv[2], meta)
