#!/usr/bin/env python3
"""Markdown table of a finished collection (profiles/collect.py): one row per (configuration, kernel).

  roofline_table.py profiles/r03

Columns: launch time (timed steps of `rocprofv3 --kernel-trace`), VALU wave-instructions per launch (instrumented build;
the ratio to the SQ_INSTS_VALU counter beside it), issue cycles per launch (executed opcode histogram x the issue
microbenchmark), their share of 1024 SIMDs x 2.4 GHz x launch time, the share of issue cycles whose opcode had no
microbenchmark row, active lanes per VALU instruction, fp32 rate (flops of the histogram x active lanes), HBM bytes per
launch (2 x FETCH_SIZE + WRITE_SIZE, KiB units) and their share of 8 TB/s.
"""
import glob
import json
import os
import sys


def main(d):
    print("| configuration | kernel | µs / launch | VALU insts / launch (PMC ÷ counted) | issue cycles / launch | issue frac | unpriced | "
          "active lanes | fp32 TFLOP/s | HBM MB / launch | HBM frac |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for f in sorted(glob.glob(os.path.join(d, "roofline_r0*_*.json"))):
        r = json.load(open(f))
        for cb, k in r["kernels"].items():
            name = k["name"].split("(")[0]
            lanes = k.get("active_lane_fraction")
            tf = None
            if k.get("flops_fp32_per_launch_64_lanes") and lanes:
                tf = k["flops_fp32_per_launch_64_lanes"] * lanes / (k["avg_us"] * 1e-6) / 1e12
            print("| %s | `%s` | %.1f | %.4g (%s) | %.4g | **%s** | %.2f %% | %s | %s | %s | %s |" % (
                r["config"], name, k["avg_us"], k.get("valu_insts_per_launch_counted", 0), k.get("sq_insts_valu_over_counted"),
                k.get("issue_cycles_per_launch", 0), k.get("issue_frac_of_peak_clock"), 100.0 * k.get("unpriced_share_of_cycles", 0),
                lanes, "%.1f" % tf if tf else "-", "%.0f" % (k["hbm_bytes_per_launch"] / 1e6) if "hbm_bytes_per_launch" in k else "-",
                k.get("hbm_frac")))


if __name__ == "__main__":
    main(sys.argv[1])
