#!/usr/bin/env python3
"""hbm_calibrate.py [OUTDIR]  (on the GPU box): FETCH_SIZE / WRITE_SIZE of rocprofv3 against KNOWN byte counts in the
access patterns of k_bounce -- dword-per-lane pool rows (aligned and straddling two physical tiles), partial-row survivor
appends, scattered 16-B final-colour stores -- and in the guide's calibrated 16-B-per-lane streaming case as control
(profiles/microbench/hbm_patterns.hip).  Two counter passes (never combined with a trace); prints and writes
OUTDIR/hbm_calibration.txt: per pattern the algorithmic bytes and the counters' raw readings (KB x 1024), i.e. the factor
by which a reading in that pattern has to be scaled."""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04", "hbm_cal")


def main():
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join("/tmp", "hbm_patterns")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe,
                    os.path.join(ROOT, "profiles", "microbench", "hbm_patterns.hip")], check=True)
    env = dict(os.environ, TMPDIR="/tmp")
    plain = subprocess.run([exe], cwd="/tmp", env=env, capture_output=True, text=True, check=True).stdout
    algo = []
    for line in plain.splitlines():
        m = re.match(r"ALGO (\S+)\s+read_MB (\S+) write_MB (\S+)", line)
        if m:
            algo.append((m.group(1), float(m.group(2)), float(m.group(3))))
    readings = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, ctr.lower())
        subprocess.run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", exe], cwd="/tmp", env=env,
                       stdout=open(os.path.join(OUT, ctr.lower() + ".log"), "w"), stderr=subprocess.STDOUT, timeout=600)
        rows = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr and ("k_read" in r["Kernel_Name"] or "k_write" in r["Kernel_Name"]):
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
        # one row per (dispatch, XCD/instance) possibly: sum per dispatch
        per = {}
        for did, name, v in rows:
            per.setdefault(did, [name, 0.0])[1] += v
        readings[ctr] = [per[k] for k in sorted(per)]
    # durations: a kernel trace of its own (never combined with counters)
    d = os.path.join(OUT, "trace")
    subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", exe], cwd="/tmp", env=env,
                   stdout=open(os.path.join(OUT, "trace.log"), "w"), stderr=subprocess.STDOUT, timeout=600)
    durs = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if ("k_read" in r["Kernel_Name"] or "k_write" in r["Kernel_Name"]))
        durs = [(e - s0) / 1e3 for s0, e in rows]
    n = len(algo)
    lines = ["%-26s %10s %13s %7s %10s %13s %7s %9s %8s" % ("pattern", "read MB", "FETCH_SIZE MB", "ratio", "write MB", "WRITE_SIZE MB", "ratio", "us", "GB/s")]
    for rep in range(2):
        for i, (name, rmb, wmb) in enumerate(algo):
            k = rep * n + i
            f = readings["FETCH_SIZE"][k][1] * 1024 / 1e6 if k < len(readings["FETCH_SIZE"]) else float("nan")
            w = readings["WRITE_SIZE"][k][1] * 1024 / 1e6 if k < len(readings["WRITE_SIZE"]) else float("nan")
            us = durs[k] if k < len(durs) else float("nan")
            lines.append("%-26s %10.1f %13.1f %7s %10.1f %13.1f %7s %9.1f %8.0f" % (name + ("" if rep == 0 else " (2)"), rmb, f, "%.3f" % (f / rmb) if rmb else "-",
                                                                                  wmb, w, "%.3f" % (w / wmb) if wmb else "-", us, (rmb + wmb) / us * 1e3 if us == us else 0))
    text = "\n".join(lines) + "\n"
    open(os.path.join(OUT, "hbm_calibration.txt"), "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
