#!/usr/bin/env python3
"""Roofline evidence of one round, per configuration (run on the GPU box):  [PT_ROUND=r04] python3 profiles/collect.py [config ...]
(profiles/collect_r03.py is round 3's copy of this script, kept with its evidence.)

For every configuration (bench.py command line) this collects, each in its own process:
  * rocprofv3 --kernel-trace --stats              -> calls and average duration per kernel;
  * rocprofv3 --pmc <SQ counters>                 -> SQ_INSTS_VALU (cross-check of the histogram), SQ_ACTIVE_INST_VALU,
                                                     SQ_THREAD_CYCLES_VALU (active lanes), SQ_BUSY_CYCLES, ...;
  * rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -> HBM bytes per launch (FETCH x 2: MI355X_MICROARCH.md);
  * the EXECUTED opcode histogram of the configuration's kernels: builds whose kernel counts its basic blocks
    (profiles/tools/build_counted.sh -> .ab/cnt_*/, built beforehand in the build container), priced opcode by opcode
    with the issue costs of profiles/microbench/issue_ops (gen_issue_ops.py; measured in this same call);
  * the plain bench.py line.
Counters are never combined with trace domains.  Results: gpurun_out/<round>/ (copy into profiles/<round>/: profiles/tools/publish.sh) and
profiles/traffic.json (what bench.py's `roofline` object reads; keyed by configuration and by the sha of the library).
"""
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("PT_ROUND", "r06")
OUT = os.path.join(ROOT, "gpurun_out", ROUND)
SIMDS, PEAK_GHZ = 1024, 2.4

# instrumented builds: name -> substring of the mangled kernel name
COUNTED = {
    "fn": "k_bounceILi0ELb1ELi0ELb1ELb0ELb0E", "fg": "k_bounceILi0ELb1ELi0ELb1ELb1ELb0E",
    "sn": "k_bounceILi0ELb1ELi0ELb1ELb0ELb1E", "sg": "k_bounceILi0ELb1ELi0ELb1ELb1ELb1E",
    "tn": "k_bounceILi0ELb1ELi1ELb0ELb0ELb0E", "tg": "k_bounceILi0ELb1ELi1ELb0ELb1ELb0E",
    "pn": "k_bounceILi0ELb1ELi3ELb1ELb0ELb0E", "pg": "k_bounceILi0ELb1ELi3ELb1ELb1ELb0E",
    "km": "k_meshILb1E", "it": "k_iterationILb1E",
}
# rocprof kernel names of the same kernels
PROF_NAME = {
    "fn": r"k_bounce<0, true, 0, true, false, false>", "fg": r"k_bounce<0, true, 0, true, true, false>",
    "sn": r"k_bounce<0, true, 0, true, false, true>", "sg": r"k_bounce<0, true, 0, true, true, true>",
    "tn": r"k_bounce<0, true, 1, false, false, false>", "tg": r"k_bounce<0, true, 1, false, true, false>",
    "pn": r"k_bounce<0, true, 3, true, false, false>", "pg": r"k_bounce<0, true, 3, true, true, false>",
    "km": r"k_mesh<true>", "it": r"k_iteration<true>",
}
# configuration -> (bench.py arguments, count_run arguments (config, flags, steps, batch), [(counted build, stage, launches per step)])
CONFIGS = {
    "c2": ("--config c2", ("c2", "compact", 3, 64), [("fn", "bounce", 7), ("fg", "bounce", 1)]),
    "c2_1spp": ("--config c2 --batch 1 --steps 200 --warmup 20", ("c2", "compact", 50, 1), [("it", "bounce", 1)]),
    "c3": ("--config c3", ("c3", "compact", 3, 64), [("fn", "bounce", 15), ("fg", "bounce", 1)]),
    "c3_sort": ("--config c3 --flags compact,sort", ("c3", "compact,sort", 3, 64), [("sn", "bounce", 15), ("sg", "bounce", 1)]),
    "c4_loop": ("--config c4 --flags compact --batch 4 --steps 3 --warmup 1", ("c4", "compact", 1, 4), [("tn", "bounce", 7), ("tg", "bounce", 1)]),
    "c4_bvh": ("--config c4 --flags compact,bvh --steps 10 --warmup 2", ("c4", "compact,bvh", 2, 64),
               [("pn", "bounce", 7), ("pg", "bounce", 1), ("km", "mesh", 8)]),
    "c5": ("--config c5 --batch 4 --steps 20 --warmup 4", ("c5", "compact", 2, 4), [("fn", "bounce", 7), ("fg", "bounce", 1)]),
}
SQ_PASS = "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_SALU"


def sh(cmd, log, env=None, timeout=600):
    e = dict(os.environ)
    e.setdefault("TMPDIR", "/tmp")
    if env:
        e.update(env)
    with open(log, "w") as f:
        return subprocess.run(cmd, cwd="/tmp", env=e, stdout=f, stderr=subprocess.STDOUT, timeout=timeout).returncode


def build_digest():
    sys.path.insert(0, ROOT)
    import bench
    return bench.build_digest()


def rocprof(tag, what, bench_args):
    d = os.path.join(OUT, tag, what.split()[0].replace("--", "").replace("-", "_") if what.startswith("--kernel") else "pmc_" + what.split()[1])
    cmd = ["rocprofv3"] + what.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py")] + \
          bench_args.split() + ["--no-cpu-baseline", "--no-roofline", "--no-per-call", "--no-sub", "--no-sustained"]      # the main pass only: its last launches are the timed steps
    # one launch at a time: in the timed pass of bench.py consecutive steps overlap on the device, which stretches every
    # launch in a trace; the roofline is the kernel's own (bench.py's event pass runs serially too)
    sh(cmd, d + ".log", env={"PTMI355_OVERLAP": "0"})
    return d


def per_kernel_rows(d, suffix):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main(which):
    os.makedirs(OUT, exist_ok=True)
    sha = build_digest()
    costs = os.path.join(OUT, "costs_%s.json" % ROUND)
    prev = os.path.join(ROOT, "profiles", "r03", "costs_r03.json")
    if not os.path.exists(costs) and os.path.exists(prev) and not os.environ.get("PT_REMEASURE_COSTS"):
        # the per-opcode issue costs are a property of the chip, measured in round 3 (profiles/r03/issue_ops_r03.json, one row per
        # opcode of the library's device assembly); an opcode that is new since then shows up as `unpriced` in the histograms
        import shutil
        shutil.copy(prev, costs)
    if not os.path.exists(costs):
        mb = os.path.join(ROOT, "profiles", "microbench")
        subprocess.run("python3 gen_issue_ops.py > issue_ops.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_ops issue_ops.hip",
                       shell=True, cwd=mb, check=True)
        with open(os.path.join(OUT, "issue_ops_%s.json" % ROUND), "w") as f:
            subprocess.run(["/tmp/issue_ops", "1024"], stdout=f, check=True, timeout=900)
        subprocess.run(["python3", os.path.join(ROOT, "profiles", "tools", "issue_costs.py"), os.path.join(OUT, "issue_ops_%s.json" % ROUND), costs], check=True)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        traffic = json.load(open(tpath))
        if "configs" not in traffic:
            traffic = {"configs": {}}
    except Exception:
        traffic = {"configs": {}}
    for tag in which:
        bench_args, (ccfg, cflags, csteps, cbatch), kernels = CONFIGS[tag]
        os.makedirs(os.path.join(OUT, tag), exist_ok=True)
        print("==", tag, flush=True)
        dtrace = rocprof(tag, "--kernel-trace --stats", bench_args)
        dsq = rocprof(tag, "--pmc " + SQ_PASS, bench_args)
        dfe = rocprof(tag, "--pmc FETCH_SIZE", bench_args)
        dwr = rocprof(tag, "--pmc WRITE_SIZE", bench_args)
        # the TIMED launches only: bench.py's warm-up steps run while the clocks still ramp (the first step of a process
        # is ~15 % slower than the twentieth) and are not in its ms_per_step either
        m = re.search(r"--steps (\d+)", bench_args)
        timed_steps = int(m.group(1)) if m else 20
        rows = sorted(per_kernel_rows(dtrace, "kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
        stage_calls = {}
        for (cb, stage, per_step) in kernels:
            stage_calls[PROF_NAME[cb]] = timed_steps * per_step
        dur = {}
        for r in rows:
            dur.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k in list(dur):
            for name, n in stage_calls.items():
                if name in k:
                    dur[k] = dur[k][-n:]
        pmc = {}
        for d in (dsq, dfe, dwr):
            for r in per_kernel_rows(d, "counter_collection.csv"):
                a = pmc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
        total_ns = sum(sum(v) for v in dur.values())
        lines = ["%-64s %7s %12s %10s %7s" % ("kernel (rocprofv3 --kernel-trace --stats; timed steps, PTMI355_OVERLAP=0)", "calls", "total_us", "avg_us", "share")]
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            lines.append("%-64s %7d %12.1f %10.2f %6.1f%%" % (re.sub(r"^void |\(anonymous namespace\)::", "", k)[:64], len(v), sum(v) / 1e3,
                                                            sum(v) / len(v) / 1e3, 100.0 * sum(v) / max(1, total_ns)))
        out = {"config": tag, "bench_args": bench_args, "build_sha16": sha, "kernels": {}}
        for (cb, stage, per_step) in kernels:
            name = PROF_NAME[cb]
            full = [k for k in dur if name in k]
            if not full:
                lines.append("!! kernel %s not in the trace" % name)
                continue
            full = full[0]
            avg_us = sum(dur[full]) / len(dur[full]) / 1e3
            counted = os.path.join(ROOT, ".ab", "cnt_" + cb)
            u32 = os.path.join(OUT, tag, cb + ".u32")
            crun = os.path.join(OUT, tag, cb + ".count_run.json")
            rc = 1
            if os.path.exists(os.path.join(counted, "libptmi355.so")):
                with open(crun, "w") as f:
                    rc = subprocess.run(["python3", os.path.join(ROOT, "profiles", "tools", "count_run.py"), ccfg, cflags,
                                         os.path.join(counted, "map.json"), u32, str(csteps), str(cbatch)], cwd=ROOT, stdout=f,
                                        stderr=subprocess.DEVNULL, env=dict(os.environ, PTMI355_LIB=os.path.join(counted, "libptmi355.so")),
                                        timeout=900).returncode
            hist = None
            if rc == 0:
                launches = csteps * per_step
                p = subprocess.run(["python3", os.path.join(ROOT, "profiles", "tools", "isa_count.py"), "hist", os.path.join(counted, "map.json"),
                                    u32, costs, str(launches)], capture_output=True, text=True)
                if p.returncode == 0:
                    open(os.path.join(OUT, tag, cb + ".hist.txt"), "w").write(p.stdout)
                    hist = json.loads(p.stdout.strip().splitlines()[-1])
            c = {k: v[0] / v[1] for k, v in pmc.get(full, {}).items()}
            # the counted run may use another batch than the bench line: scale the histogram by the rays per launch
            e = {"stage": stage, "launches_per_step": per_step, "avg_us": round(avg_us, 2), "calls": len(dur[full]), "pmc_per_launch": c}
            if hist:
                cj = json.load(open(crun))
                scale = 1.0
                if c.get("SQ_INSTS_VALU"):
                    scale = c["SQ_INSTS_VALU"] / hist["valu_per_launch"]       # bench batch / counted batch (and the cross-check when they are equal)
                e.update({"valu_insts_per_launch_counted": hist["valu_per_launch"], "counted_run": {k: cj[k] for k in ("steps", "batch", "image_md5")},
                          "sq_insts_valu_over_counted": round(scale, 4),
                          "issue_cycles_per_launch": hist["issue_cycles_per_launch"] * scale,
                          "issue_cycles_guide_rates_per_launch": hist.get("issue_cycles_guide_rates_per_launch", 0.0) * scale,
                          "cycles_per_valu_inst": hist["issue_cycles_per_launch"] / hist["valu_per_launch"],
                          "unpriced_share_of_cycles": hist["unpriced_share_of_cycles"],
                          "flops_fp32_per_launch_64_lanes": hist["flops_fp32_per_launch"] * scale,
                          "mfma_flops_per_launch": hist.get("mfma_flops_per_launch", 0.0) * scale,
                          "wave_insts_per_launch": {k: v * scale for k, v in hist["wave_insts_per_launch"].items()}})
                avail = SIMDS * PEAK_GHZ * 1e3 * avg_us
                e["issue_frac_of_peak_clock"] = round(e["issue_cycles_per_launch"] / avail, 4)
            if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
                e["active_lane_fraction"] = round(c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
            if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_BUSY_CYCLES"):
                # quad-cycles of waves executing VALU, summed over the SIMDs, against the SIMD-cycles of the launch
                e["valu_busy_pmc"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * PEAK_GHZ * 1e3 * avg_us), 4)
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                e["hbm_bytes_per_launch"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
                e["hbm_frac"] = round(e["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 8e12, 4)
            out["kernels"][cb] = dict(e, name=re.sub(r"^void |\(anonymous namespace\)::", "", full))
            lines.append("%s  %s: %.1f us/launch; VALU insts/launch counted %.4g (SQ_INSTS_VALU / counted = %s); issue cycles %.4g = %s of %d SIMDs x %.1f GHz; "
                         "unpriced %.2f %%; active lanes %s; HBM %s of 8 TB/s" % (
                             cb, out["kernels"][cb]["name"][:50], avg_us, e.get("valu_insts_per_launch_counted", 0), e.get("sq_insts_valu_over_counted"),
                             e.get("issue_cycles_per_launch", 0), e.get("issue_frac_of_peak_clock"), SIMDS, PEAK_GHZ,
                             100.0 * e.get("unpriced_share_of_cycles", 0), e.get("active_lane_fraction"), e.get("hbm_frac")))
        open(os.path.join(OUT, "rocprof_%s_%s_summary.txt" % (ROUND, tag)), "w").write("\n".join(lines) + "\n")
        json.dump(out, open(os.path.join(OUT, "roofline_%s_%s.json" % (ROUND, tag)), "w"), indent=1)
        traffic["configs"][traffic_key(tag)] = out
        json.dump(traffic, open(tpath, "w"), indent=1)
        print("\n".join(lines), flush=True)
        # the plain bench line, with the roofline object this profile feeds
        with open(os.path.join(OUT, "bench_%s_%s.json" % (ROUND, tag)), "w") as f:
            subprocess.run(["python3", os.path.join(ROOT, "bench.py")] + bench_args.split() + ["--no-cpu-baseline"], cwd=ROOT, stdout=f,
                           stderr=subprocess.DEVNULL, timeout=900)


def traffic_key(tag):
    bench_args, (ccfg, _, _, _), _ = CONFIGS[tag]
    m = re.search(r"--flags (\S+)", bench_args)
    b = re.search(r"--batch (\d+)", bench_args)
    return "%s|%s|%s" % (ccfg, m.group(1) if m else "compact", b.group(1) if b else "64")


def assemble(src):
    """profiles/traffic.json from the roofline_<round>_*.json of a finished collection (gpurun merges gpurun_out/ back, not profiles/)"""
    traffic = {"configs": {}}
    for tag in CONFIGS:
        f = os.path.join(src, "roofline_%s_%s.json" % (ROUND, tag))
        if os.path.exists(f):
            traffic["configs"][traffic_key(tag)] = json.load(open(f))
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print("profiles/traffic.json:", ", ".join(sorted(traffic["configs"])))


if __name__ == "__main__":
    if sys.argv[1:2] == ["--assemble"]:
        assemble(sys.argv[2])
    else:
        main(sys.argv[1:] or list(CONFIGS))
