#!/usr/bin/env python3
"""kernel_regs.py [ASM.s]: registers, spills, LDS and static instruction counts per kernel of a device assembly
(hipcc --cuda-device-only -S of csrc/ptmi355.hip; without an argument the assembly is made from the working tree into
.ab/plain_work.s).  A CPU-side check of what a kernel change did to occupancy before any GPU time is spent."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = "--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17".split()


def make_asm(extra=()):
    out = os.path.join(ROOT, ".ab", "plain_work.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc", "ptmi355.hip")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + list(extra) + ["--cuda-device-only", "-S", "-o", out, src], check=True)
    return out


def kernels(path):
    text = open(path).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
        rows.append((name, g("next_free_vgpr"), g("next_free_sgpr"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - |\Z)", text, re.S):
        pass
    # per-kernel metadata block (YAML): spill counts
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        blk = m.group(0)
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if nm:
            meta[nm.group(1)] = {k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1) if re.search(r"\.%s:\s+(\d+)" % k, blk) else "?"
                                 for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count")}
    # static instruction counts between the kernel's label and its s_endpgm-terminated body end
    counts = {}
    for name, *_ in rows:
        m = re.search(r"^%s:\n(.*?)^\s*\.section" % re.escape(name), text, re.S | re.M)
        if m:
            body = m.group(1)
            ins = [l.strip().split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
            counts[name] = (len(ins), sum(1 for i in ins if i.startswith("v_")), sum(1 for i in ins if i.startswith(("v_readlane", "v_writelane"))),
                            sum(1 for i in ins if i.startswith("scratch_")))
    return rows, meta, counts


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    path = args[0] if args else make_asm(extra)
    rows, meta, counts = kernels(path)
    print("%-64s %5s %5s %7s %7s %6s %6s %7s %6s %6s" % ("kernel", "vgpr", "sgpr", "sspill", "vspill", "insts", "valu", "rd/wrln", "scrtch", "lds"))
    for name, nv, ns, lds, priv in rows:
        if "k_" not in name:
            continue
        md = meta.get(name, {})
        c = counts.get(name, ("?",) * 4)
        short = name.replace("_ZN12_GLOBAL__N_1", "").replace("10BounceArgs", "")[:64]
        print("%-64s %5s %5s %7s %7s %6s %6s %7s %6s %6s" % (short, md.get("vgpr_count", nv), md.get("sgpr_count", ns), md.get("sgpr_spill_count", "?"),
                                                            md.get("vgpr_spill_count", "?"), c[0], c[1], c[2], c[3], lds))


if __name__ == "__main__":
    main()
