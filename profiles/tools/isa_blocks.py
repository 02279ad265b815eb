#!/usr/bin/env python3
"""Per basic block of one kernel's ISA (compiled with -gline-tables-only): instruction counts and source lines.
usage: isa_blocks.py full.s kernel.s [min_insts]"""
import re, sys, collections
files = {}
for l in open(sys.argv[1]):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m:
        files[int(m.group(1))] = m.group(3).replace('.hpp', '').replace('pt_', '')
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cur = (0, 0)
blocks = []
b = None
def newblock(name):
    global b
    b = {"name": name, "valu": 0, "salu": 0, "nop": 0, "mem": 0, "lds": 0, "rl": 0, "lines": collections.Counter(), "term": ""}
    blocks.append(b)
newblock("entry")
for l in open(sys.argv[2]):
    m = re.match(r'(\.LBB\d+_\d+):', l)
    if m:
        newblock(m.group(1)); continue
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)\s*(.*)', l)
    if not m: continue
    op = m.group(1)
    if op == 's_nop': b["nop"] += 1
    elif op.startswith('v_'):
        b["valu"] += 1
        if op in ('v_readlane_b32', 'v_writelane_b32'): b["rl"] += 1
        b["lines"]["%s:%d" % (files.get(cur[0], '?'), cur[1])] += 1
    elif op.startswith('s_'):
        b["salu"] += 1
        if op.startswith('s_cbranch') or op == 's_branch': b["term"] += " " + op.replace('s_cbranch_', 'c_').replace('s_branch', 'br') + ">" + m.group(2).split()[0].replace('.LBB', '')
    elif op.startswith('ds_'): b["lds"] += 1
    else: b["mem"] += 1
for k, b in enumerate(blocks):
    if b["valu"] + b["salu"] + b["mem"] + b["lds"] < minn: continue
    top = " ".join("%s(%d)" % kv for kv in b["lines"].most_common(5))
    print("%-10s v%4d rl%3d s%4d nop%3d m%3d l%3d |%s | %s" % (b["name"].replace('.LBB', ''), b["valu"], b["rl"], b["salu"], b["nop"], b["mem"], b["lds"], b["term"], top))
