#!/usr/bin/env python3
"""Attribute the instructions of one kernel's ISA (compiled with -gline-tables-only) to source lines.
usage: isa_lines.py kernel.s [file-substr]"""
import re, sys, collections
src = open(sys.argv[1]).read().splitlines()
files = {}
cur = (0, 0)
per_line = collections.Counter()
per_line_valu = collections.Counter()
ops = collections.defaultdict(collections.Counter)
for l in src:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)', l)
    if m:
        op = m.group(1)
        per_line[cur] += 1
        if op.startswith('v_'):
            per_line_valu[cur] += 1
        ops[cur][op] += 1
tot = sum(per_line.values()); totv = sum(per_line_valu.values())
print("total insts", tot, "valu", totv)
for (f, ln), c in sorted(per_line.items(), key=lambda kv: (files.get(kv[0][0], '?'), kv[0][1])):
    if c >= int(sys.argv[2]) if len(sys.argv) > 2 else 8:
        top = ", ".join("%s:%d" % kv for kv in ops[(f, ln)].most_common(4))
        print("%-18s %5d  all %4d valu %4d   %s" % (files.get(f, '?'), ln, c, per_line_valu[(f, ln)], top))
