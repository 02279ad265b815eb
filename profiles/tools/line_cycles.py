#!/usr/bin/env python3
"""Issue cycles of one instrumented kernel by SOURCE LINE:
    line_cycles.py plain_g.s KERNEL_SUBSTR map.json counts.u32 costs.json [launches] [top]
plain_g.s = the device assembly of the same sources and flags as the counted build, plus -gline-tables-only (the .loc
directives; debug line tables do not change code generation: the script checks that the kernel's blocks are the ones of
map.json, opcode for opcode).  Every executed instruction's cycles (vector: the microbenchmark table; scalar ALU / branch /
scalar load: 4.42 per SIMD, waits and nops 1.2 -- a separate pipe, listed separately) go to the innermost source line of
its .loc (the inlined-at chain is not followed)."""
import collections
import json
import re
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__file__))
import isa_count as ic  # noqa: E402

path, substr, map_path, counts_path, cost_path = sys.argv[1:6]
launches = int(sys.argv[6]) if len(sys.argv) > 6 else 1
top = int(sys.argv[7]) if len(sys.argv) > 7 else 60
lines = open(path).read().split('\n')
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
a, b, name = ic.kernel_span(lines, substr)
body = lines[a + 1:b]
blocks = ic.partition(body)
m = json.load(open(map_path))
if [[op for (_, op, _) in ins] for (_, ins) in blocks] != m["blocks"]:
    raise SystemExit("the blocks of %s differ from map.json's: not the same build" % path)
c = np.fromfile(counts_path, dtype=np.uint32)
costs = json.load(open(cost_path))
per, dflt = costs["cycles"], costs["default"]
loc_at = {}
cur = (0, 0)
for i, l in enumerate(body):
    mm = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
    if mm:
        cur = (int(mm.group(1)), int(mm.group(2)))
    loc_at[i] = cur
vec = collections.Counter(); sca = collections.Counter(); nv = collections.Counter(); ns = collections.Counter()
ops = collections.defaultdict(collections.Counter)
for bid, (first, ins) in enumerate(blocks):
    slot = (bid // m["lanes"]) * 64 + ic.slot_lane(bid % m["lanes"])
    n = int(c[slot])
    if not n:
        continue
    for (i, op, rest) in ins:
        key = loc_at[i]
        if op.startswith("v_"):
            vec[key] += n * per.get(op if op in per else re.sub(r"_(e32|e64|dpp|sdwa)$", "", op), dflt)
            nv[key] += n
            ops[key][op] += n
        elif op.startswith("s_"):
            sca[key] += n * (1.2 if op in ("s_waitcnt", "s_nop") or op.startswith("s_waitcnt") else 4.42)
            ns[key] += n
tv, ts = sum(vec.values()), sum(sca.values())
print("%s: vector %.4g, scalar %.4g issue cycles per launch" % (name, tv / launches, ts / launches))
print("%-28s %8s %6s %9s %8s %6s  top vector opcodes" % ("source line", "vec Mcyc", "%", "vec insts", "sca Mcyc", "%"))
for key in sorted(set(vec) | set(sca), key=lambda k: -(vec[k] + 0.5 * sca[k]))[:top]:
    f = "%s:%d" % (files.get(key[0], "?"), key[1])
    print("%-28s %8.2f %6.2f %9.2f %8.2f %6.2f  %s" % (f, vec[key] / launches / 1e6, 100 * vec[key] / tv, nv[key] / launches / 1e6,
          sca[key] / launches / 1e6, 100 * sca[key] / ts, " ".join("%s:%.1f" % (o, k / launches / 1e6) for o, k in ops[key].most_common(4))))
