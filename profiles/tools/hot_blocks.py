#!/usr/bin/env python3
"""The basic blocks of an instrumented kernel by issue cycles:  hot_blocks.py map.json counts.u32 costs.json [launches] [top]"""
import json
import re
import sys

import numpy as np

m = json.load(open(sys.argv[1]))
c = np.fromfile(sys.argv[2], dtype=np.uint32)
costs = json.load(open(sys.argv[3]))
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 1
top = int(sys.argv[5]) if len(sys.argv) > 5 else 40
per, dflt = costs["cycles"], costs["default"]


def lane(i):
    return i if i < 31 else i + 1


rows, tot = [], 0.0
for bid, blk in enumerate(m["blocks"]):
    n = int(c[(bid // m["lanes"]) * 64 + lane(bid % m["lanes"])])
    cyc = sum(per.get(op if op in per else re.sub(r"_(e32|e64|dpp|sdwa)$", "", op), dflt) for op in blk if op.startswith("v_"))
    rows.append((n * cyc, bid, n, len(blk), sum(op.startswith("v_") for op in blk)))
    tot += n * cyc
print("%s: %.4g issue cycles per launch" % (m["kernel"], tot / launches))
acc = 0.0
for cyc, bid, n, ni, nv in sorted(rows, reverse=True)[:top]:
    acc += cyc
    print("blk %4d  execs/launch %10.0f  insts %3d  valu %3d  share %5.1f%%  cum %5.1f%%   %s" % (
        bid, n / launches, ni, nv, 100 * cyc / tot, 100 * acc / tot, " ".join(m["blocks"][bid][:7])))
