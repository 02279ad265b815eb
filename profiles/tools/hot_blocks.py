#!/usr/bin/env python3
"""The basic blocks of an instrumented kernel by issue cycles:  hot_blocks.py map.json counts.u32 costs.json [launches] [top] [lanes.u32]
lanes.u32 = the counts of a COUNT_MODE=lanes build of the same kernel and workload (sum of active lanes per block): adds the
blocks' average active lanes and ranks them by the issue cycles spent on switched-off lanes."""
import json
import re
import sys

import numpy as np

m = json.load(open(sys.argv[1]))
c = np.fromfile(sys.argv[2], dtype=np.uint32)
costs = json.load(open(sys.argv[3]))
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 1
top = int(sys.argv[5]) if len(sys.argv) > 5 else 40
lanes = np.fromfile(sys.argv[6], dtype=np.uint32) if len(sys.argv) > 6 else None
per, dflt = costs["cycles"], costs["default"]


def lane(i):
    return i if i < 31 else i + 1


rows, tot, idle_tot = [], 0.0, 0.0
for bid, blk in enumerate(m["blocks"]):
    slot = (bid // m["lanes"]) * 64 + lane(bid % m["lanes"])
    n = int(c[slot])
    cyc = sum(per.get(op if op in per else re.sub(r"_(e32|e64|dpp|sdwa)$", "", op), dflt) for op in blk if op.startswith("v_"))
    act = (int(lanes[slot]) / n / 64.0) if (lanes is not None and n) else None
    idle = n * cyc * (1.0 - act) if act is not None else 0.0
    rows.append((idle if lanes is not None else n * cyc, n * cyc, bid, n, len(blk), sum(op.startswith("v_") for op in blk), act))
    tot += n * cyc
    idle_tot += idle
print("%s: %.4g issue cycles per launch" % (m["kernel"], tot / launches))
if lanes is not None:
    print("issue cycles x switched-off lanes: %.1f %% of all issue cycles (active lanes, weighted by issue cycles: %.3f)" % (
        100 * idle_tot / tot, 1 - idle_tot / tot))
acc = 0.0
for key, cyc, bid, n, ni, nv, act in sorted(rows, reverse=True)[:top]:
    acc += key
    print("blk %4d  execs/launch %10.0f  insts %3d  valu %3d  share %5.1f%%  %s cum %5.1f%%   %s" % (
        bid, n / launches, ni, nv, 100 * cyc / tot, ("lanes %4.1f  idle share %4.1f%% " % (64 * act, 100 * key / tot)) if act is not None else "",
        100 * acc / (idle_tot if lanes is not None else tot), " ".join(m["blocks"][bid][:6])))
