#!/usr/bin/env python3
"""trace_gaps.py KERNEL_TRACE.csv [skip]: per kernel name the mean duration, and for the whole timeline the mean gap
between the end of one kernel and the start of the next -- what the device idles between the launches of a call loop."""
import csv
import collections
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 5
rows = rows[skip:]
dur = collections.defaultdict(list)
for s, e, n in rows:
    dur[n.replace("(anonymous namespace)::", "")[:60]].append((e - s) / 1e3)
span = (rows[-1][1] - rows[0][0]) / 1e3
busy = 0.0
cur_end = rows[0][0]
for s, e, n in rows:
    if e > cur_end:
        busy += (e - max(s, cur_end)) / 1e3
        cur_end = e
print("timeline %.0f us, some kernel running %.0f us (%.1f %%), idle %.0f us" % (span, busy, 100 * busy / span, span - busy))
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s calls %5d  mean %8.1f us  total %9.0f us" % (n, len(v), sum(v) / len(v), sum(v)))
