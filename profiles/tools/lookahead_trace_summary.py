"""Summarise a rocprofv3 --kernel-trace CSV of profiles/tools/lookahead_latency.py: the per-call gathers (duration, spacing)
and the windows traced beside them (every bounce launch's duration, the gaps between a window's launches, a window's span).
usage: python profiles/tools/lookahead_trace_summary.py <kernel_trace.csv>"""
import csv
import re
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))


def short(n):
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:40]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"], int(r["Grid_Size_X"])) for r in rows)
g = [e for e in ev if e[2] in ("k_gather_one", "k_la_apply", "k_gather_bits")]
d = np.array([(e[1] - e[0]) / 1e3 for e in g])
gap = np.array([(g[i + 1][0] - g[i][1]) / 1e3 for i in range(len(g) - 1)])
print("per-call kernel (k_la_apply / k_gather_one): %d launches, duration median %.1f p90 %.1f max %.1f us; idle between two: median %.1f p90 %.1f us" %
      (len(g), np.median(d), np.percentile(d, 90), d.max(), np.median(gap), np.percentile(gap, 90)))
cp = [e for e in ev if e[2] == "k_la_compact"]
if cp:
    print("k_la_compact: %d launches, median %.1f us (largest grids: %.1f us)" % (len(cp), np.median([(e[1] - e[0]) / 1e3 for e in cp]), max((e[1] - e[0]) / 1e3 for e in cp)))
b = [e for e in ev if e[2] == "k_bounce"]
# a window's launches follow each other on ONE queue without a gap (the last eight windows' worth are printed); two windows
# on the two lanes' queues may overlap in time
byq = {}
for e in b:
    byq.setdefault(e[3], []).append(e)
wins = []
for q, es in byq.items():
    cur = []
    for e in es:
        if cur and e[0] - cur[-1][1] > 20000:
            wins.append(cur); cur = []
        cur.append(e)
    if cur:
        wins.append(cur)
wins.sort(key=lambda w: w[0][0])
for w in wins[-8:]:
    durs = [(e[1] - e[0]) / 1e3 for e in w]
    others = [x for x in b if x[3] != w[0][3] and x[1] > w[0][0] and x[0] < w[-1][1]]
    inside = [x for x in g if x[0] >= w[0][0] and x[1] <= w[-1][1]]
    print("window on queue %s: %d launches, span %.0f us, sum of launches %.0f: %s; %d gathers inside; overlaps %d launches of the other lane" %
          (w[0][3], len(w), (w[-1][1] - w[0][0]) / 1e3, sum(durs), " ".join("%.0f" % x for x in durs), len(inside), len(others)))
