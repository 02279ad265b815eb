#!/usr/bin/env python3
"""trace_overlap.py KERNEL_TRACE.csv STEPS [LAUNCHES_PER_STEP]: the timed steps of a bench.py run in a rocprofv3 kernel
trace, when consecutive steps OVERLAP on the device (the default: csrc/ptmi355.hip, enqueue_batch_direct).

Under overlap a kernel's own duration says little (two launches share the chip, each takes longer) and the SUM of the
durations exceeds the wall time.  What can be held against bench.py's ms_per_step is the timeline: the span from the
first kernel of the last STEPS steps to the end of the last, and inside it the time during which SOME kernel of the
session ran (the union of the intervals) -- both per step.  VERDICT r04 item 4: `kernel time per step <= ms_per_step`
verified on a trace of the configuration that produced `value`, not explained."""
import collections
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 8
import re
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"^void ", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")))
              for r in csv.DictReader(open(path)))
ours = [r for r in rows if r[2].startswith("k_")]
bounces = [r for r in ours if r[2].startswith("k_bounce")]
if len(bounces) < steps * per_step:
    sys.exit("only %d k_bounce launches in the trace, %d wanted" % (len(bounces), steps * per_step))
t0 = bounces[-steps * per_step][0]
region = [r for r in ours if r[0] >= t0]
t1 = max(r[1] for r in region)
busy, cur = 0, t0
for s, e, _ in region:
    if e > cur:
        busy += e - max(s, cur)
        cur = e
dur = collections.defaultdict(list)
for s, e, n in region:
    dur[n[:66]].append((e - s) / 1e3)
total = sum(sum(v) for v in dur.values())
print("last %d steps (%d k_bounce launches each): span %.1f us = %.4f ms per step" % (steps, per_step, (t1 - t0) / 1e3, (t1 - t0) / 1e6 / steps))
print("  some kernel of the session running: %.1f us = %.4f ms per step (%.1f %% of the span)" % (busy / 1e3, busy / 1e6 / steps, 100.0 * busy / (t1 - t0)))
print("  sum of the kernels' own durations:   %.1f us = %.4f ms per step (%.2f x the span: launches overlap)" % (total, total / 1e3 / steps, total * 1e3 / (t1 - t0)))
print("%-66s %6s %10s %12s" % ("kernel (inside the span)", "calls", "mean us", "us per step"))
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-66s %6d %10.1f %12.1f" % (n, len(v), sum(v) / len(v), sum(v) / steps))
