#!/usr/bin/env python3
"""percall_loop.py MODE [N]: N pathtrace() calls at 800x800 (C2), one iteration per call -- the reference's calling
pattern (src/main.cpp:130-140) -- for a kernel trace or a wall-clock breakdown.
MODE: sync (host image, synchronous), async (PT_ASYNC_IMAGE), none (no host image, synchronous), enq (pt_trace_batch_async)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
pt = ge.load_package()
L = pt.library()
scene = bench.load_scene(pt, "cornell")
host = np.zeros((800 * 800, 3), dtype=np.float32)
flags = pt.PT_COMPACT | (pt.PT_ASYNC_IMAGE if mode == "async" else 0)
pt.pathtraceInit(scene, flags=flags, max_batch=1)
call = {"sync": lambda it: L.pt_trace(None, 0, it, host.ctypes.data), "async": lambda it: L.pt_trace(None, 0, it, host.ctypes.data),
        "none": lambda it: L.pt_trace(None, 0, it, None), "enq": lambda it: pt.trace_batch_async(it, 1)}[mode]
for k in range(20):
    call(1 + k)
pt.synchronize()
r0 = pt.total_rays()
t0 = time.perf_counter()
for k in range(n):
    call(21 + k)
pt.synchronize()
el = time.perf_counter() - t0
print("%s: %d calls, %.1f us per call, %.1f Mrays/s" % (mode, n, el / n * 1e6, (pt.total_rays() - r0) / el / 1e6))
pt.pathtraceFree()
