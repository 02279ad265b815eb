#!/usr/bin/env python3
"""Time per iteration of the one-launch plan (k_iteration) against the trace depth and the batch size: what one more
bounce costs at 1 spp.  usage: iter_depth.py [spp ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pt = ge.load_package()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell__" + k]
spps = [int(v) for v in sys.argv[1:]] or [1]
for spp in spps:
    for depth in (1, 2, 3, 4, 6, 8):
        scene = pt.Scene(g("geoms"), g("materials"), g("camera"), depth)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=spp)
        for k in range(20):
            pt.trace_batch_async(1 + k * spp, spp)
        pt.synchronize()
        r0 = pt.total_rays()
        t0 = time.perf_counter()
        n = 200
        for k in range(n):
            pt.trace_batch_async(100 + k * spp, spp)
        pt.synchronize()
        dt = time.perf_counter() - t0
        rays = pt.total_rays() - r0
        print("spp %2d depth %d: %7.1f us per batch, %6.2f Grays/s, %8d rays per batch" % (spp, depth, dt / n * 1e6, rays / dt / 1e9, rays // n), flush=True)
        pt.pathtraceFree()
