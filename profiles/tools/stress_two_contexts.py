"""Repeat tests/test_multi_device.py::test_c2_frame_over_two_contexts' body (800x800, a batch of 8 with the host image + one
pathtrace(), one context against two contexts on device 0) and report every run whose images or ray counts differ.
usage: python profiles/tools/stress_two_contexts.py [runs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pt = ge.load_package()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
scene = pt.Scene(z["cornell__geoms"], z["cornell__materials"], z["cornell__camera"], int(z["cornell__depth"]))
n = 800 * 800
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(**kw):
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=8, **kw)
    img = np.zeros((n, 3), dtype=np.float32)
    pt.trace_batch(1, 8, img)
    one = pt.pathtrace(None, 0, 9).copy()
    rays = pt.total_rays()
    pt.pathtraceFree()
    return img, one, rays


want = run()
bad = 0
for k in range(runs):
    got = run(devices=[0, 0]) if k % 2 == 0 else run()
    for name, a, b in (("batch", got[0], want[0]), ("call", got[1], want[1])):
        d = np.flatnonzero((a.view(np.uint32) != b.view(np.uint32)).reshape(n, 3).any(axis=1))
        if d.size:
            bad += 1
            print("run %d (%s): %s image, %d pixels differ, rows %d..%d" % (k, "two contexts" if k % 2 == 0 else "one context", name, d.size, d[0] // 800, d[-1] // 800))
    if got[2] != want[2]:
        bad += 1
        print("run %d: rays %d against %d" % (k, got[2], want[2]))
print("%d runs, %d mismatches" % (runs, bad))
