#!/usr/bin/env python3
"""soak_host_image.py [calls]: the launch-written host image over a long run of pathtrace() calls at 800x800 -- synchronous
(PT_PIN_IMAGE) and PT_ASYNC_IMAGE -- the host buffer compared with the device's running sum every 97 calls and the final
sums with the every-pixel-every-call plan (PT_PIN_IMAGE without PT_HOST_SPARSE)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
pt = ge.load_package(); L = pt.library()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell__%s" % k]
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
n = 800 * 800


def run(flags, env):
    for k, v in env.items():
        os.environ[k] = v
    pt.pathtraceInit(scene, flags=flags, pin_image=False)
    host = np.full((n, 3), -1.0, dtype=np.float32)
    bad = 0
    for it in range(1, N + 1):
        assert L.pt_trace(None, 0, it, host.ctypes.data) == 0
        if it % 97 == 0:
            dev = pt.get_image(n)                        # synchronises: an asynchronous buffer is complete after it
            bad += int((dev.view(np.uint32) != host.view(np.uint32)).sum())
    pt.synchronize()
    md5 = hashlib.md5(host.tobytes()).hexdigest()
    same = host.tobytes() == pt.get_image(n).tobytes()
    pt.pathtraceFree()
    for k in env:
        os.environ.pop(k)
    return md5, bad, same


ref = run(pt.PT_COMPACT | pt.PT_PIN_IMAGE, {})
for name, flags in (("synchronous", pt.PT_COMPACT | pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE), ("PT_ASYNC_IMAGE", pt.PT_COMPACT | pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE | pt.PT_ASYNC_IMAGE)):
    got = run(flags, {})
    print("%-15s %d calls: host == device at every check: %s, at the end: %s, final sums == every-pixel plan: %s" % (name, N, got[1] == 0, got[2], got[0] == ref[0]))
    assert got[1] == 0 and got[2] and got[0] == ref[0]
print("ok")
