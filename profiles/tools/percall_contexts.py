#!/usr/bin/env python3
"""percall_contexts.py: pathtrace() per call with the page-locked host image (800x800), one context and several contexts on
this one GPU, with and without the exchange-free path (PTMI355_MULTI_DIRECT)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pt = ge.load_package()
if not pt.has_experiments():
    raise SystemExit("percall_contexts.py flips experiment variables (PTMI355_MULTI_DIRECT, ...): load a -DPT_EXPERIMENTS build through PTMI355_LIB"); L = pt.library()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell__%s" % k]
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
n = 800 * 800
for name, kw, env in (("one context", {}, {}), ("two contexts, exchange-free", dict(devices=[0, 0], tile=(0, 1, 8)), {}),
                      ("two contexts, exchange + frame copy", dict(devices=[0, 0], tile=(0, 1, 8)), {"PTMI355_MULTI_DIRECT": "0"}),
                      ("four contexts, exchange-free", dict(devices=[0, 0, 0, 0], tile=(0, 1, 8)), {}),
                      ("four contexts, exchange + frame copy", dict(devices=[0, 0, 0, 0], tile=(0, 1, 8)), {"PTMI355_MULTI_DIRECT": "0"})):
    for k, v in env.items():
        os.environ[k] = v
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_PIN_IMAGE, pin_image=False, **kw)
    host = np.zeros((n, 3), dtype=np.float32)
    for it in range(1, 33):
        assert L.pt_trace(None, 0, it, host.ctypes.data) == 0
    r0 = pt.counters()[0]
    t0 = time.perf_counter()
    N = 256
    for it in range(33, 33 + N):
        assert L.pt_trace(None, 0, it, host.ctypes.data) == 0
    dt = time.perf_counter() - t0
    rays = pt.counters()[0] - r0
    ok = host.tobytes() == pt.get_image(n).tobytes()
    pt.pathtraceFree()
    for k in env:
        os.environ.pop(k)
    print("%-40s %.4f ms per call  %8.1f Mrays/s  host == device: %s" % (name, dt / N * 1e3, rays / dt / 1e6, ok))
