#!/usr/bin/env python3
"""Phase stamps of one bounce kernel launch (a -DPT_STAMPS=<depth> build prints them to stderr after every synchronous
batch): usage  PTMI355_LIB=.ab/<stamps build>/libptmi355.so stamps.py CONFIG FLAGS [BATCH]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pt = ge.load_package()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
name = {"c2": "cornell", "c3": "cornell_glass"}[sys.argv[1]]
g = lambda k: z[name + "__" + k]
flags = 0
for f in sys.argv[2].split(","):
    flags |= {"compact": pt.PT_COMPACT, "sort": pt.PT_SORT_MATERIAL, "unfused": pt.PT_UNFUSED, "": 0}[f]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
pt.pathtraceInit(scene, flags=flags, max_batch=batch)
for k in range(3):
    pt.trace_batch(1 + k * batch, batch)
pt.pathtraceFree()
