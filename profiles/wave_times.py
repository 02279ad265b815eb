#!/usr/bin/env python3
"""When do the waves of k_bounce end?  Diagnostic for the persistent grid's load balance.

Builds two variants of the library with -DPT_WAVE_TIMES (every wave records s_memrealtime at its start and end, and
its hardware slot) -- one with the arbiter's own order (-DPT_NO_ROTATE_PRIO), one as shipped -- traces three 64-spp
batches of C2 with each and prints, per bounce: the launch's duration, the mean residency of a wave (its lifetime over
the launch's duration) and the mean end time by the wave's slot on its SIMD.
Usage (GPU box): python profiles/wave_times.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "project3-cuda-path-tracer_amd")


def build(name, extra):
    out = os.path.join(ROOT, "gpurun_out", "wave_times", name)
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libptmi355.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                    "-shared", "-std=c++17", "-DPT_WAVE_TIMES"] + extra + ["-o", lib, os.path.join(PKG, "csrc", "ptmi355.hip")],
                   check=True, cwd=PKG, stderr=subprocess.DEVNULL)
    return lib


def measure(lib):
    code = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
pt = ge.load_package(); L = pt.library()
z = np.load(os.path.join(%r, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell__%%s" %% k]
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=64)
for it in range(3):
    pt.trace_batch_async(1 + 64 * it, 64)
pt.synchronize()
t = np.zeros((8, 8192, 2), dtype=np.uint64); x = np.zeros((8, 8192), dtype=np.uint32)
assert L.ptdbg_wave_times(t.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)) == 0
W = 5120
for d in range(8):
    t0 = t[d, :W, 0].astype(np.int64); t1 = t[d, :W, 1].astype(np.int64)
    s0 = t0.min(); dur = t1.max() - s0
    e = (t1 - s0) / dur
    slot = (x[d, :W] >> 4) & 0xf
    print("  bounce %%d: launch %%5.0f us, waves start within %%.1f us, mean residency %%.2f, mean end by slot: %%s" %% (
        d, dur / 100.0, (t0.max() - s0) / 100.0, ((t1 - t0) / dur).mean(),
        " ".join("%%d:%%.2f" %% (k, e[slot == k].mean()) for k in sorted(set(slot.tolist())))))
pt.pathtraceFree()
''' % (ROOT, ROOT)
    env = dict(os.environ, PTMI355_LIB=lib)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    sys.stdout.write(p.stdout)
    if p.returncode:
        sys.stdout.write(p.stderr[-2000:])


if __name__ == "__main__":
    print("oldest-first (the arbiter's own order, -DPT_NO_ROTATE_PRIO):")
    measure(build("fifo", ["-DPT_NO_ROTATE_PRIO"]))
    print("user priority rotated with the tile counter (as shipped):")
    measure(build("rotate", []))
