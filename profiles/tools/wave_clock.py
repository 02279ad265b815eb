import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PTMI355_LIB"] = os.path.join(ROOT, ".ab", "wclk", "libptmi355.so")
import __graft_entry__ as ge
pt = ge.load_package(); L = pt.library()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell__%s" % k]
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=64)
for it in range(6):
    pt.trace_batch_async(1 + 64 * it, 64)
pt.synchronize()
t = np.zeros((8, 8192, 2), dtype=np.uint64); x = np.zeros((8, 8192), dtype=np.uint32)
assert L.ptdbg_wave_times(t.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)) == 0
for d in range(8):
    v = x[d][x[d] > 0].astype(np.float64) / 1000.0 * 0.1     # GHz
    print("bounce %d: %d waves, shader clock over a wave's life: median %.3f GHz, p10 %.3f, p90 %.3f" % (d, len(v), np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
pt.pathtraceFree()
