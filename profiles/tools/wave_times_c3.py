import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PTMI355_LIB"] = os.path.join(ROOT, ".ab", "wclk", "libptmi355.so")
os.environ["PTMI355_OVERLAP"] = "0"
import __graft_entry__ as ge
pt = ge.load_package(); L = pt.library()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell_glass__%s" % k]
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
for name, flags in (("unsorted", pt.PT_COMPACT), ("sorted", pt.PT_COMPACT | pt.PT_SORT_MATERIAL)):
    pt.pathtraceInit(scene, flags=flags, max_batch=64)
    for it in range(3):
        pt.trace_batch_async(1 + 64 * it, 64)
    pt.synchronize()
    t = np.zeros((8, 8192, 2), dtype=np.uint64); x = np.zeros((8, 8192), dtype=np.uint32)
    assert L.ptdbg_wave_times(t.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)) == 0
    print("==", name)
    for d in range(8):
        W = int((t[d, :, 1] > 0).sum())
        if W == 0: continue
        t0 = t[d, :W, 0].astype(np.int64); t1 = t[d, :W, 1].astype(np.int64)
        s0 = t0.min(); dur = t1.max() - s0
        end = (t1 - s0) / dur
        life = (t1 - t0).mean() / dur
        # end time by position of the wave's run in the pool (deciles of the wave index)
        dec = [end[int(W * k / 10):int(W * (k + 1) / 10)].mean() for k in range(10)]
        print("bounce %d: %5d waves, %4.0f us, mean life %.2f of the launch; mean end by decile of the run index: %s" % (d, W, dur / 100.0, life, " ".join("%.2f" % v for v in dec)))
    pt.pathtraceFree()
