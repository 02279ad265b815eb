"""Per-call latency of pathtrace() under PT_LOOKAHEAD by position inside a 64-iteration window (800x800 Cornell):
the first calls of a window run beside the tracing of the next one, the last ones on an otherwise idle device.
usage: python profiles/tools/lookahead_latency.py [host|nohost|hostpbo|pbo] [windows]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pt = ge.load_package()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
scene = pt.Scene(z["cornell__geoms"], z["cornell__materials"], z["cornell__camera"], int(z["cornell__depth"]))
mode = sys.argv[1] if len(sys.argv) > 1 else "host"
K = int(os.environ.get("LA_K", "64"))
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = pt.library()
n = 800 * 800
host = np.zeros((n, 3), dtype=np.float32)
if os.environ.get("LA_NUMA"):
    # the host image on one NUMA node (mbind before the first touch): do the calls' PCIe writes care how far it is from the GPU?
    import ctypes, mmap
    node = int(os.environ["LA_NUMA"])
    size = ((n * 12 + 4095) // 4096) * 4096
    m = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc = ctypes.CDLL(None, use_errno=True)
    mask = ctypes.c_ulong(1 << node)
    rc = libc.syscall(237, ctypes.c_void_p(base), ctypes.c_ulong(size), 2, ctypes.byref(mask), ctypes.c_ulong(64), 0)      # mbind(MPOL_BIND)
    host = np.frombuffer(m, dtype=np.float32, count=n * 3).reshape(n, 3)
    host[:] = 0.0
    print("host image bound to NUMA node %d: mbind rc %d errno %d" % (node, rc, ctypes.get_errno()))
elif os.environ.get("LA_HUGE"):
    # the host image on transparent huge pages (2 MiB-aligned anonymous mapping + MADV_HUGEPAGE): do the device's scattered
    # 12-byte writes into it cost less when the frame is four pages instead of 1875?
    import ctypes, mmap
    size = ((n * 12 + (2 << 20) - 1) // (2 << 20) + 1) * (2 << 20)
    m = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    off = (-base) % (2 << 20)
    libc = ctypes.CDLL(None, use_errno=True)
    rc = libc.madvise(ctypes.c_void_p(base + off), ctypes.c_size_t(size - (2 << 20)), 14)
    host = np.frombuffer(m, dtype=np.float32, count=n * 3, offset=off).reshape(n, 3)
    host[:] = 0.0
    thp = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip() if os.path.exists("/sys/kernel/mm/transparent_hugepage/enabled") else "?"
    anon = [l for l in open("/proc/self/smaps_rollup") if "AnonHugePages" in l]
    print("huge pages: madvise rc %d, THP %s, %s" % (rc, thp, anon[0].strip() if anon else "?"))
flags = pt.PT_COMPACT | pt.PT_LOOKAHEAD | (pt.PT_PIN_IMAGE | pt.PT_HOST_SPARSE if mode in ("host", "hostpbo") else 0)
pt.pathtraceInit(scene, flags=flags, max_batch=K, pin_image=False)
buf = host.ctypes.data if mode in ("host", "hostpbo") else None
pbo = None
if mode in ("hostpbo", "pbo"):                # the reference's GL host hands over a PBO every call (main.cpp:131-137)
    import torch
    pbo_t = torch.zeros(n * 4, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    pbo = pbo_t.data_ptr()
for it in range(1, 85):
    L.pt_trace(pbo, 0, it, buf)
lat = np.zeros((windows, 64))
t_all = time.perf_counter()
for w in range(windows):
    for k in range(64):
        t0 = time.perf_counter()
        L.pt_trace(pbo, 0, 85 + 64 * w + k, buf)
        lat[w, k] = time.perf_counter() - t0
el = time.perf_counter() - t_all
pt.pathtraceFree()
us = lat * 1e6
print("mode %s: %.1f us per call over %d calls" % (mode, el / (windows * 64) * 1e6, windows * 64))
print("call 0 of a window (enqueues the next window): median %.1f us, mean %.1f; the other calls: mean %.1f, max %.1f us" %
      (np.median(us[:, 0]), us[:, 0].mean(), us[:, 1:].mean(), us[:, 1:].max()))
for a, b in ((1, 8), (8, 16), (16, 24), (24, 32), (32, 40), (40, 48), (48, 56), (56, 64)):
    print("calls %2d-%2d: median %.1f  p10 %.1f  p90 %.1f us" % (a, b - 1, np.median(us[:, a:b]), np.percentile(us[:, a:b], 10), np.percentile(us[:, a:b], 90)))
