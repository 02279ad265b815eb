#!/usr/bin/env python3
"""Whole-frame golden values for BASELINE configs[3] (C4: Cornell + the 100 032-triangle UV sphere, 800x800, depth 8,
naive loop over every triangle -- INSTRUCTION.md:123-128), iteration 1, from the ORACLE (oracle/ptoracle.c:
glm::intersectRayTriangle per triangle, external/include/glm/gtx/intersect.inl:37-74, plus the completion spec's
hit-point test).  2.5 * 10^11 ray-triangle tests: minutes on the build container's 8 cores, so it is run HERE, once,
and the result committed (tests/golden/c4_frame.npz); the GPU test then holds the HIP loop and the HIP hierarchy against
these values instead of against each other.

    python3 tests/golden/make_c4_golden.py [threads]

Stored: md5 of the float3 image, md5 of each 16-row strip (50: a mismatch names its strip), the per-bounce live counts,
the ray total, and 4096 sampled pixels (index + value) for a readable diff."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
    pt = ge.load_package()                       # (host-side mesh generator only: no GPU, no library call)
    po.build()
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    g = lambda k: z["cornell__" + k]
    tris = pt.meshes.uv_sphere()
    assert len(tris) == 100032
    geoms, tris, meshes = pt.meshes.add_mesh(g("geoms"), tris, material_id=1)
    depth = int(g("depth"))
    ref = po.Tracer(geoms, g("materials"), g("camera"), depth, tris=tris.view(po.TRI_DT), meshes=meshes.view(po.MESH_DT))
    W, H = 800, 800
    t0 = time.time()
    live = np.zeros(depth, dtype=np.int64)
    strip = 16
    for r in range(H // strip):                  # strip by strip: progress, and the same entry point the strip tests use
        st = ref.iterate_rows(1, r * strip, (r + 1) * strip, threads=threads)
        live += np.asarray(st.live[:depth], dtype=np.int64)
        print("strip %2d/%d  %.0f s" % (r + 1, H // strip, time.time() - t0), flush=True)
    img = ref.image
    strips = np.array([hashlib.md5(img[r * strip * W:(r + 1) * strip * W].tobytes()).hexdigest() for r in range(H // strip)])
    idx = np.random.default_rng(4).choice(W * H, 4096, replace=False).astype(np.int32)
    idx.sort()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "c4_frame.npz"),
                        image_md5=np.array(hashlib.md5(img.tobytes()).hexdigest()), strip_md5=strips, strip_rows=np.int32(strip),
                        live=live, rays=np.int64(live.sum()), sample_index=idx, sample_value=img[idx].copy(),
                        triangles=np.int32(len(tris)), seconds=np.float32(time.time() - t0), threads=np.int32(threads))
    print("done: %d rays, %.0f s, md5 %s" % (live.sum(), time.time() - t0, hashlib.md5(img.tobytes()).hexdigest()))


if __name__ == "__main__":
    main()
