#!/usr/bin/env python3
"""mesh_stats.py [batch] [steps]: C4 + hierarchy traced synchronously through a -DPT_MESH_STATS build (PTMI355_LIB=
.ab/<name>/libptmi355.so): every synchronous batch prints k_mesh's diagnostic counters (csrc/pt_k_mesh.hpp: g_mesh_stats)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pt = ge.load_package()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
g = lambda k: z["cornell__" + k]
tris = pt.meshes.uv_sphere(n_lat=97, n_lon=521)
geoms, tris, meshes = pt.meshes.add_mesh(g("geoms"), tris, material_id=1)
scene = pt.Scene(geoms, g("materials"), g("camera"), int(g("depth")), triangles=tris, meshes=meshes)
print(pt.version(), file=sys.stderr)
pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_MESH_BVH, max_batch=batch)
for k in range(steps):
    pt.trace_batch(1 + k * batch, batch, None)
    print("-- after step %d: %d rays since init" % (k + 1, pt.total_rays()), file=sys.stderr)
pt.pathtraceFree()
