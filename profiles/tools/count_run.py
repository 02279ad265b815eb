#!/usr/bin/env python3
"""Run a bench.py workload on a library whose kernel counts its basic blocks (profiles/tools/build_counted.sh) and write
the counts:   PTMI355_LIB=.ab/NAME/libptmi355.so count_run.py CONFIG FLAGS .ab/NAME/map.json OUT.u32 [steps] [batch]
Prints the image md5 (must equal the uninstrumented build's: the instrumentation changes no result) and the number of
launches of every stage."""
import ctypes as C, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
config, flags_s, map_path, out_path = sys.argv[1:5]
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
batch = int(sys.argv[6]) if len(sys.argv) > 6 else 64
words = json.load(open(map_path))["words"]
os.environ["PTMI355_DBG_COUNTS"] = str(words)
import __graft_entry__ as ge
pt = ge.load_package()
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
name = {"c0": "cornell_64", "c2": "cornell", "c3": "cornell_glass", "c4": "cornell", "c5": "cornell_4k"}[config]
g = lambda k: z[name + "__" + k]
scene = pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")))
if config == "c4":
    tris = pt.meshes.uv_sphere(n_lat=97, n_lon=521)
    geoms, tris, meshes = pt.meshes.add_mesh(scene.geoms, tris, material_id=1)
    scene = pt.Scene(geoms, scene.materials, scene.camera, scene.traceDepth, triangles=tris, meshes=meshes)
flags = 0
for f in flags_s.split(","):
    flags |= {"compact": pt.PT_COMPACT, "sort": pt.PT_SORT_MATERIAL, "unfused": pt.PT_UNFUSED, "bvh": pt.PT_MESH_BVH, "": 0}[f]
pt.pathtraceInit(scene, flags=flags, max_batch=batch)
L = pt.library()
buf = np.zeros(words, dtype=np.uint32)
pt.set_profiling(True)
for k in range(steps):
    pt.trace_batch_async(1 + k * batch, batch)
pt.synchronize()
assert L.ptdbg_counts(buf.ctypes.data_as(C.c_void_p), words) == words
prof = pt.get_profile()
W, H = scene.resolution
img = pt.get_image(W * H)
rays, first, iters = pt.counters()
pt.pathtraceFree()
buf.tofile(out_path)
print(json.dumps({"config": config, "flags": flags_s, "steps": steps, "batch": batch, "rays": rays, "first_bounce_rays": first,
                  "image_md5": hashlib.md5(img.tobytes()).hexdigest(), "launches": {k: v[1] for k, v in prof.items() if v[1]},
                  "stage_ms": {k: round(v[0], 3) for k, v in prof.items() if v[1]}, "counted_blocks_nonzero": int((buf != 0).sum())}))
