#!/usr/bin/env python3
"""Soak / fuzz run on a GPU box (not part of the pytest suites): random scenes (cubes and spheres with random
rotations, thin slabs, enclosing shells, tiny and huge primitives, random materials) traced for a few iterations
under every pipeline, and the cull-stress rays of tests/cull_model.py through pt_intersect_once -- all compared with
the CPU oracle bit for bit.  `mesh` mode: PT_MESH_BVH on random soups and smooth meshes incl. grazing rays.
Usage: python tests/tools/fuzz_gpu.py [first_seed] [count]   |   python tests/tools/fuzz_gpu.py mesh [first_seed] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import cull_model  # noqa: E402
import mesh_cases  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def mesh_fuzz(pt, first, count):
    """PT_MESH_BVH against the oracle's loop over every triangle: random soups (slivers, zero-area, huge, duplicates)
    and smooth closed meshes (UV spheres: every silhouette ray grazes some triangle) with rays aimed at vertices,
    along edges, tangentially past the surface, and the bounce rays of whole iterations."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    base, mats, cam, depth = z["cornell_64__geoms"], z["cornell_64__materials"], z["cornell_64__camera"], int(z["cornell_64__depth"])
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        if (seed - first) % 25 == 0:
            print("mesh seed %d (%.0f s, %d mismatches so far)" % (seed, time.time() - t0, bad), flush=True)
        rng = np.random.default_rng(seed)
        if seed % 2:
            tris = mesh_cases.soup(pt.TRI_DT, rng)
        else:
            tris = pt.meshes.uv_sphere(center=tuple(rng.uniform(-2, 2, 3) + (0, 5, 0)), radius=float(rng.uniform(0.3, 2.5)),
                                       n_lat=int(rng.integers(4, 60)), n_lon=int(rng.integers(6, 120)))
        geoms, tris, meshes = pt.meshes.add_mesh(base[:6], tris, material_id=int(rng.integers(1, 5)))
        scene = pt.Scene(geoms, mats, cam, depth, triangles=tris, meshes=meshes)
        og, ot, om = geoms.view(po.GEOM_DT), tris.view(po.TRI_DT), meshes.view(po.MESH_DT)
        ref = po.Tracer(og, mats, cam, depth, trig=po.TRIG_SHARED, tris=ot, meshes=om)
        pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_MESH_BVH, max_batch=2)
        img = np.zeros((64 * 64, 3), dtype=np.float32)
        pt.trace_batch(1, 2, img)
        ref.iterate(1); ref.iterate(2)
        same = img.tobytes() == ref.image.tobytes()
        for it0, cnt in ((3, 1), (4, 2), (6, 1), (7, 1)):             # overlapped on the lanes (each with its own mesh buffers)
            pt.trace_batch_async(it0, cnt)
        pt.synchronize()
        for it in range(3, 8):
            ref.iterate(it)
        img = pt.get_image(64 * 64)
        pt.pathtraceFree()
        if not same or img.tobytes() != ref.image.tobytes():
            bad += 1
            print("mesh seed %d: IMAGE DIFFERS (%d pixels)" % (seed, int((img != ref.image).any(axis=1).sum())), flush=True)
        # aimed rays: vertices, edge midpoints, points just off the surface (tangential), from random origins
        k = min(len(tris), 1500)
        origin, d, _ = mesh_cases.aimed_rays(tris, rng, k)
        paths = np.zeros(k, dtype=pt.PATH_DT)
        paths["origin"], paths["direction"] = origin.astype(np.float32), d.astype(np.float32)
        want, _ = po.compute_intersections(paths.view(po.PATH_DT), og, ot, om)
        for name, extra in (("hierarchy", pt.PT_MESH_BVH), ("loop", 0)):
            pt.pathtraceInit(scene, flags=pt.PT_COMPACT | pt.PT_UNFUSED | extra)
            got, _ = pt.intersect_once(paths)
            pt.pathtraceFree()
            if got.tobytes() != want.tobytes():
                bad += 1
                diff = np.nonzero((got["t"].view(np.uint32) != want["t"].view(np.uint32)) | (got["materialId"] != want["materialId"]))[0]
                print("mesh seed %d (%s): %d of %d aimed rays differ (first: %s -> got t=%r, want t=%r)" %
                      (seed, name, len(diff), k, paths[diff[0]] if len(diff) else "normal only", got["t"][diff[0]] if len(diff) else 0,
                       want["t"][diff[0]] if len(diff) else 0), flush=True)
    print("mesh fuzz: seeds %d..%d, %d mismatching cases, %.0f s" % (first, first + count - 1, bad, time.time() - t0))
    return bad


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "mesh":
        pt = ge.load_package()
        po.build()
        return 1 if mesh_fuzz(pt, int(sys.argv[2]) if len(sys.argv) > 2 else 1, int(sys.argv[3]) if len(sys.argv) > 3 else 20) else 0
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    pt = ge.load_package()
    po.build()
    H = pt.host_binding.host_library()
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    cam = z["cornell_64__camera"]
    bad = 0
    t0 = time.time()
    for seed in range(first, first + count):
        if (seed - first) % 50 == 0:
            print("seed %d (%.0f s, %d mismatches so far)" % (seed, time.time() - t0, bad), flush=True)
        rng = np.random.default_rng(seed)
        nm = int(rng.integers(2, 12))
        mats = np.zeros(nm, dtype=pt.MATERIAL_DT)
        for m in mats:
            m["color"] = rng.uniform(0.1, 1.0, 3)
            m["spec_color"] = rng.uniform(0.5, 1.0, 3)
            kind = rng.integers(4)
            m["hasReflective"], m["hasRefractive"] = (1.0, 0.0) if kind == 1 else ((0.0, 1.0) if kind == 2 else (0.0, 0.0))
            m["indexOfRefraction"] = rng.uniform(1.05, 2.5)
            m["emittance"] = rng.uniform(1, 6) if kind == 3 else 0.0
        mats[0]["emittance"] = 4.0
        ng = int(rng.integers(1, 40))
        geoms = np.zeros(ng, dtype=pt.GEOM_DT)
        for k, g in enumerate(geoms):
            g["type"] = rng.integers(2)
            g["materialid"] = rng.integers(nm)
            g["translation"] = rng.uniform(-5, 5, 3) + (0, 5, 0)
            g["rotation"] = rng.uniform(-180, 180, 3) * (rng.random() < 0.6) + rng.choice([0.0, 90.0, 180.0, 45.0], 3) * (rng.random() < 0.3)
            sc = rng.uniform(0.2, 4.0, 3)
            r = rng.random()
            if r < 0.2:
                sc[rng.integers(3)] = rng.choice([0.02, 0.005, 0.001])
            elif r < 0.25:
                sc = np.full(3, rng.choice([30.0, 100.0]))
            elif r < 0.3:
                sc = np.full(3, rng.choice([1e-2, 1e-4]))
            elif r < 0.33:
                sc[rng.integers(3)] = 0.0
            g["scale"] = sc
            H.pth_build_geom_matrices(geoms.ctypes.data + k * pt.GEOM_DT.itemsize)
        depth = int(rng.integers(1, 12))
        # the camera moves too (the bounce-0 candidate masks and the cull boxes' reach follow it): near the scene, far
        # outside it, sometimes looking elsewhere; and either launch plan (a kernel per bounce / one launch per batch)
        cam = np.array(z["cornell_64__camera"], copy=True).reshape(1)
        r = rng.random()
        if r < 0.5:
            cam["position"][0] += rng.uniform(-3, 3, 3).astype(np.float32)
        elif r < 0.6:
            cam["position"][0] += (rng.uniform(-1, 1, 3) * rng.choice([30.0, 300.0])).astype(np.float32)
        if rng.random() < 0.2:
            a = rng.uniform(-0.8, 0.8)
            cam["view"][0] = np.float32([np.sin(a), 0.0, np.cos(a)]) * np.float32(np.sign(cam["view"][0][2]) or 1.0)
        if rng.random() < 0.5:
            os.environ["PTMI355_WHOLE_MAX"] = "0"
        else:
            os.environ.pop("PTMI355_WHOLE_MAX", None)
        scene = pt.Scene(geoms, mats, cam, depth)
        ogeoms, omats = geoms.view(po.GEOM_DT), mats.view(po.MATERIAL_DT)
        # (a) whole iterations under three pipelines: a batch of two, then pathtrace() per call with the host image
        for flags, oflags in ((pt.PT_COMPACT, po.F_COMPACT), (0, 0), (pt.PT_COMPACT | pt.PT_SORT_MATERIAL, po.F_COMPACT | po.F_SORT)):
            ref = po.Tracer(ogeoms, omats, cam, depth, flags=oflags, trig=po.TRIG_SHARED)
            pt.pathtraceInit(scene, flags=flags, max_batch=2)
            img = np.zeros((64 * 64, 3), dtype=np.float32)
            pt.trace_batch(1, 2, img)
            ref.iterate(1); ref.iterate(2)
            same = img.tobytes() == ref.image.tobytes()
            got = pt.pathtrace(None, 0, 3)
            ref.iterate(3)
            same = same and got.tobytes() == ref.image.tobytes()
            # batches nobody waits for: consecutive ones overlap on the device (lanes), the sums stay in iteration order
            for it0, cnt in ((4, 1), (5, 1), (6, 2), (8, 1), (9, 1)):
                pt.trace_batch_async(it0, cnt)
            pt.synchronize()
            for it in range(4, 10):
                ref.iterate(it)
            got = pt.get_image(64 * 64)
            same = same and got.tobytes() == ref.image.tobytes()
            pt.pathtraceFree()
            if not same:
                bad += 1
                print("seed %d flags %d: IMAGE DIFFERS (%d pixels)" % (seed, flags, int((got != ref.image).any(axis=1).sum())), flush=True)
        # (b) cull-stress rays through computeIntersections
        finite = np.isfinite(geoms["transform"]).all(axis=(1, 2))
        rays = cull_model.stress_rays(geoms[finite], rng, per_geom=max(200, 6000 // max(1, int(finite.sum()))))
        paths = np.zeros(len(rays), dtype=pt.PATH_DT)
        paths["origin"], paths["direction"] = rays[:, :3], rays[:, 3:]
        pt.pathtraceInit(scene, max_batch=1 + len(rays) // (64 * 64))
        got, got_out = pt.intersect_once(paths)
        pt.pathtraceFree()
        want, want_out = po.compute_intersections(paths.view(po.PATH_DT), ogeoms)
        same = (got["t"].view(np.uint32) == want["t"].view(np.uint32)) & \
               (got["normal"].view(np.uint32) == want["normal"].view(np.uint32)).all(axis=1) & (got["materialId"] == want["materialId"])
        hit = want["t"] > 0
        same &= ~hit | (got_out == want_out)
        if not same.all():
            bad += 1
            i = int(np.nonzero(~same)[0][0])
            print("seed %d: %d of %d stress rays differ, e.g. ray %s got t=%r want t=%r" % (seed, int((~same).sum()), len(rays), rays[i], got["t"][i], want["t"][i]), flush=True)
    print("fuzz: seeds %d..%d, %d mismatching cases, %.0f s" % (first, first + count - 1, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
