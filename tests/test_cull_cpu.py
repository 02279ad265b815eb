"""The cull stage may only drop (ray, primitive) pairs the reference's own float arithmetic misses: for rays that
stress the boxes of pt_cull.hpp (faces, edges, corners, grazing, inside, behind, axis-parallel, non-finite), every
pair the ORACLE reports as a hit must be a candidate of the numpy model of the kernel's test.  CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cull_model  # noqa: E402


@pytest.fixture(scope="module")
def pt():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def _check(pt, po, geoms, eye, rng, per_geom):
    boxes, rmax, rej = pt.cull_boxes(geoms, eye)
    rays = cull_model.stress_rays(geoms, rng, per_geom=per_geom)
    paths = np.zeros(len(rays), dtype=po.PATH_DT)
    paths["origin"], paths["direction"] = rays[:, :3], rays[:, 3:]
    total_hits = total_cand = 0
    for gi in range(len(geoms)):
        want, _ = po.compute_intersections(paths, geoms[gi:gi + 1].view(po.GEOM_DT))
        # any t the loop of pathtrace.cu:176-199 would take (t > 0), and NaN distances (they come from a passed test)
        hit = (want["t"] > 0) | np.isnan(want["t"])
        cand, _wild = cull_model.candidates(rays, boxes[gi], rmax, rej[gi])
        lost = hit & ~cand
        assert not lost.any(), "geom %d: %d hits outside the box, e.g. ray %s" % (gi, lost.sum(), rays[np.nonzero(lost)[0][0]])
        total_hits += int(hit.sum()); total_cand += int(cand.sum())
    return total_hits, total_cand, len(rays) * len(geoms)


def test_cornell_boxes(pt, po, scenes):
    s = scenes["cornell"]
    eye = s["camera"].view(pt.CAMERA_DT)[0]["position"]
    hits, cand, pairs = _check(pt, po, s["geoms"], eye, np.random.default_rng(7), 6000)
    assert hits > 20000 and cand < pairs                   # the rays do hit, and the boxes do cull


@pytest.mark.parametrize("seed", range(6))
def test_random_transforms(pt, po, scenes, seed):
    """Rotated, non-uniformly scaled (thin slabs, 1000:1), tiny and huge primitives; singular ones get no box."""
    rng = np.random.default_rng(100 + seed)
    H = pt.host_binding.host_library()
    ng = 10
    geoms = np.zeros(ng, dtype=pt.GEOM_DT)
    for k, g in enumerate(geoms):
        g["type"] = rng.integers(2)
        g["translation"] = rng.uniform(-4, 4, 3) + (0, 5, 0)
        g["rotation"] = rng.uniform(-180, 180, 3) * (rng.random() < 0.8)
        sc = rng.uniform(0.3, 3.0, 3)
        if k % 3 == 0:
            sc[rng.integers(3)] = rng.choice([0.02, 0.003])
        if k == 1:
            sc = np.array([25.0, 25.0, 25.0])
        if k == 2:
            sc = np.array([1e-3, 1e-3, 1e-3])
        if k == 4:
            sc = np.array([0.0, 1.0, 1.0])                 # singular: inverse holds inf / NaN
        if k == 5:
            g["type"], g["rotation"] = 1, 0.0              # an axis-aligned cube: has an exact early-miss axis
        g["scale"] = sc
        H.pth_build_geom_matrices(geoms.ctypes.data + k * pt.GEOM_DT.itemsize)
    boxes, _, rej = pt.cull_boxes(geoms, (0.0, 5.0, 10.5))
    assert rej[5, 0] in (0, 1, 2) and rej[4, 0] == 3 and (rej[geoms["type"] == 0, 0] == 3).all()   # cubes only; diagonal row for the axis-aligned one
    assert (rej[(geoms["type"] == 1) & (np.arange(len(geoms)) != 4), 0] != 3).all()              # every finite cube has a row
    assert np.isinf(boxes[4]).all()                        # no culling for the singular one
    assert np.isfinite(boxes[0]).all()
    _check(pt, po, geoms, (0.0, 5.0, 10.5), rng, 2500)


def test_far_scene_axis_parallel_rays(pt, po):
    """the kernel forms n = -o / d with 1 / d clamped to +-2^100, so |o| must stay below 2^27 for a ray
    that is parallel to an axis (d_k = 0) -- beyond it the product is inf, both planes of the slab the ray runs INSIDE
    come out -inf and the ray would be culled.  make_boxes caps the origin bound at 2^27: rays from farther out are
    `wild` (candidates of every primitive).  A cube 6e8 units across, rays along the axes from inside it, 3e8 out."""
    H = pt.host_binding.host_library()
    geoms = np.zeros(1, dtype=pt.GEOM_DT)
    geoms[0]["type"] = 1
    geoms[0]["scale"] = (6e8, 6e8, 6e8)
    H.pth_build_geom_matrices(geoms.ctypes.data)
    boxes, rmax, rej = pt.cull_boxes(geoms, (0.0, 0.0, 0.0))
    assert rmax <= 2.0 ** 27
    rng = np.random.default_rng(1)
    n = 600
    rays = np.zeros((n, 6), dtype=np.float32)
    rays[:, :3] = rng.uniform(-2.9e8, 2.9e8, (n, 3))
    ax = rng.integers(0, 3, n)
    rays[np.arange(n), 3 + ax] = rng.choice([-1.0, 1.0], n)
    rays[: n // 2, ax[0] % 3] = rng.choice([-2.95e8, 2.95e8], n // 2)          # close to a face, far beyond 2^28
    paths = np.zeros(n, dtype=po.PATH_DT)
    paths["origin"], paths["direction"] = rays[:, :3], rays[:, 3:]
    want, _ = po.compute_intersections(paths, geoms.view(po.GEOM_DT))
    hit = (want["t"] > 0) | np.isnan(want["t"])
    assert hit.sum() > n // 2                                                  # from inside: they do hit
    cand, wild = cull_model.candidates(rays, boxes[0], rmax, rej[0])
    assert not (hit & ~cand).any()
    assert wild[np.abs(rays[:, :3]).sum(axis=1) > 2.0 ** 27].all()
