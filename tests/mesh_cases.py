"""Triangle meshes and rays that are hard on a spatial hierarchy (shared by tests/test_bvh_cpu.py, the GPU parity
tests and tests/tools/fuzz_gpu.py): random soups with slivers, zero-area and metre-sized triangles, smooth closed
meshes, and rays aimed at vertices / edges / interiors from origins that lie (almost) in the target triangle's
plane -- where single-precision glm::intersectRayTriangle reports barycentrics that are rounding noise and the
spec's hit-point test (oracle/ptoracle.c: pto_tri_point_ok) decides."""
import numpy as np


def soup(tri_dt, rng, n=None):
    n = int(rng.integers(50, 3000)) if n is None else n
    c = rng.uniform(-3, 3, (n, 3)) + (0, 5, 0)
    size = 10 ** rng.uniform(-2.5, 0.6, (n, 1))
    v = [c + rng.normal(size=(n, 3)) * size for _ in range(3)]
    sl = rng.random(n) < 0.1
    v[2][sl] = v[1][sl] + (v[1][sl] - v[0][sl]) * 1e-4 + rng.normal(size=(sl.sum(), 3)) * 1e-6
    tris = np.zeros(n, dtype=tri_dt)
    tris["v0"], tris["v1"], tris["v2"] = v
    return tris


def aimed_rays(tris, rng, k):
    """(origin, direction, graze) float64 arrays: k rays aimed at picked triangles; half of the origins lie within
    1e-7 .. 1e-2 of the target triangle's plane."""
    pick = rng.integers(len(tris), size=k)
    tv = np.stack([tris["v0"][pick], tris["v1"][pick], tris["v2"][pick]], axis=1).astype(np.float64)
    w = rng.dirichlet((0.3, 0.3, 0.3), size=k)
    w[: k // 4] = np.eye(3)[rng.integers(3, size=k // 4)]                         # exact vertices
    target = (tv * w[:, :, None]).sum(axis=1)
    nrm = np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    origin = rng.uniform(-6, 6, (k, 3)) + (0, 5, 0)
    graze = rng.random(k) < 0.5
    inplane = rng.normal(size=(k, 3))
    inplane -= nrm * (inplane * nrm).sum(axis=1, keepdims=True)
    inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-30)
    origin[graze] = (target + inplane * rng.uniform(1, 8, (k, 1)) +
                     nrm * (10 ** rng.uniform(-7, -2, (k, 1))) * rng.choice([-1, 1], (k, 1)))[graze]
    d = target - origin
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    return origin, d, graze
