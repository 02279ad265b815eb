"""Numpy model of the cull stage of the intersection kernels (csrc/pt_kernels.hpp: cull_ray / cull_box) and a
generator of rays that stress it: shared by the CPU test (no oracle hit may lie outside its primitive's box) and
the GPU gate test (bit-exact intersections for those rays)."""
import numpy as np

F = np.float32
BIG = F(2.0 ** 100)


def centre_half(box):
    """csrc/pt_cull.hpp: centre_half -- what the kernel's cull record holds for the box [lo, hi]."""
    lo, hi = box[0].astype(np.float64), box[1].astype(np.float64)
    c = ((lo + hi) * 0.5).astype(F)
    h64 = np.nextafter(np.fmax(hi - c.astype(np.float64), c.astype(np.float64) - lo), np.inf)
    h = h64.astype(F)
    h = np.where(h.astype(np.float64) < h64, np.nextafter(h, F(np.inf)), h).astype(F)
    fin = np.isfinite(lo) & np.isfinite(hi)
    return np.where(fin, c, F(0)).astype(F), np.where(fin, h, F(np.inf)).astype(F)


def candidates(rays, box, rmax, reject=None):
    """rays[n, 6] float32 (origin, direction); box[2, 3] (lo, hi); reject = (mode, row[4]) of the exact
    one-axis early miss or None; returns (candidate mask, wild mask)."""
    o = rays[:, :3].astype(F)
    d = rays[:, 3:].astype(F)
    with np.errstate(all="ignore"):
        os_ = (np.abs(o[:, 0]) + np.abs(o[:, 1])).astype(F) + np.abs(o[:, 2])
        ds = (np.abs(d[:, 0]) + np.abs(d[:, 1])).astype(F) + np.abs(d[:, 2])
        wild = ~(os_ <= F(rmax)) | ~((ds >= F(2.0 ** -20)) & (ds <= F(2.0 ** 20)))
        ix = np.clip((F(1) / d).astype(F), -BIG, BIG)                      # v_rcp_f32 (1 ulp) + v_med3_f32
        n = (-o * ix).astype(F)
        c, h = centre_half(box)
        tm = (c.astype(np.float64) * ix.astype(np.float64) + n.astype(np.float64)).astype(F)         # v_fma_f32
        hw = h.astype(np.float64) * np.abs(ix).astype(np.float64)
        ta = (tm.astype(np.float64) - hw).astype(F)                                                  # v_fma_f32 (-half, |1/d|, t_mid)
        tb = (tm.astype(np.float64) + hw).astype(F)
        tn = np.fmax(np.fmax(ta[:, 0], ta[:, 1]), np.fmax(ta[:, 2], F(0)))
        tf = np.fmin(np.fmin(tb[:, 0], tb[:, 1]), tb[:, 2])
        cand = ~(tn > tf)
        if reject is not None and int(reject[0]) != 3:
            mode, row = int(reject[0]), reject[1:5].astype(F)
            if mode == 4:                          # general row, glm's order: (m0 x + m1 y) + (m2 z + m3)
                qk = ((row[0] * o[:, 0]).astype(F) + (row[1] * o[:, 1]).astype(F)).astype(F) + \
                     ((row[2] * o[:, 2]).astype(F) + row[3]).astype(F)
                vk = ((row[0] * d[:, 0]).astype(F) + (row[1] * d[:, 1]).astype(F)).astype(F) + (row[2] * d[:, 2]).astype(F)
                qk, vk = qk.astype(F), vk.astype(F)
            else:
                qk = ((row[mode] * o[:, mode]).astype(F) + row[3]).astype(F)
                vk = (row[mode] * d[:, mode]).astype(F)
            cand &= ~((np.abs(qk) > F(0.5)) & ((qk * vk).astype(F) > F(0)))
    return cand | wild, wild


def _unit(v):
    n = np.linalg.norm(v, axis=1, keepdims=True)
    return v / np.where(n > 0, n, 1.0)


def stress_rays(geoms, rng, per_geom=4000, reach=12.0):
    """Rays aimed at / grazing / inside / behind every primitive of `geoms` (numpy GEOM_DT array), float32 [n, 6]."""
    out = []
    for g in geoms:
        T = g["transform"].astype(np.float64).T                            # m[col][row] -> row-major 4x4
        if not np.isfinite(T).all():
            continue
        n = per_geom
        # points on / near the surface of the unit cube in object space: faces, edges, corners
        p = rng.uniform(-0.5, 0.5, (n, 3))
        kind = rng.integers(0, 4, n)
        for axis_count in (1, 2, 3):                                        # snap 1, 2 or 3 coordinates to +-0.5
            sel = kind == axis_count
            for k in range(axis_count):
                ax = (rng.integers(0, 3, n) + k) % 3
                sgn = rng.choice([-0.5, 0.5], n)
                p[sel, ax[sel]] = sgn[sel]
        if g["type"] == 0:                                                  # sphere: project a share onto the surface
            sel = rng.random(n) < 0.6
            p[sel] = 0.5 * _unit(rng.normal(size=(n, 3)))[sel]
        # offsets from 1e-7 to ~1 object units, both sides
        off = rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 1.0], n)[:, None] * _unit(rng.normal(size=(n, 3)))
        target_obj = np.concatenate([p + off, np.ones((n, 1))], axis=1)
        target = (target_obj @ T.T)[:, :3]
        origin = rng.uniform(-reach, reach, (n, 3)) + np.array([0.0, 5.0, 0.0])
        # a share of the origins ON another point of the surface (bounce rays), INSIDE, or far away
        q = rng.uniform(-0.5, 0.5, (n, 3))
        q[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-0.5, 0.5], n)
        on_surface = (np.concatenate([q, np.ones((n, 1))], axis=1) @ T.T)[:, :3]
        inside = (np.concatenate([rng.uniform(-0.45, 0.45, (n, 3)), np.ones((n, 1))], axis=1) @ T.T)[:, :3]
        mode = rng.integers(0, 10, n)
        origin[mode == 0] = on_surface[mode == 0]
        origin[mode == 1] = inside[mode == 1]
        origin[mode == 2] *= 30.0                                           # beyond the origin bound: `wild`
        # bounce rays: the origin sits where getPointOnRay leaves it, up to 3e-4 OBJECT units off the face it hit
        # (1e-6 world units off a 0.01-thick wall), on either side -- the exact one-axis early miss must decide these
        # as the reference's own rounding does
        lift = q.copy()
        ax = np.argmax(np.abs(q) == 0.5, axis=1)
        lift[np.arange(n), ax] *= 1.0 + rng.choice([-6e-4, -2e-4, -1e-5, 1e-5, 2e-4, 6e-4], n)
        bounce = (np.concatenate([lift, np.ones((n, 1))], axis=1) @ T.T)[:, :3]
        sel = (mode == 3) | (mode == 4)
        origin[sel] = bounce[sel]
        d = target - origin
        away = _unit(rng.normal(size=(n, 3)))                                # leaving in a random direction
        d[mode == 4] = away[mode == 4]
        scale = rng.choice([1.0, 1.0, 1.0, 1e-3, 37.0], n)[:, None]        # un-normalised directions too
        d = _unit(d) * scale
        # behind: flip a share so the primitive lies behind the ray
        flip = rng.random(n) < 0.15
        d[flip] = -d[flip]
        rays = np.concatenate([origin, d], axis=1).astype(F)
        # axis-parallel and zero-component directions through / past the box
        m = n // 8
        ap = rays[:m].copy()
        ax = rng.integers(0, 3, m)
        ap[:, 3:] = 0.0
        ap[np.arange(m), 3 + ax] = rng.choice([-1.0, 1.0], m)
        ap[:, :3] = target[:m] - ap[:, 3:] * rng.uniform(0.5, 9.0, (m, 1))
        zc = rays[m:2 * m].copy()
        zc[np.arange(m), 3 + rng.integers(0, 3, m)] = rng.choice([0.0, -0.0], m)
        out += [rays, ap.astype(F), zc.astype(F)]
    rays = np.concatenate(out)
    # a few non-finite / degenerate ones
    bad = rays[:64].copy()
    bad[0:8, 0] = np.nan; bad[8:16, 4] = np.nan; bad[16:24, 1] = np.inf; bad[24:32, 5] = -np.inf
    bad[32:40, 3:] = 0.0; bad[40:48, 3:] = 1e-30; bad[48:56, 3:] *= 1e30; bad[56:64, :3] = 1e20
    return np.concatenate([rays, bad]).astype(F)
