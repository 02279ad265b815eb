"""The every-triangle loop's first stage (MESH_TILES; csrc/pt_kernels.hpp: mesh_sweep, csrc/ptmi355.hip:
make_tri_bounds) on the CPU: a per-triangle sphere {c, Rs} decides which (ray, triangle) pairs get the exact
glm::intersectRayTriangle + hit-point test.  It may only ever drop pairs the completion spec would not count.  Checked
here against the oracle's own accept decision for every pair: rays aimed at vertices, edges and interiors, rays that
graze metre-sized triangles from inside their plane (where glm's float test accepts rounding noise), slivers,
zero-area triangles, far origins.  The kernel's arithmetic (fused multiply-adds in binary32) is modelled in binary64
with the documented error budget left over as margin."""
import numpy as np
import pytest

import __graft_entry__ as ge
import mesh_cases


@pytest.fixture(scope="module")
def pt():
    ge.load_package().build()
    return ge.load_package()


def line_distance(o, d, c):
    """distance of every ray's LINE to every centre: [rays, triangles], binary64"""
    dn = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-300)
    w = c[None, :, :] - o[:, None, :]
    along = (w * dn[:, None, :]).sum(axis=2, keepdims=True)
    return np.linalg.norm(w - along * dn[:, None, :], axis=2)


def check(pt, po, tris, origin, direction, origin_bound):
    rays = np.zeros(len(origin), dtype=po.PATH_DT)
    rays["origin"], rays["direction"] = origin.astype(np.float32), direction.astype(np.float32)
    acc = po.mesh_accepted(tris.view(po.TRI_DT), rays).astype(bool)
    b = pt.tri_bounds(tris, origin_bound).astype(np.float64)
    o64, d64 = rays["origin"].astype(np.float64), rays["direction"].astype(np.float64)
    inside = np.abs(o64).sum(axis=1) <= origin_bound               # the others are "wild": candidates of every triangle
    dist = line_distance(o64, d64, b[:, :3])
    rs = np.sqrt(np.maximum(b[:, 3], 0.0))
    # what make_tri_bounds reserves for the kernel's own rounding (4 x 2^-19 x reach, reach >= origin bound): accepted
    # pairs must clear the radius by at least that much in exact arithmetic
    margin = 2.0 ** -18 * origin_bound
    bad = acc & inside[:, None] & (dist > (rs - margin)[None, :])
    assert not bad.any(), (int(bad.sum()), np.argwhere(bad)[:5].tolist())
    matrix_form_keeps(pt, tris, rays, acc, inside, origin_bound)
    return acc, dist, rs


E_FORM, FAR_M2 = np.float32(4.0e-5), np.float32(1.21)          # csrc/pt_k_trisweep.hpp: TRI_FORM_E, TRI_FAR_M2


def half_pair(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def matrix_form_keeps(pt, tris, rays, acc, inside, origin_bound):
    """Round 6: the kernel evaluates the same spheres on the matrix pipe, as ONE bilinear form per pair (csrc/pt_k_trisweep.hpp:
    mesh_sweep; the triangles' side from pt_tri_records, the rays' side restated here in the kernel's own single-precision
    operations).  Products of binary16 slots are exact; the MFMA's 31 binary32 additions may round in any order, so the
    model adds the worst case, 32 x 2^-24 x the sum of the terms' absolute values: a pair the oracle accepts must still
    come out NEGATIVE (a candidate), and its ray must not have been classified as passing the mesh at a distance."""
    rec, frame = pt.tri_records(tris, origin_bound)
    a = rec[:len(tris)].astype(np.float64)                          # [triangles, 32]
    o, d = rays["origin"].astype(np.float32), rays["direction"].astype(np.float32)
    f32 = np.float32
    with np.errstate(all="ignore"):
        n2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        sc = (f32(1.0) / np.sqrt(n2)).astype(f32)                   # (v_rsq_f32: within an ulp or two of this)
        dn = d * sc[:, None]
        op = ((o - frame[None, :3]) * frame[3]).astype(f32)
        m = np.cross(op.astype(np.float64), dn.astype(np.float64)).astype(f32)      # (three fma each: one rounding, like the kernel's)
        M = ((m.astype(np.float64) ** 2).sum(axis=1)).astype(f32)
        w = np.cross(dn, m).astype(f32)
        v = np.stack([dn[:, 0] * dn[:, 0], dn[:, 1] * dn[:, 1], dn[:, 2] * dn[:, 2], dn[:, 0] * dn[:, 1], dn[:, 0] * dn[:, 2], dn[:, 1] * dn[:, 2],
                      f32(-2) * w[:, 0], f32(-2) * w[:, 1], f32(-2) * w[:, 2]], axis=1).astype(f32)
        b = np.zeros((len(o), 32), dtype=np.float64)
        hi, lo = half_pair(v)
        b[:, 0:27:3], b[:, 1:27:3], b[:, 2:27:3] = hi, lo, hi           # a term's slots here: hi, lo, hi (there: hi, hi, lo)
        b[:, 27] = b[:, 28] = 1.0
        mh, ml = half_pair((M - E_FORM).astype(f32))
        b[:, 29], b[:, 30] = mh, ml
        far = ~(M <= FAR_M2)
        val = b @ a.T                                               # exact products, summed in binary64
        mag = np.abs(b) @ np.abs(a).T
    worst = val + 32 * 2.0 ** -24 * mag
    must = acc & inside[:, None]
    assert not (must & far[:, None]).any(), "an accepted pair's ray was classified as passing the mesh at a distance"
    lost = must & ~(worst < 0)
    assert not lost.any(), (int(lost.sum()), np.argwhere(lost)[:5].tolist(), worst[lost][:5].tolist())
    # the stage is still worth having on this mesh: what it keeps is a small multiple of what round 5's fp32 form kept
    ok = inside & ~far
    if ok.any() and len(tris) >= 64:
        kept = (val[ok] < 0).mean()
        assert kept < 0.5, kept


@pytest.mark.parametrize("size", [(8, 16), (30, 60)])
def test_smooth_mesh(pt, po, size):
    tris = pt.meshes.uv_sphere(center=(1.5, 3.0, 1.0), radius=1.5, n_lat=size[0], n_lon=size[1])
    rng = np.random.default_rng(5)
    o, d, _ = mesh_cases.aimed_rays(tris, rng, 600)
    acc, dist, rs = check(pt, po, tris, o, d, 64.0)
    assert acc.sum() > 300
    # the stage is worth having: it passes a few pairs per ray, not a few per cent of the mesh
    cand = (dist <= rs[None, :]).sum(axis=1).mean()
    assert cand < 0.05 * len(tris) + 8, cand
    # radii track the triangles: the largest is a small multiple of the longest edge's half
    edge = max(np.linalg.norm(tris["v1"] - tris["v0"], axis=1).max(), np.linalg.norm(tris["v2"] - tris["v0"], axis=1).max())
    assert rs.max() < 1.2 * edge


@pytest.mark.parametrize("seed", [17, 37, 101, 102, 7])
def test_grazing_soups(pt, po, seed):
    rng = np.random.default_rng(seed)
    tris = mesh_cases.soup(pt.TRI_DT, rng, n=400)
    o, d, graze = mesh_cases.aimed_rays(tris, rng, 500)
    acc, _, _ = check(pt, po, tris, o, d, 64.0)
    assert acc[graze].sum() > 0 and acc[~graze].sum() > 0


def test_far_origins_and_odd_triangles(pt, po):
    rng = np.random.default_rng(3)
    tris = mesh_cases.soup(pt.TRI_DT, rng, n=200)
    tris["v1"][5] = tris["v0"][5]                                  # zero area
    tris["v2"][6] = tris["v1"][6]
    tris["v0"][7] = (np.nan, 0, 0)                                 # never accepted; its sphere must not matter
    tris["v1"][8] = (np.inf, 0, 0)
    o, d, _ = mesh_cases.aimed_rays(tris, rng, 300)
    far = o + (d * -3000.0)                                        # the same lines from 3000 units away
    acc, _, _ = check(pt, po, tris, np.concatenate([o, far]), np.concatenate([d, d]), 16384.0)
    assert acc[:300].sum() > 0
    b = pt.tri_bounds(tris, 16384.0)
    assert np.isinf(b[7, 3]) and np.isinf(b[8, 3]) and np.isfinite(b[5, 3]) and b[5, 3] > 0
    # padding entries of the device array: nothing is a candidate
    raw = np.zeros((8, 4), dtype=np.float32)
    assert pt.library().pt_tri_bounds(tris[:5].ctypes.data, 5, np.float32(64.0), raw.ctypes.data) == 8
    assert (raw[5:, 3] == -1.0).all() and (raw[:5, 3] > 0).all()


def test_rows_of_an_iteration(po, scenes):
    """pto_trace_rows_mt (the oracle for ONE strip of a frame: what the C4 strip test on the GPU is held against) ==
    the same rows of the whole-frame oracle iteration, threads or not."""
    s = scenes["cornell_glass_64"]
    W, H = [int(v) for v in s["camera"][0]["resolution"]]
    whole = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
    whole.iterate(1)
    whole.iterate(2)
    for y0, y1, threads in ((0, H, 3), (16, 32, 1), (40, 45, 4)):
        part = po.Tracer(s["geoms"], s["materials"], s["camera"], s["depth"])
        st = [part.iterate_rows(it, y0, y1, threads) for it in (1, 2)]
        assert st[0].live[0] == (y1 - y0) * W
        assert part.image[y0 * W:y1 * W].tobytes() == whole.image[y0 * W:y1 * W].tobytes()
        rest = np.ones(W * H, dtype=bool)
        rest[y0 * W:y1 * W] = False
        assert not part.image[rest].any()
