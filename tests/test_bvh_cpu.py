"""PT_MESH_BVH on the CPU: the hierarchy pt_init builds (host code of libptmi355.so, no GPU
needed) is structurally sound, and the kernel's stackless walk -- restated here over the built nodes, with
the oracle's triangle test at the leaves -- finds the same winner as the oracle's loop over every triangle."""
import numpy as np
import pytest

import __graft_entry__ as ge


@pytest.fixture(scope="module")
def pt():
    ge.load_package().build()
    return ge.load_package()


def _tree(pt, tris):
    """nodes (uint32 records), child boxes decoded to grid planes [n, 2, 2, 3] (child, lo/hi, xyz), order, grid.
    A record stores each box as centre m and half extent e on the grid (csrc/pt_bvh.hpp): planes m - e, m + e."""
    nodes, order, grid = pt.binding.bvh_build(tris)
    w = nodes[:, 0:6].reshape(-1, 2, 3).astype(np.int64)
    m = np.stack([w[:, :, 0] & 0xffff, w[:, :, 0] >> 16, w[:, :, 1] & 0xffff], axis=-1)
    e = np.stack([w[:, :, 1] >> 16, w[:, :, 2] & 0xffff, w[:, :, 2] >> 16], axis=-1)
    g = np.stack([m - e, m + e], axis=2)
    return nodes, g.astype(np.float64), order, grid


LEAF = 8


def _children(nodes, k):
    """[(link, count, is_leaf)] of record k; split axis"""
    kids = []
    for c in (0, 1):
        v = int(nodes[k, 6 + c])
        kids.append((v & 0xffffff, (v >> 24) & 7, bool((v >> 24) & LEAF)))
    return kids, (int(nodes[k, 6]) >> 28) & 3


def _miss(nodes, k, octant):
    return int(nodes[k, 8 + octant].astype(np.int32))


@pytest.mark.parametrize("size", [(8, 16), (30, 60), (97, 521)])
def test_tree_structure(pt, size):
    tris = pt.meshes.uv_sphere(n_lat=size[0], n_lon=size[1])
    nodes, gbox, order, grid = _tree(pt, tris)
    n = len(nodes)
    origin, step, pad = grid[0:3].astype(np.float64), grid[3:6].astype(np.float64), float(grid[6])
    box = origin + gbox * step                                        # world-space planes
    assert sorted(order.tolist()) == list(range(len(tris)))          # every triangle in exactly one leaf slot
    verts = np.stack([tris["v0"], tris["v1"], tris["v2"]], axis=1)      # (T, 3, 3)
    leaves, parents = [], np.zeros(n, dtype=np.int32)
    for k in range(n):
        kids, axis = _children(nodes, k)
        assert axis <= 2
        for c, (link, count, is_leaf) in enumerate(kids):
            lo, hi = box[k, c, 0], box[k, c, 1]
            if is_leaf:
                assert 1 <= count <= 4
                leaves.append((link, count))
                v = verts[order[link:link + count]].reshape(-1, 3)
                # the box holds its triangles with the padding to spare, and is not wastefully loose
                assert (v.min(axis=0) - 0.9 * pad > lo).all() and (v.max(axis=0) + 0.9 * pad < hi).all()
                assert (v.min(axis=0) - pad - 6 * step < lo).all() and (v.max(axis=0) + pad + 6 * step > hi).all()
            else:
                assert k < link < n and count == 0                    # records are laid out parent first
                parents[link] += 1
                # a child's boxes lie inside its own box up to the re-centring of each box on the grid (planes m -+ e:
                # one spare step on either side, the lower one more when lo + hi is odd) -- the walk needs no exact
                # nesting: a box only decides which exact triangle tests run
                for cc in (0, 1):
                    assert (gbox[link, cc, 0] >= gbox[k, c, 0] - 2).all() and (gbox[link, cc, 1] <= gbox[k, c, 1] + 2).all()
    assert parents[0] == 0 and (parents[1:] == 1).all()
    leaves.sort()
    assert leaves[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(leaves, leaves[1:]))
    assert leaves[-1][0] + leaves[-1][1] == len(tris)
    # per octant: "hit everything" visits every record exactly once and ends; the root's links end the walk
    for octant in range(8):
        seen = np.zeros(n, dtype=bool)
        node, steps = 0, 0
        while node >= 0:
            assert not seen[node]
            seen[node] = True
            steps += 1
            kids, axis = _children(nodes, node)
            inner = [link for link, _, is_leaf in kids if not is_leaf]
            if len(inner) == 2:
                node = kids[(octant >> axis) & 1][0]
            elif inner:
                node = inner[0]
            else:
                node = _miss(nodes, node, octant)
        assert seen.all() and steps == n
        assert _miss(nodes, 0, octant) == -1


def _walk(nodes, gbox, order, grid, tris, po, o, d):
    """The kernel's walk (pt_kernels.hpp: bvh_ray / bvh_slab / bvh_step) for one ray; leaf tests through the oracle."""
    f = np.float32
    o32, d32 = o.astype(f), d.astype(f)
    origin, step, prune = grid[0:3], grid[3:6], f(grid[7])
    spec_pad = po.mesh_pad(tris.view(po.TRI_DT))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        off_axis = np.where(np.abs(d32) < f(1e-20), np.copysign(f(1e-20), d32), d32).astype(f)
        inv = (f(1) / off_axis).astype(f)
        kk = (step * inv).astype(f)
        bb = ((origin - o32).astype(f) * inv).astype(f)
    octant = int(d32[0] < 0) | (int(d32[1] < 0) << 1) | (int(d32[2] < 0) << 2)
    state = {"best": f(np.finfo(f).max), "i": -1}
    bary = po.Vec3()

    def hit(lo, hi):
        with np.errstate(invalid="ignore", over="ignore"):
            t1, t2 = lo * kk.astype(np.float64) + bb, hi * kk.astype(np.float64) + bb
            reach = state["best"] + prune
        tn = max(np.fmax.reduce(np.fmin(t1, t2)), 0.0)
        tf = np.fmin.reduce(np.fmax(t1, t2))
        return tn <= tf and tn <= reach

    def leaf(first, cnt):
        for s in range(first, first + cnt):
            k = int(order[s])
            T = tris[k]
            if po.lib().pto_ray_triangle(po.vec3(o32), po.vec3(d32), po.vec3(T["v0"]), po.vec3(T["v1"]),
                                         po.vec3(T["v2"]), bary):
                tz = f(bary.z)
                if tz > 0 and (state["best"] > tz or (state["best"] == tz and k < state["i"])) and \
                        po.lib().pto_tri_point_ok(po.vec3(o32), po.vec3(d32), tz, tris[k:k + 1].ctypes.data, spec_pad):
                    state["best"], state["i"] = tz, k

    node, visited = 0, 0
    while node >= 0:
        visited += 1
        kids, axis = _children(nodes, node)
        go = []
        for c, (link, count, is_leaf) in enumerate(kids):
            h = hit(gbox[node, c, 0], gbox[node, c, 1])
            if h and is_leaf:
                leaf(link, count)
            go.append(h and not is_leaf)
        if go[0] and go[1]:
            node = kids[(octant >> axis) & 1][0]
        elif go[0] or go[1]:
            node = kids[1 if go[1] else 0][0]
        else:
            node = _miss(nodes, node, octant)
    return state["i"], state["best"], visited


def test_walk_finds_the_naive_winner(pt, po):
    tris = pt.meshes.uv_sphere(n_lat=40, n_lon=80)                    # 6240 triangles
    nodes, gbox, order, grid = _tree(pt, tris)
    amax = max(1.0, float(np.abs(np.stack([tris["v0"], tris["v1"], tris["v2"]])).max()))
    assert grid[6] == np.ldexp(np.float32(amax), -13) and grid[7] == 0                  # pad = 2 * the spec's pad; no prune margin
    assert po.mesh_pad(tris.view(po.TRI_DT)) == np.ldexp(np.float32(amax), -14)
    rng = np.random.default_rng(5)
    verts = np.stack([tris["v0"], tris["v1"], tris["v2"]], axis=1)
    rays = np.zeros(600, dtype=po.PATH_DT)
    for k in range(len(rays)):
        o = rng.uniform(-5, 8, 3)
        kind = k % 4
        T = verts[rng.integers(len(tris))].astype(np.float64)
        if kind == 0:
            target = T[rng.integers(3)]                               # straight at a vertex (shared by ~6 triangles)
        elif kind == 1:
            w = rng.uniform(0, 1)
            target = w * T[0] + (1 - w) * T[1]                        # at a point of an edge
        elif kind == 2:
            w = rng.dirichlet((1, 1, 1))
            target = w @ T                                            # interior
        else:
            target = np.array([1.5, 3.0, 1.0]) + rng.normal(size=3) * 2.0   # may miss
        if k % 16 == 15:
            o = np.array([1.5, 3.0, 1.0]) + rng.normal(size=3) * 0.3  # from inside the sphere: back faces only
        d = target - o
        d /= np.linalg.norm(d)
        if k % 8 == 5:                                                # exactly axis-parallel, through or past the mesh
            ax = rng.integers(3)
            d = np.zeros(3)
            d[ax] = rng.choice([-1.0, 1.0])
            if k % 16 == 5:
                o = target - d * rng.uniform(2, 6)
        rays["origin"][k], rays["direction"][k] = o, d
    want_i, want_t = po.mesh_winners(tris.view(po.TRI_DT), rays)
    assert (want_i >= 0).sum() > 250 and (want_i < 0).sum() > 30
    total = 0
    for k in range(len(rays)):
        gi, gt, visited = _walk(nodes, gbox, order, grid, tris, po, rays["origin"][k], rays["direction"][k])
        total += visited
        assert visited < 0.2 * len(nodes), k      # no ray degenerates into a sweep of the tree (axis-parallel ones included)
        assert gi == want_i[k], k
        if gi >= 0:
            assert np.float32(gt).tobytes() == np.float32(want_t[k]).tobytes()
    assert total / len(rays) < 0.05 * len(nodes)                      # it actually culls
    print("records visited per ray: %.1f of %d" % (total / len(rays), len(nodes)))


@pytest.mark.parametrize("seed", [17, 37, 101])
def test_walk_on_grazing_soups(pt, po, seed):
    """Rays that run (almost) inside the plane of metre-sized triangles: glm's single-precision test accepts noise
    there (seeds 17 and 37 hold rays whose unfiltered "hit point" lies 2-5 units outside the triangle's box, and
    whose reported bary.z is far smaller than the distance to the box -- the walk without the spec's hit-point test
    lost those to pruning).  With the test, walk == loop."""
    import mesh_cases
    rng = np.random.default_rng(seed)
    tris = mesh_cases.soup(pt.TRI_DT, rng)
    rng.integers(1, 5)                                                # tests/tools/fuzz_gpu.py draws a material here
    k = min(len(tris), 1500)
    origin, d, graze = mesh_cases.aimed_rays(tris, rng, k)
    nodes, gbox, order, grid = _tree(pt, tris)
    rays = np.zeros(k, dtype=po.PATH_DT)
    rays["origin"], rays["direction"] = origin.astype(np.float32), d.astype(np.float32)
    want_i, want_t = po.mesh_winners(tris.view(po.TRI_DT), rays)
    # the filter is not vacuous on these inputs: the raw glm loop picks a different winner for some grazing ray
    f = np.float32
    raw_differs = 0
    bary = po.Vec3()
    sel = np.nonzero(graze)[0][:120].tolist() + np.nonzero(~graze)[0][:40].tolist()
    extra = {17: [18, 807], 37: [59]}.get(seed, [])
    for kk in sorted(set(sel + extra)):
        gi, gt, _ = _walk(nodes, gbox, order, grid, tris, po, rays["origin"][kk], rays["direction"][kk])
        assert gi == want_i[kk], (seed, kk)
        if gi >= 0:
            assert f(gt).tobytes() == f(want_t[kk]).tobytes()
        if kk in extra:
            best, hit = f(np.finfo(f).max), -1
            for i in range(len(tris)):
                T = tris[i]
                if po.lib().pto_ray_triangle(po.vec3(rays["origin"][kk]), po.vec3(rays["direction"][kk]), po.vec3(T["v0"]),
                                             po.vec3(T["v1"]), po.vec3(T["v2"]), bary) and 0 < bary.z < best:
                    best, hit = f(bary.z), i
            raw_differs += hit != want_i[kk]
    if extra:
        assert raw_differs > 0


def test_degenerate_inputs(pt):
    # empty mesh, one triangle, coincident triangles (no split separates them)
    nodes, gbox, order, grid = _tree(pt, np.zeros(0, dtype=pt.TRI_DT))
    # both children are leaves of ZERO triangles with a point-sized box (centre 0, half extent 0): whatever hits it queues nothing
    assert len(nodes) == 1 and (gbox[0, :, 0] == 0).all() and (gbox[0, :, 1] == 0).all()
    assert [c[1:] for c in _children(nodes, 0)[0]] == [(0, True), (0, True)]
    assert all(_miss(nodes, 0, o) == -1 for o in range(8))
    one = np.zeros(1, dtype=pt.TRI_DT)
    one["v1"][0], one["v2"][0] = (1, 0, 0), (0, 1, 0)
    nodes, gbox, order, grid = _tree(pt, one)
    assert len(nodes) == 1 and order.tolist() == [0]
    assert [c[1:] for c in _children(nodes, 0)[0]] == [(1, True), (0, True)]
    box = grid[0:3] + gbox[0, 0] * grid[3:6]
    assert (box[0] < 0).all() and (box[1] > [1, 1, 0]).all() and (gbox[0, 1] == 0).all()
    same = np.repeat(one, 37)
    nodes, gbox, order, grid = _tree(pt, same)
    assert sorted(order.tolist()) == list(range(37))
    counts = [c[1] for k in range(len(nodes)) for c in _children(nodes, k)[0] if c[2]]
    assert sum(counts) == 37 and max(counts) <= 4
