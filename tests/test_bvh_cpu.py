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
    nodes, order = pt.binding.bvh_build(tris)
    ints = nodes.view(np.int32)
    return nodes, ints, order


LEAF = 8


def _children(ints, k):
    """[(link, count, is_leaf)] of record k; split axis"""
    kids = [(int(ints[k, 12 + 2 * c]), int(ints[k, 13 + 2 * c]) & 7, bool(ints[k, 13 + 2 * c] & LEAF)) for c in (0, 1)]
    return kids, (int(ints[k, 13]) >> 4) & 3


@pytest.mark.parametrize("size", [(8, 16), (30, 60), (97, 521)])
def test_tree_structure(pt, size):
    tris = pt.meshes.uv_sphere(n_lat=size[0], n_lon=size[1])
    nodes, ints, order = _tree(pt, tris)
    n = len(nodes)
    assert sorted(order.tolist()) == list(range(len(tris)))          # every triangle in exactly one leaf slot
    verts = np.stack([tris["v0"], tris["v1"], tris["v2"]], axis=1)      # (T, 3, 3)
    leaves, parents = [], np.zeros(n, dtype=np.int32)
    for k in range(n):
        kids, axis = _children(ints, k)
        assert axis <= 2
        for c, (link, count, is_leaf) in enumerate(kids):
            lo, hi = nodes[k, 6 * c:6 * c + 3], nodes[k, 6 * c + 3:6 * c + 6]
            if is_leaf:
                assert 1 <= count <= 4
                leaves.append((link, count))
                v = verts[order[link:link + count]].reshape(-1, 3)
                assert (v.min(axis=0) > lo).all() and (v.max(axis=0) < hi).all()
            else:
                assert k < link < n and count == 0                    # records are laid out parent first
                parents[link] += 1
                for cc in (0, 1):                                     # a child's boxes lie inside its own box
                    assert (nodes[link, 6 * cc:6 * cc + 3] >= lo).all() and (nodes[link, 6 * cc + 3:6 * cc + 6] <= hi).all()
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
            kids, axis = _children(ints, node)
            inner = [link for link, _, is_leaf in kids if not is_leaf]
            if len(inner) == 2:
                node = kids[(octant >> axis) & 1][0]
            elif inner:
                node = inner[0]
            else:
                node = int(ints[node, 16 + octant])
        assert seen.all() and steps == n
        assert ints[0, 16 + octant] == -1


def _walk(nodes, ints, order, tris, po, o, d, prune):
    """The kernel's walk (pt_kernels.hpp: bvh_walk) for one ray; leaf tests through the oracle."""
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (np.float32(1) / d32).astype(np.float32)
        off = (-(o32 * inv)).astype(np.float32)
    octant = int(d32[0] < 0) | (int(d32[1] < 0) << 1) | (int(d32[2] < 0) << 2)
    state = {"best": np.float32(np.finfo(np.float32).max), "i": -1}
    bary = po.Vec3()

    def hit(lo, hi):
        with np.errstate(invalid="ignore", over="ignore"):
            t1, t2 = lo * inv + off, hi * inv + off
            reach = state["best"] + np.float32(prune)
        tn = max(np.fmax.reduce(np.fmin(t1, t2)), 0.0)
        tf = np.fmin.reduce(np.fmax(t1, t2))
        return tn <= tf and tn <= reach

    def leaf(first, cnt):
        for s in range(first, first + cnt):
            k = int(order[s])
            T = tris[k]
            if po.lib().pto_ray_triangle(po.vec3(o32), po.vec3(d32), po.vec3(T["v0"]), po.vec3(T["v1"]),
                                         po.vec3(T["v2"]), bary):
                tz = np.float32(bary.z)
                if tz > 0 and (state["best"] > tz or (state["best"] == tz and k < state["i"])):
                    state["best"], state["i"] = tz, k

    node, visited = 0, 0
    while node >= 0:
        visited += 1
        kids, axis = _children(ints, node)
        go = []
        for c, (link, count, is_leaf) in enumerate(kids):
            h = hit(nodes[node, 6 * c:6 * c + 3], nodes[node, 6 * c + 3:6 * c + 6])
            if h and is_leaf:
                leaf(link, count)
            go.append(h and not is_leaf)
        if go[0] and go[1]:
            node = kids[(octant >> axis) & 1][0]
        elif go[0] or go[1]:
            node = kids[1 if go[1] else 0][0]
        else:
            node = int(ints[node, 16 + octant])
    return state["i"], state["best"], visited


def test_walk_finds_the_naive_winner(pt, po):
    tris = pt.meshes.uv_sphere(n_lat=40, n_lon=80)                    # 6240 triangles
    nodes, ints, order = _tree(pt, tris)
    amax = max(1.0, float(np.abs(np.stack([tris["v0"], tris["v1"], tris["v2"]])).max()))
    prune = 16.0 * np.ldexp(np.float32(amax), -13)
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
        rays["origin"][k], rays["direction"][k] = o, d
    want_i, want_t = po.mesh_winners(tris.view(po.TRI_DT), rays)
    assert (want_i >= 0).sum() > 250 and (want_i < 0).sum() > 30
    total = 0
    for k in range(len(rays)):
        gi, gt, visited = _walk(nodes, ints, order, tris, po, rays["origin"][k], rays["direction"][k], prune)
        total += visited
        assert gi == want_i[k], k
        if gi >= 0:
            assert np.float32(gt).tobytes() == np.float32(want_t[k]).tobytes()
    assert total / len(rays) < 0.05 * len(nodes)                      # it actually culls
    print("records visited per ray: %.1f of %d" % (total / len(rays), len(nodes)))


def test_degenerate_inputs(pt):
    # empty mesh, one triangle, coincident triangles (no split separates them)
    nodes, order = pt.binding.bvh_build(np.zeros(0, dtype=pt.TRI_DT))
    ints = nodes.view(np.int32)
    assert len(nodes) == 1 and (nodes[0, 0:3] > nodes[0, 3:6]).all() and (nodes[0, 6:9] > nodes[0, 9:12]).all()
    assert ints[0, 13] == LEAF and ints[0, 15] == LEAF and (ints[0, 16:24] == -1).all()
    one = np.zeros(1, dtype=pt.TRI_DT)
    one["v1"][0], one["v2"][0] = (1, 0, 0), (0, 1, 0)
    nodes, order = pt.binding.bvh_build(one)
    ints = nodes.view(np.int32)
    assert len(nodes) == 1 and order.tolist() == [0] and ints[0, 13] == (LEAF | 1) and ints[0, 15] == LEAF
    assert (nodes[0, 0:3] < 0).all() and (nodes[0, 3:6] > 0).all() and (nodes[0, 6:9] > nodes[0, 9:12]).all()
    same = np.repeat(one, 37)
    nodes, order = pt.binding.bvh_build(same)
    ints = nodes.view(np.int32)
    assert sorted(order.tolist()) == list(range(37))
    counts = [(ints[k, 13 + 2 * c] & 7) for k in range(len(nodes)) for c in (0, 1) if ints[k, 13 + 2 * c] & LEAF]
    assert sum(counts) == 37 and max(counts) <= 4
