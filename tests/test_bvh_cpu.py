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


@pytest.mark.parametrize("size", [(8, 16), (30, 60), (97, 521)])
def test_tree_structure(pt, size):
    tris = pt.meshes.uv_sphere(n_lat=size[0], n_lon=size[1])
    nodes, ints, order = _tree(pt, tris)
    n = len(nodes)
    count = ints[:, 7] >> 2
    axis = ints[:, 7] & 3
    leaf = count > 0
    assert sorted(order.tolist()) == list(range(len(tris)))          # every triangle in exactly one leaf slot
    assert count[leaf].max() <= 4 and count[leaf].sum() == len(tris)
    assert (axis <= 2).all()
    # leaves tile the slot range
    firsts = ints[leaf, 6]
    o = np.argsort(firsts)
    assert firsts[o][0] == 0 and (firsts[o][1:] == (firsts[o] + count[leaf][o])[:-1]).all()
    # boxes: a leaf's box holds its triangles, a parent's box holds its children's
    verts = np.stack([tris["v0"], tris["v1"], tris["v2"]], axis=1)      # (T, 3, 3)
    for k in np.nonzero(leaf)[0]:
        v = verts[order[ints[k, 6]:ints[k, 6] + count[k]]].reshape(-1, 3)
        assert (v.min(axis=0) > nodes[k, 0:3]).all() and (v.max(axis=0) < nodes[k, 3:6]).all()
    inner = np.nonzero(~leaf)[0]
    for c in (0, 1):
        ch = ints[inner, 6] + c
        assert (ch > inner).all() and (ch < n).all()
        assert (nodes[ch, 0:3] >= nodes[inner, 0:3]).all() and (nodes[ch, 3:6] <= nodes[inner, 3:6]).all()
    # children are claimed by exactly one parent
    kids = np.concatenate([ints[inner, 6], ints[inner, 6] + 1])
    assert sorted(kids.tolist()) == list(range(1, n))
    # per octant: "hit everything" visits every node exactly once and ends; "miss the root" ends at once
    for octant in range(8):
        seen = np.zeros(n, dtype=bool)
        node, steps = 0, 0
        while node >= 0:
            assert not seen[node]
            seen[node] = True
            steps += 1
            if leaf[node]:
                node = ints[node, 8 + octant]
            else:
                node = ints[node, 6] + ((octant >> axis[node]) & 1)
        assert seen.all() and steps == n
        assert ints[0, 8 + octant] == -1


def _walk(nodes, ints, order, tris, po, o, d, prune):
    """The kernel's walk (pt_kernels.hpp: bvh_walk) for one ray; leaf tests through the oracle."""
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (np.float32(1) / d32).astype(np.float32)
        off = (-(o32 * inv)).astype(np.float32)
    octant = int(d32[0] < 0) | (int(d32[1] < 0) << 1) | (int(d32[2] < 0) << 2)
    best, best_i, node, visited = np.float32(np.finfo(np.float32).max), -1, 0, 0
    bary = po.Vec3()
    while node >= 0:
        visited += 1
        nd = nodes[node]
        with np.errstate(invalid="ignore", over="ignore"):
            t1 = nd[0:3] * inv + off
            t2 = nd[3:6] * inv + off
        tn = max(np.fmax.reduce(np.fmin(t1, t2)), 0.0)
        tf = np.fmin.reduce(np.fmax(t1, t2))
        nxt = int(ints[node, 8 + octant])
        with np.errstate(over="ignore"):
            reach = best + np.float32(prune)
        if tn <= tf and tn <= reach:
            cnt = ints[node, 7] >> 2
            if cnt == 0:
                nxt = int(ints[node, 6]) + ((octant >> (ints[node, 7] & 3)) & 1)
            else:
                for s in range(ints[node, 6], ints[node, 6] + cnt):
                    k = int(order[s])
                    T = tris[k]
                    if po.lib().pto_ray_triangle(po.vec3(o32), po.vec3(d32), po.vec3(T["v0"]), po.vec3(T["v1"]),
                                                 po.vec3(T["v2"]), bary):
                        tz = np.float32(bary.z)
                        if tz > 0 and (best > tz or (best == tz and k < best_i)):
                            best, best_i = tz, k
        node = nxt
    return best_i, best, visited


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


def test_degenerate_inputs(pt):
    # empty mesh, one triangle, coincident triangles (no split separates them)
    nodes, order = pt.binding.bvh_build(np.zeros(0, dtype=pt.TRI_DT))
    assert len(nodes) == 1 and (nodes[0, 0:3] > nodes[0, 3:6]).all()
    one = np.zeros(1, dtype=pt.TRI_DT)
    one["v1"][0], one["v2"][0] = (1, 0, 0), (0, 1, 0)
    nodes, order = pt.binding.bvh_build(one)
    assert len(nodes) == 1 and order.tolist() == [0]
    same = np.repeat(one, 37)
    nodes, order = pt.binding.bvh_build(same)
    ints = nodes.view(np.int32)
    assert sorted(order.tolist()) == list(range(37))
    assert ((ints[:, 7] >> 2)[(ints[:, 7] >> 2) > 0]).sum() == 37
