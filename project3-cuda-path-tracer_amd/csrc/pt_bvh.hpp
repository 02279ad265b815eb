// pt_bvh.hpp -- host-side builder of the bounding-volume hierarchy that culls triangle tests
// (the assignment's "hierarchical spatial data structure", INSTRUCTION.md:129-139,
// 218-240 -- the reference has none).  Included by ptmi355.hip only; no device code here.
//
// Layout: a binary tree (binned surface-area splits, leaves of 1..LEAF_MAX = 2
// triangles) stored as one 64-byte record per INTERNAL node that carries the boxes of BOTH
// children, so one fetch per step decides two boxes.  A walk is bound by how many address-divergent
// vector loads the texture path can take, so the record is made small: box planes are 16-bit
// grid coordinates over the mesh's bounds (rounded outwards; grid step = extent / 65533), two
// 16-B loads bring both boxes and both links, a third 4-B load the miss link.  A box is stored as CENTRE and HALF
// EXTENT on that grid, m = (lo + hi) >> 1 and e = hi - m + 1, i.e. the planes m -+ e, which contain [lo, hi] with a
// grid step to spare on either side: the slab test then is three fused multiply-adds per axis -- t_mid = m k + b,
// t_mid -+ e |k| -- instead of two plus a v_min and a v_max (which issue at half an fma's rate), and m needs no
// integer-to-float conversion (pt_k_bvh.hpp: bvh_slab; the spare step pays for the rounding of that trick).
//   [0..2]   left child box   m.x | m.y << 16,  m.z | e.x << 16,  e.y | e.z << 16
//   [3..5]   right child box, same packing
//   [6]      left link  | info << 24      [7] right link | info << 24
//              info = count | leaf << 3 | split axis of THIS node << 4   (axis only in the left info)
//              leaf child: link = first record in the leaf-ordered triangle array, count = 0..LEAF_MAX
//              internal child: link = record index of that child                    (links < 2^24)
//   [8..15]  miss link per ray-direction octant (bit k set when dir[k] < 0): the record to continue
//            with when this subtree is finished; -1 ends the walk
// world plane = origin[axis] + grid * step[axis]  (Tree::origin / Tree::step).
// The kernel walks the tree WITHOUT a stack.  At a record it tests both child boxes, intersects
// the triangles of hit leaf children at once, and continues with a hit internal child -- the
// nearer one (by the sign of dir[axis]) when both are hit; otherwise it follows miss[octant].
// The miss link of a near internal child is its internal sibling, that of a far or only internal
// child is the parent's: for a fixed octant the links spell out one depth-first, near-first order.
// A sibling reached through a miss link is entered without knowing whether its own box was hit;
// child boxes lie inside the parent's, so a missed sibling simply fails both child tests.
//
// Culling must never change which triangle wins (the loop over all triangles is the specification), and
// glm::intersectRayTriangle in single precision is no help: for a ray that runs almost inside the plane of
// a large triangle it accepts barycentrics that are rounding noise, with a "hit point" o + d*tz metres away
// from the triangle -- no box around the triangle can promise to contain that.  The completion spec (DESIGN.md
// section 3 "Triangles", oracle/ptoracle.c: pto_tri_point_ok) therefore counts a triangle hit only when the
// point fl(o + fl(d*tz)) lies inside the triangle's bounding box widened by spec_pad = 2^-14 * max(1, largest
// finite |coordinate| of the mesh), and that is exactly the handle a hierarchy needs.  Proof that the walk
// returns the loop's winner (T, tz):
//   * every box on the path from the root to T's leaf contains box(T) widened by `pad` = 2 * spec_pad (the
//     grid rounding only adds to it), so the accepted point lies inside each of them with spec_pad to spare
//     on every side;
//   * the kernel's slab test (v_rcp + fused multiply-adds on grid coordinates, pt_k_bvh.hpp: bvh_slab)
//     places a box plane within 4.5 * 2^-23 * (|plane - o|) + step / 2 of where it is (the step / 2: the centre enters
//     the fused multiply-add as the float 2^23 + m and 2^23 k is folded into the ray's offset b, whose rounding is half
//     a grid step -- the planes m -+ e carry a whole one for it), i.e. below spec_pad for every ray
//     origin within ~128 * max(1, amax) of the mesh; pt_init knows the bound R on |origin|_1 of every ray that is
//     not `wild` (wild rays -- only a caller's own, through pt_intersect_once -- take the whole tree: pt_k_bvh.hpp: bvh_walk;
//     the bound covers the scene and the camera, and pt_set_camera re-derives it and REBUILDS the trees when the
//     camera leaves it) and widens the pad to spec_pad + 8 * 2^-23 * (R + amax) where that is more (build(): a far
//     camera) -- so along each axis the computed entry parameter is <= tz <= the computed exit
//     parameter: the box is hit, and it is entered no later than tz;
//   * tz is the smallest accepted parameter of the whole mesh, so the running best never drops below it and
//     "entry <= best" cannot prune the path (`prune` = 0: no distance margin is needed any more);
//   * triangles with a non-finite coordinate never pass glm's test (a or tz is NaN), so clamping their boxes
//     to the grid loses nothing.
// Ties (equal tz) are visited for the same reason and resolved by the original index, as the loop does.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace ptbvh {

constexpr int NODE_WORDS = 16;
constexpr int GRID_MAX = 65535;
constexpr int LINK_BITS = 24;
#ifndef PT_LEAF_MAX
#define PT_LEAF_MAX 2
#endif
constexpr int LEAF_MAX = PT_LEAF_MAX;    // triangles per leaf (the count field has 3 bits)
constexpr int SAH_BINS = 16;
constexpr int INFO_LEAF = 8;         // bit 3 of an info word

struct Box {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
    void grow(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Tree {
    std::vector<float> nodes;        // NODE_WORDS per record, record 0 is the root
    std::vector<int32_t> order;      // leaf-ordered triangle slots -> index into the caller's triangle range
    float pad = 0.0f, prune = 0.0f;
    float spec_pad = 0.0f;           // pad of the spec's hit-point test (every triangle record carries it in word 10)
    float origin[3] = {0, 0, 0}, step[3] = {1, 1, 1};   // world plane = origin + grid * step
    int depth = 0;
    int num_nodes() const { return (int)(nodes.size() / NODE_WORDS); }
};

namespace detail {

struct Split {                       // the binary tree before it is laid out
    Box box;
    int left = -1;                   // children left, left + 1; -1: leaf
    int lo = 0, count = 0;           // leaf: slots [lo, lo + count)
    int axis = 0;
};

struct Work {
    std::vector<Box> tbox;
    std::vector<float> cen;          // 3 per triangle
    std::vector<int32_t> idx;        // permutation being partitioned
    std::vector<Split> split;
};

inline void set_i(float *w, int32_t v) { memcpy(w, &v, 4); }

// builds the subtree of idx[lo, hi) into split[self]; returns its depth
inline int build(Work &w, int self, int lo, int hi, int level) {
    Box b; b.reset();
    Box cb; cb.reset();
    for (int k = lo; k < hi; ++k) { b.grow(w.tbox[w.idx[k]]); cb.grow(&w.cen[3 * (size_t)w.idx[k]]); }
    w.split[self].box = b;
    const int count = hi - lo;
    int axis = 0;
    {
        float ext = -1.0f;
        for (int k = 0; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > ext) { ext = cb.hi[k] - cb.lo[k]; axis = k; }
    }
    int mid = -1;
    if (count > LEAF_MAX) {
        // binned surface-area heuristic over the three axes (below level 40: plain median splits, so the
        // recursion depth stays bounded whatever the input looks like)
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3 && level < 40; ++ax) {
            const float c0 = cb.lo[ax], c1 = cb.hi[ax];
            if (!(c1 > c0)) continue;
            Box bins[SAH_BINS]; int cnt[SAH_BINS];
            for (int k = 0; k < SAH_BINS; ++k) { bins[k].reset(); cnt[k] = 0; }
            const float sc = (float)SAH_BINS / (c1 - c0);
            for (int k = lo; k < hi; ++k) {
                int bi = (int)((w.cen[3 * (size_t)w.idx[k] + ax] - c0) * sc);
                bi = std::max(0, std::min(SAH_BINS - 1, bi));
                bins[bi].grow(w.tbox[w.idx[k]]); cnt[bi]++;
            }
            float right_area[SAH_BINS]; int right_cnt[SAH_BINS];
            Box acc; acc.reset(); int c = 0;
            for (int k = SAH_BINS - 1; k > 0; --k) {
                if (cnt[k]) acc.grow(bins[k]);
                c += cnt[k];
                right_area[k] = c ? acc.half_area() : 0.0f; right_cnt[k] = c;
            }
            acc.reset(); c = 0;
            for (int k = 0; k < SAH_BINS - 1; ++k) {
                if (cnt[k]) acc.grow(bins[k]);
                c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = k; }
            }
        }
        if (best_axis >= 0) {
            axis = best_axis;
            const float c0 = cb.lo[axis], sc = (float)SAH_BINS / (cb.hi[axis] - cb.lo[axis]);
            auto it = std::partition(w.idx.begin() + lo, w.idx.begin() + hi, [&](int32_t t) {
                int bi = (int)((w.cen[3 * (size_t)t + axis] - c0) * sc);
                bi = std::max(0, std::min(SAH_BINS - 1, bi));
                return bi <= best_bin;
            });
            mid = (int)(it - w.idx.begin());
        }
        if (mid <= lo || mid >= hi) {
            // all centroids coincide (or the bins could not separate them): split the list in half
            mid = lo + count / 2;
            std::nth_element(w.idx.begin() + lo, w.idx.begin() + mid, w.idx.begin() + hi, [&](int32_t a, int32_t c) {
                const float ca = w.cen[3 * (size_t)a + axis], cc = w.cen[3 * (size_t)c + axis];
                return ca < cc || (ca == cc && a < c);
            });
        }
    }
    if (mid < 0) {                    // leaf
        w.split[self].lo = lo; w.split[self].count = count; w.split[self].axis = axis;
        return 1;
    }
    const int left = (int)w.split.size();
    w.split.resize(w.split.size() + 2);
    w.split[self].left = left; w.split[self].axis = axis;
    const int dl = build(w, left, lo, mid, level + 1);
    const int dr = build(w, left + 1, mid, hi, level + 1);
    return 1 + std::max(dl, dr);
}

// Record numbering: rec[s] = record index of internal split node s.  The first TOP_RECORDS records are the
// internal nodes most rays visit -- grown greedily from the root, always taking the open node with the largest
// box surface (the surface-area heuristic's visit probability) -- so that ANY prefix of the record array is a
// connected top of the tree the walk kernel can keep in LDS (pt_kernels.hpp: k_mesh); the rest follows
// depth-first.  Parents are numbered before their children either way.
constexpr int TOP_RECORDS = 512;
inline void number_rest(const Work &w, int s, std::vector<int32_t> &rec, int &next) {
    if (w.split[s].left < 0) return;
    if (rec[(size_t)s] < 0) rec[(size_t)s] = next++;
    number_rest(w, w.split[s].left, rec, next);
    number_rest(w, w.split[s].left + 1, rec, next);
}
inline void number(const Work &w, int root, std::vector<int32_t> &rec, int &next) {
    if (w.split[root].left < 0) return;
    auto weight = [&](int s) { const float a = w.split[s].box.half_area(); return std::isfinite(a) ? a : 3.0e38f; };
    std::vector<std::pair<float, int>> open;                 // max-heap on (surface, -node)
    open.push_back({weight(root), -root});
    while (!open.empty() && next < TOP_RECORDS) {
        std::pop_heap(open.begin(), open.end());
        const int s = -open.back().second;
        open.pop_back();
        rec[(size_t)s] = next++;
        for (int c = w.split[s].left; c <= w.split[s].left + 1; ++c)
            if (w.split[c].left >= 0) { open.push_back({weight(c), -c}); std::push_heap(open.begin(), open.end()); }
    }
    number_rest(w, root, rec, next);
}

inline void set_u(float *w, uint32_t v) { memcpy(w, &v, 4); }

// grid coordinate of a box plane, rounded outwards with one step to spare (the kernel evaluates
// origin + grid * step through a fused form that can be an ulp or two off)
inline uint32_t grid_lo(float x, float o, float s) {
    const double g = std::floor(((double)x - (double)o) / (double)s) - 1.0;
    return (uint32_t)std::max(0.0, std::min((double)GRID_MAX, g));
}
inline uint32_t grid_hi(float x, float o, float s) {
    const double g = std::ceil(((double)x - (double)o) / (double)s) + 1.0;
    return (uint32_t)std::max(0.0, std::min((double)GRID_MAX, g));
}

inline void pack_box(float *r, const Box &b, float pad, const Tree &t, bool empty) {
    uint32_t m[3], e[3];
    for (int a = 0; a < 3; ++a) {
        // an empty child is a leaf of zero triangles: whatever hits its (point-sized) box queues nothing
        const uint32_t lo = empty ? 0u : grid_lo(b.lo[a] - pad, t.origin[a], t.step[a]);
        const uint32_t hi = empty ? 0u : grid_hi(b.hi[a] + pad, t.origin[a], t.step[a]);
        m[a] = (lo + hi) >> 1;                                   // m - e <= lo - 1,  m + e = hi + 1
        e[a] = empty ? 0u : std::min<uint32_t>((uint32_t)GRID_MAX, hi - m[a] + 1u);
    }
    set_u(&r[0], m[0] | (m[1] << 16)); set_u(&r[1], m[2] | (e[0] << 16)); set_u(&r[2], e[1] | (e[2] << 16));
}

inline void emit(const Work &w, const std::vector<int32_t> &rec, int s, const int32_t miss[8], float pad, Tree &out) {
    const Split &n = w.split[s];
    float *r = &out.nodes[(size_t)rec[(size_t)s] * NODE_WORDS];
    const int kid[2] = {n.left, n.left + 1};
    bool inner[2];
    for (int c = 0; c < 2; ++c) {
        const Split &k = w.split[kid[c]];
        pack_box(&r[3 * c], k.box, pad, out, false);
        inner[c] = k.left >= 0;
        const uint32_t link = (uint32_t)(inner[c] ? rec[(size_t)kid[c]] : k.lo);
        const uint32_t info = (uint32_t)((inner[c] ? 0 : (k.count | INFO_LEAF)) | (c == 0 ? n.axis << 4 : 0));
        set_u(&r[6 + c], link | (info << LINK_BITS));
    }
    for (int o = 0; o < 8; ++o) set_i(&r[8 + o], miss[o]);
    int32_t m[2][8];
    for (int o = 0; o < 8; ++o) {
        const int near = (o >> n.axis) & 1;                  // dir[axis] < 0: the upper (right) child is nearer
        m[near][o] = inner[near ^ 1] ? rec[(size_t)kid[near ^ 1]] : miss[o];
        m[near ^ 1][o] = miss[o];
    }
    for (int c = 0; c < 2; ++c)
        if (inner[c]) emit(w, rec, kid[c], m[c], pad, out);
}

}  // namespace detail

// pad of the spec's hit-point test for a mesh of `count` triangles, 9 floats each (oracle/ptoracle.c: pto_mesh_pad)
inline float spec_pad(const float *v, int count) {
    float amax = 0.0f;
    for (size_t k = 0; k < 9 * (size_t)count; ++k) {
        const float m = std::fabs(v[k]);
        if (m <= 3.402823466e+38f && m > amax) amax = m;
    }
    return std::ldexp(std::max(1.0f, amax), -14);
}

// largest finite |coordinate| of the mesh
inline float coord_max(const float *v, int count) {
    float amax = 0.0f;
    for (size_t k = 0; k < 9 * (size_t)count; ++k) {
        const float m = std::fabs(v[k]);
        if (m <= 3.402823466e+38f && m > amax) amax = m;
    }
    return amax;
}

// v: `count` triangles, 9 floats each (v0 v1 v2, world space).  origin_bound: the |origin|_1 bound of the rays that
// will walk the tree (pt_init: SceneDev::rmax, which covers the scene and the camera and is re-derived when the
// camera leaves it); < 0: the bound the pad of 2 * spec_pad covers by itself, ~128 * max(1, amax).
inline void build(const float *v, int count, Tree &out, double origin_bound = -1.0) {
    out.nodes.clear(); out.order.clear(); out.depth = 0;
    detail::Work w;
    w.tbox.resize((size_t)count); w.cen.resize(3 * (size_t)count); w.idx.resize((size_t)count);
    for (int i = 0; i < count; ++i) {
        Box &b = w.tbox[i]; b.reset();
        for (int k = 0; k < 3; ++k) b.grow(v + 9 * (size_t)i + 3 * k);
        for (int k = 0; k < 3; ++k) {
            w.cen[3 * (size_t)i + k] = 0.5f * (b.lo[k] + b.hi[k]);
        }
        w.idx[i] = i;
    }
    out.spec_pad = spec_pad(v, count);
    out.pad = 2.0f * out.spec_pad;
    if (origin_bound >= 0.0) {
        // the slab test misplaces a plane by at most 4 * 2^-23 * |plane - o| <= 4 * 2^-23 * (amax + pad + |o|); the
        // accepted point must keep its spec_pad of clearance with that on top: twice the error, for the rounding of
        // the planes to the grid and of this sum itself
        const double slab = 8.0 * 0x1p-23 * (origin_bound + (double)coord_max(v, count) + 4.0 * (double)out.spec_pad);
        const float need = (float)((double)out.spec_pad + slab) * 1.0000005f;
        if (std::isfinite(need) && need > out.pad) out.pad = need;
    }
    out.prune = 0.0f;
    w.split.resize(1);
    if (count > 0) out.depth = detail::build(w, 0, 0, count, 0);
    // the grid spans the padded bounds of the whole mesh with a few steps of margin on either side
    for (int a = 0; a < 3; ++a) {
        float lo = 0.0f, hi = 0.0f;
        if (count > 0 && std::isfinite(w.split[0].box.lo[a]) && std::isfinite(w.split[0].box.hi[a])) {
            lo = w.split[0].box.lo[a]; hi = w.split[0].box.hi[a];
        }
        const float ext = (hi - lo) + 2.0f * out.pad;
        out.step[a] = std::max(ext / (float)(GRID_MAX - 8), 1e-30f);
        out.origin[a] = (lo - out.pad) - 4.0f * out.step[a];
    }
    const int32_t end[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    if (count > 0 && w.split[0].left >= 0) {
        std::vector<int32_t> rec(w.split.size(), -1);
        int next = 0;
        detail::number(w, 0, rec, next);
        out.nodes.assign((size_t)next * NODE_WORDS, 0.0f);
        detail::emit(w, rec, 0, end, out.pad, out);
    } else {
        // at most LEAF_MAX triangles: one record whose left child is that leaf (its box when there is one) and whose
        // right child is an EMPTY leaf: pack_box gives an empty child a point box at the grid's origin (m = 0, e = 0) --
        // a ray through that point "hits" it, and finds a leaf of zero triangles: nothing to test.  The invariant the
        // walk relies on is that an empty child always carries INFO_LEAF and count 0 (pack_box; test_bvh_cpu.py)
        out.nodes.assign(NODE_WORDS, 0.0f);
        float *r = out.nodes.data();
        detail::pack_box(&r[0], w.split[0].box, out.pad, out, count == 0);
        detail::pack_box(&r[3], w.split[0].box, out.pad, out, true);
        detail::set_u(&r[6], (uint32_t)((count | INFO_LEAF) << LINK_BITS));
        detail::set_u(&r[7], (uint32_t)(INFO_LEAF << LINK_BITS));
        for (int o = 0; o < 8; ++o) detail::set_i(&r[8 + o], -1);
    }
    out.order = w.idx;
}

}  // namespace ptbvh
