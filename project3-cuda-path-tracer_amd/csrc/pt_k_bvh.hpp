// pt_k_bvh.hpp -- the stackless walk of a mesh hierarchy (record layout: pt_bvh.hpp), as the inline walk and k_mesh step it
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

// Stackless walk of one mesh's hierarchy (record layout and link construction: pt_bvh.hpp).
// Per lane and step: fetch one 64-B record (both child boxes on a 16-bit grid + both links as two
// 16-B loads, the octant's miss link as a third), slab-test both boxes against [0, best + prune],
// intersect the triangles of hit leaf children, then continue with a hit internal child (the nearer
// one when both are hit) or follow the miss link.  The box test only has to be conservative (it
// decides which exact triangle tests run, never their outcome), so it uses v_rcp and fused
// multiply-adds; NaNs drop out of v_min/v_max, which errs towards visiting.  For a fixed octant the
// links spell out one depth-first order, so a walk visits a record at most once; `guard` bounds it
// for NaN rays all the same.
struct BvhRay {                       // what a walk keeps per (ray, mesh)
    f3 ro, rd;
    float kx, ky, kz, bx, by, bz;     // slab form over the mesh's grid: t = grid * k + b, with b stored as b - 2^23 k (bvh_slab)
    int oct;
};
// origin / step: the mesh's grid (world plane = origin + grid * step)
__device__ __forceinline__ BvhRay bvh_ray(f3 ro, f3 rd, f3 origin, f3 step) {
    BvhRay r;
    r.ro = ro; r.rd = rd;
    // A direction component of (nearly) zero would turn that axis' planes into inf - inf = NaN, which the
    // min/max drop: the box test would then ignore the axis and an axis-parallel ray would visit every record
    // in front of it.  The box tests use 1e-20 instead (the triangle tests keep the true direction): over any
    // distance in the scene the ray moves by far less than the box padding, so the test stays conservative.
    auto off_axis = [](float c) { return __builtin_fabsf(c) < 1e-20f ? __builtin_copysignf(1e-20f, c) : c; };
    const float ix = __builtin_amdgcn_rcpf(off_axis(rd.x)), iy = __builtin_amdgcn_rcpf(off_axis(rd.y)),
                iz = __builtin_amdgcn_rcpf(off_axis(rd.z));
    r.kx = step.x * ix; r.ky = step.y * iy; r.kz = step.z * iz;
    // b - 2^23 k in one rounding: bvh_slab multiplies k by the FLOAT 2^23 + m (the centre's 16 bits under the exponent
    // of 2^23: no conversion instruction), and the 2^23 k cancels exactly inside the fused multiply-add
    r.bx = __builtin_fmaf(-8388608.0f, r.kx, (origin.x - ro.x) * ix);
    r.by = __builtin_fmaf(-8388608.0f, r.ky, (origin.y - ro.y) * iy);
    r.bz = __builtin_fmaf(-8388608.0f, r.kz, (origin.z - ro.z) * iz);
    r.oct = (rd.x < 0.0f ? 1 : 0) | (rd.y < 0.0f ? 2 : 0) | (rd.z < 0.0f ? 4 : 0);
    return r;
}
// entry / exit parameters of the box packed in three dwords (pt_bvh.hpp: centre m and half extent e on the grid), clipped
// to t >= 0: per axis t_mid = (2^23 + m) k + (b - 2^23 k), entry / exit = t_mid -+ e |k|.  Eighteen fused multiply-adds
// for a record's two boxes and six conversions (e), where planes lo / hi took twelve of each and twelve v_min / v_max.
__device__ __forceinline__ void bvh_slab(const BvhRay &r, uint32_t w0, uint32_t w1, uint32_t w2, float &tn, float &tf) {
    const float mx = __uint_as_float((w0 & 0xffffu) | 0x4b000000u), my = __uint_as_float((w0 >> 16) | 0x4b000000u);
    const float mz = __uint_as_float((w1 & 0xffffu) | 0x4b000000u);
    const float ex = (float)(w1 >> 16), ey = (float)(w2 & 0xffffu), ez = (float)(w2 >> 16);
    const float tmx = __builtin_fmaf(mx, r.kx, r.bx), tmy = __builtin_fmaf(my, r.ky, r.by), tmz = __builtin_fmaf(mz, r.kz, r.bz);
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(-ex, __builtin_fabsf(r.kx), tmx), __builtin_fmaf(-ey, __builtin_fabsf(r.ky), tmy)),
                         __builtin_fmaxf(__builtin_fmaf(-ez, __builtin_fabsf(r.kz), tmz), 0.0f));
    tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(ex, __builtin_fabsf(r.kx), tmx), __builtin_fmaf(ey, __builtin_fabsf(r.ky), tmy)),
                         __builtin_fmaf(ez, __builtin_fabsf(r.kz), tmz));
}
#ifndef PT_BVH_TOP
#define PT_BVH_TOP 512
#endif
constexpr int BVH_TOP = PT_BVH_TOP;                  // records of the meshes' tree tops kept in LDS (pt_bvh.hpp numbers the most visited first)
constexpr int BVH_TOP_STRIDE = 20;            // dwords per record in LDS: 80 B apart, so random records spread over all banks
struct BvhRec { uint4 a, b; int miss; };   // the three loads of one record: boxes, boxes + links, miss[octant]
__device__ __forceinline__ BvhRec bvh_fetch(const float *__restrict__ nodes, int node, int oct) {
    const uint4 *n4 = reinterpret_cast<const uint4 *>(nodes + (size_t)node * BVH_NODE_WORDS);
    BvhRec rec;
    rec.a = n4[0]; rec.b = n4[1];
    rec.miss = reinterpret_cast<const int *>(n4)[8 + oct];
    return rec;
}
// k_mesh: the same record from the LDS copy of the tree tops when it is one of the mesh's first (top >> 16) records.
// A walk spends its first ~10 steps there; each such fetch is three LDS reads instead of three address-divergent
// global loads, which are what bounds the walk (64 distinct lines per instruction through the texture path).
// `all_nodes`: every mesh's records back to back (a wave-uniform pointer); the lane's tree starts at record `root`.  The byte offset
// is 32 bits (upload_bvh refuses more than 4 GiB of records), so the loads take the scalar base + vector offset form:
// no 64-bit address arithmetic per lane and step.  `tops_lds` = the LDS byte address of the tree tops: the two sources
// are read through their own address spaces (ds_read / global_load under the lanes' masks).  Written with generic
// pointers the compiler merged the two branches into ONE set of flat loads on a selected pointer -- every top record
// then went through the flat path's address check instead of a plain LDS read (round 3: found in the block listing).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4_t lds_u4_t;
typedef __attribute__((address_space(3))) const int lds_i32_t;
typedef __attribute__((address_space(1))) const u32x4_t glb_u4_t;
typedef __attribute__((address_space(1))) const int glb_i32_t;
__device__ __forceinline__ uint4 as_uint4(u32x4_t v) { return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ BvhRec bvh_fetch_top(const float *__restrict__ all_nodes, int root, uint32_t tops_lds, uint32_t top, int node, int oct) {
    BvhRec rec;
    if ((uint32_t)node < (top >> 16)) {
        const uint32_t off = tops_lds + ((top & 0xffffu) + (uint32_t)node) * (uint32_t)(BVH_TOP_STRIDE * 4);
        lds_u4_t *r = (lds_u4_t *)(size_t)off;
        rec.a = as_uint4(r[0]); rec.b = as_uint4(r[1]);
        rec.miss = ((lds_i32_t *)(size_t)off)[8 + oct];
        return rec;
    }
    const uint32_t off = ((uint32_t)root + (uint32_t)node) * (uint32_t)(BVH_NODE_WORDS * 4);
    const char *p = reinterpret_cast<const char *>(all_nodes) + off;
    rec.a = as_uint4(*(glb_u4_t *)p); rec.b = as_uint4(*(glb_u4_t *)(p + 16));
    rec.miss = *(glb_i32_t *)(p + 32 + 4 * oct);
    return rec;
}
// box tests of a fetched record: the record to continue with (< 0: the walk is over) and the hit leaf
// children as first | count << 24 (-1: none), to be tested by bvh_leaf
__device__ __forceinline__ int bvh_decide(const BvhRec &rec, const BvhRay &r, float reach, int &leaf_l, int &leaf_r,
                                          int *skip = nullptr) {
    const int link_l = (int)(rec.b.z & 0xffffffu), info_l = (int)(rec.b.z >> 24);
    const int link_r = (int)(rec.b.w & 0xffffffu), info_r = (int)(rec.b.w >> 24);
    float tn_l, tf_l, tn_r, tf_r;
    bvh_slab(r, rec.a.x, rec.a.y, rec.a.z, tn_l, tf_l);
    bvh_slab(r, rec.a.w, rec.b.x, rec.b.y, tn_r, tf_r);
    const bool hit_l = tn_l <= tf_l && tn_l <= reach;
    const bool hit_r = tn_r <= tf_r && tn_r <= reach;
    leaf_l = (hit_l && (info_l & 8)) ? (link_l | ((info_l & 7) << 24)) : -1;
    leaf_r = (hit_r && (info_r & 8)) ? (link_r | ((info_r & 7) << 24)) : -1;
    const bool go_l = hit_l && !(info_l & 8), go_r = hit_r && !(info_r & 8);
    const bool right_near = (r.oct >> ((info_l >> 4) & 3)) & 1;
    int next = rec.miss;
    if (go_l && go_r) next = right_near ? link_r : link_l;             // the far one follows through the near one's miss link
    else if (go_l) next = link_l;
    else if (go_r) next = link_r;
    if (skip) {
        // the walk enters one internal child while its internal sibling was missed: that sibling is where the
        // entered subtree's miss links lead if it is the far one -- the caller may skip it (straight to rec.miss)
        *skip = -1;
        const bool inner_l = !(info_l & 8), inner_r = !(info_r & 8);
        if (inner_l && inner_r && (go_l != go_r)) {
            const bool entered_right = go_r;
            if (entered_right == right_near) *skip = entered_right ? link_l : link_r;   // the missed one is the far child
        }
    }
    return next;
}
__device__ __forceinline__ void bvh_leaf(const float *__restrict__ btris, const BvhRay &r, int leaf, float &best, int &best_i) {
    const float4 *t4 = reinterpret_cast<const float4 *>(btris + (size_t)(leaf & 0xffffff) * TRI_WORDS);
    const int cnt = leaf >> 24;
#ifdef PT_LEAF_UNROLL
#pragma unroll PT_LEAF_UNROLL
#endif
    for (int k = 0; k < cnt; ++k) {
        const float4 P = t4[3 * k], Q = t4[3 * k + 1], S = t4[3 * k + 2];
        float tz;
        const f3 v0 = ptd::mk(P.x, P.y, P.z), e1 = ptd::mk(P.w, Q.x, Q.y), e2 = ptd::mk(Q.z, Q.w, S.x);
        if (ptd::ray_triangle(r.ro, r.rd, v0, e1, e2, tz)) {
            const int orig = __float_as_int(S.y);                        // index in the caller's triangle array
            if (tz > 0.0f && (best > tz || (best == tz && orig < best_i)) && ptd::tri_point_ok(r.ro, r.rd, tz, v0, e1, e2, S.z)) {
                best = tz; best_i = orig;
            }
        }
    }
}
// one step at record `node` (>= 0); returns the record to continue with, < 0 when the walk is over
__device__ __forceinline__ int bvh_step(const float *__restrict__ nodes, const float *__restrict__ btris,
                                        float prune, const BvhRay &r, int node, float &best, int &best_i) {
    const BvhRec rec = bvh_fetch(nodes, node, r.oct);
    int leaf_l, leaf_r;
    const int next = bvh_decide(rec, r, best + prune, leaf_l, leaf_r);
    if (leaf_l >= 0) bvh_leaf(btris, r, leaf_l, best, best_i);
    if (leaf_r >= 0) bvh_leaf(btris, r, leaf_r, best, best_i);
    return next;
}
// `grid`: origin xyz, step xyz of the mesh (six floats of its geom record)
// `wild`: the ray lies outside what the boxes' padding was derived for (pt_bvh.hpp: |origin|_1 beyond the scene's bound, a
// non-finite or absurdly scaled direction: cull_ray) -- only caller-supplied rays can (pt_intersect_once): paths start at the
// camera, which the bound covers, and continue from points of the scene.  Such a ray takes the whole tree: with k = b = 0
// every box's entry and exit parameters are 0, every box is "hit", and the links spell out one depth-first order, so each
// record is visited once (`guard` = the record count + 1) and every triangle gets the exact test, like the loop's.
template <typename P>
__device__ __forceinline__ void bvh_walk(const float *__restrict__ nodes, const float *__restrict__ btris, P grid,
                                         float prune, int guard, f3 ro, f3 rd, float &best, int &best_i, bool wild = false) {
    BvhRay r = bvh_ray(ro, rd, ptd::mk(grid[0], grid[1], grid[2]), ptd::mk(grid[3], grid[4], grid[5]));
    if (wild) { r.kx = r.ky = r.kz = 0.0f; r.bx = r.by = r.bz = 0.0f; }
    int node = 0;
    for (int it = 0; it < guard && node >= 0; ++it) node = bvh_step(nodes, btris, prune, r, node, best, best_i);
}

// nearest mesh hit of a path so far (meshes fold in geom order: strict `>` keeps the first on ties)
struct MeshBest { float t; int geom, tri; };

}  // namespace
