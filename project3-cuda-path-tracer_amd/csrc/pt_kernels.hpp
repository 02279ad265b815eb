// pt_kernels.hpp -- the HIP kernels of libptmi355.so (included by ptmi355.hip only), counterparts of
// the reference's src/pathtrace.cu kernels: k_raygen (generateRayFromCamera :122-143), k_intersect
// (computeIntersections :149-213), k_bounce (intersect + shade/scatter + stable compaction, fused),
// k_sort_* (material sort), k_shade_fake (shadeFakeMaterial :224-266), k_gather (finalGather
// :269-278), k_tonemap (sendImageToPBO :48-68) and the AoS import/export helpers.
#pragma once

namespace {

// ---------------------------------------------------------------------------
// generateRayFromCamera -> SoA pool, `count` samples (stepping interface; the
// batch path generates rays inside bounce 0)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_raygen(Pool p, pt_camera cam, Lens lens, TileMap map, int count,
                                                  int iter0, int trace_depth, Control *ctl) {
    uint32_t total = (uint32_t)map.tile_pixels * (uint32_t)count;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) { ctl->nlive[0] = total; }
    if (i >= total) return;
    const uint32_t smp = i / (uint32_t)map.tile_pixels;
    const uint32_t j = i - smp * (uint32_t)map.tile_pixels;
    if (iter0 < 0) iter0 = (int)ctl->iter0;                  // graph replay
    f3 o, d;
    camera_ray(cam, lens, trace_depth, iter0 + (int)smp, local_to_pixel(map, (int)j), map.W, o, d);
    char *q = p.slot(i);
    pf(q, 0) = o.x; pf(q, 1) = o.y; pf(q, 2) = o.z;
    pf(q, 3) = d.x; pf(q, 4) = d.y; pf(q, 5) = d.z;
    pf(q, 6) = 1.0f; pf(q, 7) = 1.0f; pf(q, 8) = 1.0f;
    ppid(q) = i;
}

// ---------------------------------------------------------------------------
// scene staging + intersection (computeIntersections, pathtrace.cu:149-213)
// ---------------------------------------------------------------------------
// Dynamic LDS carve (no static __shared__: the dynamic base stays 16-B aligned, guide G17):
// [ctl: 16 dwords][mats: nmats*12 dwords][tri tile: TRI_TILE*9 dwords (if any)]
// Geom records are NOT staged: every lane of every wave reads the same record, so they are
// fetched with wave-uniform (scalar, SGPR) loads straight from the 1-KB record array, which
// costs no VGPRs and no LDS bandwidth; materials are per-lane gathers and live in LDS.
constexpr int LDS_CTL_WORDS = 16;    // [0] last-block flag, [2..5] scan scratch
constexpr int GF_WORDS = 32;         // staged per geom for per-lane gathers: transform[12], type, materialid, 2 pad,
                                     // invTranspose[12], 4 pad
// per-wave candidate queue: ring of 128 slots, SoA: qo.xyz qd.xyz t_obj (7 planes), meta, and
// the 64 per-lane best keys (u64)
constexpr int Q_SLOTS = 128;
constexpr int Q_WORDS = 7 * Q_SLOTS + Q_SLOTS + 2 * 64;
__host__ __device__ constexpr int scene_lds_words(int nmats, int ngeoms) {
    return ((nmats * ptd::MAT_WORDS + 3) & ~3) + ngeoms * GF_WORDS;
}
__device__ __forceinline__ void stage_scene(float *lds_mats, const SceneDev &sc) {
    const int mw = sc.nmats * ptd::MAT_WORDS;
    for (int k = threadIdx.x; k < mw; k += BLOCK) lds_mats[k] = sc.mats[k];
    {   // per-lane gathers of the tail: forward transform (12) + type + material per geom
        float *gf = lds_mats + ((mw + 3) & ~3);
        for (int k = threadIdx.x; k < sc.ngeoms * GF_WORDS; k += BLOCK) {
            const int g = k / GF_WORDS, w = k - g * GF_WORDS;
            float v = 0.0f;
            if (w < 12) v = sc.geoms[g * ptd::GEOM_WORDS + ptd::G_FWD + w];
            else if (w < 16) v = sc.geoms[g * ptd::GEOM_WORDS + (w - 12)];          // type, materialid, mesh range
            else if (w < 28) v = sc.geoms[g * ptd::GEOM_WORDS + ptd::G_INVT + (w - 16)];
            gf[k] = v;
        }
    }
    __syncthreads();
}

// Geom records are read through the CONSTANT address space: the array is immutable for the
// lifetime of the launch and the address is wave-uniform, so the loads become s_load_dwordxN
// (scalar cache -> SGPRs) instead of per-lane vector loads.
typedef const __attribute__((address_space(4))) float cfloat;
__device__ __forceinline__ cfloat *as_const(const float *p) {
    return (cfloat *)(unsigned long long)p;
}

// One lane-dense pass over up to 64 queued candidates [head, head+count): lane k evaluates the
// shared tail of candidate head+k for whichever lane queued it and folds the distance into that
// lane's best key with an LDS 64-bit min.  key = (bits(t) << 32) | absolute slot: positive floats
// order like their bit patterns and slots are issued in geom order, so the minimum key is the
// smallest t with the lowest geom index on ties -- pathtrace.cu:192's strict `t_min > t` scan.
__device__ __forceinline__ void queue_pass(float *wq, const float *gf, uint32_t head, uint32_t count, f3 ro) {
    const int lane = threadIdx.x & 63;
    float *qf = wq;
    uint32_t *qi = reinterpret_cast<uint32_t *>(wq + 7 * Q_SLOTS);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(wq + 8 * Q_SLOTS);
    const bool on = (uint32_t)lane < count;
    const uint32_t abs_slot = head + (uint32_t)lane;
    const uint32_t s = abs_slot & (Q_SLOTS - 1);
    const uint32_t meta = on ? qi[s] : 0u;
    const int origin = (int)(meta & 63u);
    // the queued lane's world-space ray origin
    const f3 oro = ptd::mk(__shfl(ro.x, origin), __shfl(ro.y, origin), __shfl(ro.z, origin));
    if (on) {
        const f3 qo = ptd::mk(qf[0 * Q_SLOTS + s], qf[1 * Q_SLOTS + s], qf[2 * Q_SLOTS + s]);
        const f3 qd = ptd::mk(qf[3 * Q_SLOTS + s], qf[4 * Q_SLOTS + s], qf[5 * Q_SLOTS + s]);
        const float t_obj = qf[6 * Q_SLOTS + s];
        const float *fwd = gf + (meta >> 10) * GF_WORDS;                  // per-lane gather of the transform
        f3 obj_p;
        const float t = ptd::world_distance(fwd, oro, qo, qd, t_obj, obj_p);
        qf[0 * Q_SLOTS + s] = obj_p.x; qf[1 * Q_SLOTS + s] = obj_p.y; qf[2 * Q_SLOTS + s] = obj_p.z;
        if (t > 0.0f)
            __hip_atomic_fetch_min(&best[origin], ((unsigned long long)__float_as_uint(t) << 32) | abs_slot,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

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
    float kx, ky, kz, bx, by, bz;     // slab form over the mesh's grid: t = grid * k + b
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
    r.bx = (origin.x - ro.x) * ix; r.by = (origin.y - ro.y) * iy; r.bz = (origin.z - ro.z) * iz;
    r.oct = (rd.x < 0.0f ? 1 : 0) | (rd.y < 0.0f ? 2 : 0) | (rd.z < 0.0f ? 4 : 0);
    return r;
}
// entry / exit parameters of the box packed in three dwords (pt_bvh.hpp), clipped to t >= 0
__device__ __forceinline__ void bvh_slab(const BvhRay &r, uint32_t w0, uint32_t w1, uint32_t w2, float &tn, float &tf) {
    const float lx = (float)(w0 & 0xffffu), ly = (float)(w0 >> 16), lz = (float)(w1 & 0xffffu);
    const float hx = (float)(w1 >> 16), hy = (float)(w2 & 0xffffu), hz = (float)(w2 >> 16);
    const float t1x = __builtin_fmaf(lx, r.kx, r.bx), t2x = __builtin_fmaf(hx, r.kx, r.bx);
    const float t1y = __builtin_fmaf(ly, r.ky, r.by), t2y = __builtin_fmaf(hy, r.ky, r.by);
    const float t1z = __builtin_fmaf(lz, r.kz, r.bz), t2z = __builtin_fmaf(hz, r.kz, r.bz);
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)),
                         __builtin_fmaxf(__builtin_fminf(t1z, t2z), 0.0f));
    tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)),
                         __builtin_fmaxf(t1z, t2z));
}
struct BvhRec { uint4 a, b; int miss; };   // the three loads of one record: boxes, boxes + links, miss[octant]
__device__ __forceinline__ BvhRec bvh_fetch(const float *__restrict__ nodes, int node, int oct) {
    const uint4 *n4 = reinterpret_cast<const uint4 *>(nodes + (size_t)node * BVH_NODE_WORDS);
    BvhRec rec;
    rec.a = n4[0]; rec.b = n4[1];
    rec.miss = reinterpret_cast<const int *>(n4)[8 + oct];
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
        if (ptd::ray_triangle(r.ro, r.rd, ptd::mk(P.x, P.y, P.z), ptd::mk(P.w, Q.x, Q.y), ptd::mk(Q.z, Q.w, S.x), tz)) {
            const int orig = __float_as_int(S.y);                        // index in the caller's triangle array
            if (tz > 0.0f && (best > tz || (best == tz && orig < best_i))) { best = tz; best_i = orig; }
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
template <typename P>
__device__ __forceinline__ void bvh_walk(const float *__restrict__ nodes, const float *__restrict__ btris, P grid,
                                         float prune, int guard, f3 ro, f3 rd, float &best, int &best_i) {
    const BvhRay r = bvh_ray(ro, rd, ptd::mk(grid[0], grid[1], grid[2]), ptd::mk(grid[3], grid[4], grid[5]));
    int node = 0;
    for (int it = 0; it < guard && node >= 0; ++it) node = bvh_step(nodes, btris, prune, r, node, best, best_i);
}

template <int MESH>
__device__ __forceinline__ void intersect_scene(const float *__restrict__ geoms, const SceneDev &sc,
                                                float *tri_lds, bool active,
                                                f3 ro, f3 rd, ptd::Hit &h, float *wq = nullptr,
                                                const float *gf = nullptr, const float4 *pre_hit = nullptr) {
    h.t = FLT_MAX; h.geom = -1; h.outside = 1; h.aux = ptd::mk(0, 0, 0);
    const int ngeoms = sc.ngeoms;
    const float *__restrict__ tris = sc.tris;
    const int lane_q = threadIdx.x & 63;
    float *qf = wq;
    uint32_t *qi = reinterpret_cast<uint32_t *>(wq + 7 * Q_SLOTS);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(wq + 8 * Q_SLOTS);
    best[lane_q] = ~0ull;
    unsigned long long seen = ~0ull;
    uint32_t q_head = 0, q_total = 0;                   // wave-uniform
    int w_geom = -1, w_meta = 0;
    f3 w_objp = ptd::mk(0, 0, 0);
    // after a pass: lanes whose best key changed latch the winner's record while it is still intact
    auto latch = [&]() {
        const unsigned long long key = best[lane_q];
        if (key != seen) {
            seen = key;
            const uint32_t s = (uint32_t)key & (Q_SLOTS - 1);
            w_meta = (int)qi[s];
            w_geom = w_meta >> 10;
            w_objp = ptd::mk(qf[0 * Q_SLOTS + s], qf[1 * Q_SLOTS + s], qf[2 * Q_SLOTS + s]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    for (int g = 0; g < ngeoms; ++g) {
        cfloat *rec = as_const(geoms) + g * ptd::GEOM_WORDS;           // wave-uniform address -> s_load
        const int type = __float_as_int(rec[0]);
        if (MESH == MESH_PRE && type == PT_TRIANGLE_MESH) continue;        // k_mesh already walked every mesh
        if (MESH == MESH_BVH && type == PT_TRIANGLE_MESH) {
            // same winner as the loop below (smallest bary.z, lowest triangle index on ties), found by
            // walking the mesh's bounding-volume hierarchy instead of testing every triangle
            const int root = __float_as_int(rec[2]);
            const int count = __float_as_int(rec[3]);
            float best = FLT_MAX;
            int best_i = -1;
            if (active && count > 0)
                bvh_walk(sc.bvh_nodes + (size_t)root * BVH_NODE_WORDS, sc.bvh_tris, rec + ptd::G_INV, sc.bvh_prune,
                         sc.bvh_guard, ro, rd, best, best_i);
            if (active && best_i >= 0) {
                f3 p = ptd::add(ro, ptd::scale(rd, best));
                const float t = ptd::length(ptd::sub(ro, p));
                if (t > 0.0f && h.t > t) {
                    h.t = t; h.geom = g; h.outside = 1;
                    h.aux = ptd::mk(__int_as_float(best_i), 0.0f, 0.0f);
                }
            }
            continue;
        }
        if (MESH == MESH_TILES && type == PT_TRIANGLE_MESH) {
            // completion spec 8.0: nearest triangle by strictly smaller bary.z, first wins ties
            const int first = __float_as_int(rec[2]);
            const int count = __float_as_int(rec[3]);
            float best = FLT_MAX;
            int best_i = -1;
            for (int base = 0; base < count; base += TRI_TILE) {
                const int nt = min(TRI_TILE, count - base);
                const int nt4 = (nt + 3) & ~3;                   // the tile is zero-padded to a multiple of 4
                __syncthreads();
                {   // global -> LDS, 16 B per thread per step; zero triangles (a = 0 < eps: never hit) as padding
                    const float4 *src = reinterpret_cast<const float4 *>(tris + (size_t)(first + base) * TRI_WORDS);
                    float4 *dst = reinterpret_cast<float4 *>(tri_lds);
                    for (int k = threadIdx.x; k < nt4 * 3; k += BLOCK)
                        dst[k] = k < nt * 3 ? src[k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                __syncthreads();
                if (active) {
                    // four triangles per step: their twelve ds_read_b128 (wave-uniform addresses, LDS
                    // broadcasts) are issued together so the LDS latency is paid once per four tests
                    const float4 *tl = reinterpret_cast<const float4 *>(tri_lds);
                    for (int k = 0; k < nt4; k += 4) {
                        float4 w[12];
#pragma unroll
                        for (int j = 0; j < 12; ++j) w[j] = tl[k * 3 + j];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float4 A = w[3 * j], B = w[3 * j + 1], C = w[3 * j + 2];
                            float tz;
                            if (ptd::ray_triangle(ro, rd, ptd::mk(A.x, A.y, A.z), ptd::mk(A.w, B.x, B.y),
                                                  ptd::mk(B.z, B.w, C.x), tz)) {
                                if (tz > 0.0f && best > tz) { best = tz; best_i = first + base + k + j; }
                            }
                        }
                    }
                }
            }
            if (active && best_i >= 0) {
                f3 p = ptd::add(ro, ptd::scale(rd, best));
                const float t = ptd::length(ptd::sub(ro, p));
                if (t > 0.0f && h.t > t) {
                    h.t = t; h.geom = g; h.outside = 1;
                    h.aux = ptd::mk(__int_as_float(best_i), 0.0f, 0.0f);
                }
            }
            continue;
        }
        {   // object-space test per lane; hits are queued and their tails run lane-dense (queue_pass)
            f3 qo = ptd::mk(0, 0, 0), qd = ptd::mk(0, 0, 1);
            float t_obj = 0.0f;
            int code = 7, cand_outside = 1;
            bool hit = false;
            if (active) {
                if (type == PT_CUBE) hit = ptd::box_slab(rec, ro, rd, qo, qd, t_obj, code, cand_outside);
                else if (type == PT_SPHERE) hit = ptd::sphere_quad(rec, ro, rd, qo, qd, t_obj, cand_outside);
            }
            const uint64_t m = __ballot(hit);
            if (m) {
                if (hit) {
                    const uint32_t s = (q_total + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane_q) - 1)))) &
                                       (Q_SLOTS - 1);
                    qf[0 * Q_SLOTS + s] = qo.x; qf[1 * Q_SLOTS + s] = qo.y; qf[2 * Q_SLOTS + s] = qo.z;
                    qf[3 * Q_SLOTS + s] = qd.x; qf[4 * Q_SLOTS + s] = qd.y; qf[5 * Q_SLOTS + s] = qd.z;
                    qf[6 * Q_SLOTS + s] = t_obj;
                    qi[s] = (uint32_t)lane_q | ((uint32_t)cand_outside << 6) | ((uint32_t)code << 7) | ((uint32_t)g << 10);
                }
                q_total += (uint32_t)__popcll((unsigned long long)m);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (q_total - q_head >= 64) {            // a full wave of tails is waiting
                    queue_pass(wq, gf, q_head, 64, ro);
                    q_head += 64;
                    latch();
                }
            }
            continue;
        }
    }
    if (q_total > q_head) {
        queue_pass(wq, gf, q_head, q_total - q_head, ro);
        latch();
    }
    if (MESH == MESH_PRE && pre_hit) {                           // this lane's nearest mesh hit, found by k_mesh
        const float4 m = *pre_hit;
        h.t = m.x; h.geom = __float_as_int(m.y); h.outside = 1; h.aux = ptd::mk(m.z, 0.0f, 0.0f);
    }
    if (w_geom >= 0) {
        const unsigned long long key = seen;
        const float t = __uint_as_float((uint32_t)(key >> 32));
        if (h.t > t || (h.t == t && w_geom < h.geom)) {      // meshes fold straight into h: keep geom order on ties
            h.t = t; h.geom = w_geom; h.outside = (w_meta >> 6) & 1;
            const int type = __float_as_int(gf[w_geom * GF_WORDS + 12]);
            h.aux = (type == PT_CUBE) ? ptd::mk(__int_as_float((w_meta >> 7) & 7), 0.0f, 0.0f) : w_objp;
        }
    }
}

// normal + materialId of the winning primitive: a per-lane gather from the records staged in LDS
// (gf) -- ~100 cycles instead of an L2 round trip -- or from the global record array
__device__ __forceinline__ void resolve_hit(const float *__restrict__ geoms, const float *gf,
                                            const float *__restrict__ tris, const ptd::Hit &h, float &t, f3 &n,
                                            int &mat) {
    if (h.geom < 0) { t = -1.0f; n = ptd::mk(0, 0, 0); mat = 0; return; }
    (void)geoms;
    const float *rec = gf + h.geom * GF_WORDS;
    const int type = __float_as_int(rec[12]);
    mat = __float_as_int(rec[13]);
    const float *fwd = rec, *invt = rec + 16;
    t = h.t;
    if (type == PT_CUBE) n = ptd::cube_normal(fwd, h.aux);
    else if (type == PT_SPHERE) n = ptd::sphere_normal(invt, h.aux, h.outside);
    else {
        const float *tv = tris + (size_t)__float_as_int(h.aux.x) * TRI_WORDS;
        n = ptd::normalize(ptd::cross(ptd::mk(tv[3], tv[4], tv[5]), ptd::mk(tv[6], tv[7], tv[8])));
    }
}

// ---- reading a range-packed pool -------------------------------------------------------
// Wave-cooperative 64-ary search: largest r in [0, W) with base[r] <= P (P < base[W]).
__device__ __forceinline__ uint32_t find_range(const uint32_t *base, uint32_t W, uint32_t P) {
    const int lane = threadIdx.x & 63;
    uint32_t lo = 0, hi = W;                        // answer in [lo, hi)
    for (int guard = 0; guard < 8 && hi - lo > 1; ++guard) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + (uint32_t)lane * step;
        const uint32_t v = idx < hi ? base[idx] : 0xffffffffu;
        const uint64_t ok = __ballot(v <= P);       // base[] is non-decreasing: a prefix of the lanes
        const uint32_t k = (uint32_t)__popcll((unsigned long long)ok);
        const uint32_t nlo = lo + (k ? k - 1 : 0) * step;
        hi = min(hi, nlo + step);
        lo = nlo;
    }
    return lo;
}

// Source slots of the 64 logical paths p = p0 + lane, starting the search at range `cur`
// (wave-uniform, base[cur] <= p0).  Lane l first holds base[cur + l]; a 6-step binary search
// reads other lanes' values with ds_bpermute.  Returns the slot; `cur` advances to the range of
// the tile's last path so the next tile of the run starts where this one ended.
__device__ __forceinline__ uint32_t resolve_src(const RangeDir &dir, uint32_t span, uint32_t &cur, uint32_t p,
                                                bool active, Control *ctl) {
    const int lane = threadIdx.x & 63;
    const uint32_t *base = dir.base();
    bool resolved = !active;
    uint32_t src = 0, rng = cur;
    uint32_t s = cur;
    // bounded: a sane directory resolves within W/63 + 1 windows; every wave reaches the exit
    for (uint32_t guard = 0;; ++guard) {
        if (guard > dir.W / 63 + 1) {
            if (lane == 0) atomicOr(&ctl->error, 2u);
            break;
        }
        const uint32_t t = s + (uint32_t)lane;
        const uint32_t w = t <= dir.W ? base[t] : 0xffffffffu;
        int lo = 0, hi = 63;                        // w(lane 0) <= p always holds for unresolved lanes
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi + 1) >> 1;
            const uint32_t wm = (uint32_t)__shfl((int)w, mid);
            if (wm <= p) lo = mid; else hi = mid - 1;
        }
        const uint32_t wl = (uint32_t)__shfl((int)w, lo);
        if (!resolved && lo < 63) { resolved = true; rng = s + (uint32_t)lo; src = rng * span + (p - wl); }
        if (!__any(!resolved)) break;
        s += 63;
    }
    // the highest active lane holds the tile's last path
    const uint64_t act = __ballot(active);
    if (act) cur = (uint32_t)__builtin_amdgcn_readlane((int)rng, 63 - __builtin_clzll((unsigned long long)act));
    return src;
}

// standalone computeIntersections: materialises the ShadeableIntersection planes
// (indexed by LOGICAL path index)
template <int MESH>
__global__ __launch_bounds__(BLOCK, PT_ISECT_WAVES) void k_intersect(Pool in, Isect out, SceneDev sc,
                                                                    const uint32_t *n_ptr, uint32_t n_fixed,
                                                                    RangeDir dir_in, const uint32_t *nprev_ptr,
                                                                    Control *ctl) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float *mats_lds = lds_raw + LDS_CTL_WORDS;
    const float *gf = mats_lds + ((sc.nmats * ptd::MAT_WORDS + 3) & ~3);
    float *wq = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + (threadIdx.x >> 6) * Q_WORDS;
    float *tri_lds = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + WAVES * Q_WORDS;
    stage_scene(mats_lds, sc);
    const float *gsrc = sc.geoms;
    const int lane = threadIdx.x & 63;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);
    const bool packed = dir_in.mem && nprev_ptr;
    const uint32_t span = packed ? range_tiles(*nprev_ptr, W) * TILE : 0;
    uint32_t cur = 0;
    if (packed && wid * R < tiles) cur = find_range(dir_in.base(), W, wid * R * TILE);
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t tile = wid * R + r;
        if (MESH != MESH_TILES && tile >= tiles) break;
        const bool have = tile < tiles;
        const uint32_t i = tile * TILE + lane;
        bool active = have && i < n;
        uint32_t src = i;
        if (packed && have) src = resolve_src(dir_in, span, cur, i, active, ctl);
        f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1);
        if (active) {
            char *q = in.slot(src);
            if (ppid(q) == DEAD_PID) active = false;
            ro = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
            rd = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
        }
        ptd::Hit h;
        intersect_scene<MESH>(gsrc, sc, tri_lds, active, ro, rd, h, wq, gf);
        if (have && i < n) {
            float t; f3 nrm; int mat;
            resolve_hit(gsrc, gf, sc.tris, h, t, nrm, mat);
            // a miss writes only t; the other fields read as the zeros of pathtrace.cu:343's memset
            out.plane(0)[i] = t; out.plane(1)[i] = nrm.x; out.plane(2)[i] = nrm.y; out.plane(3)[i] = nrm.z;
            out.mat()[i] = mat | (h.outside ? 0 : (int)0x80000000u);
        }
    }
}

// ---------------------------------------------------------------------------
// stable compaction: range counts -> range bases, by the last workgroup out
// ---------------------------------------------------------------------------
// Hand-off (guide G16): each wave stores its range count with an agent-scope atomic
// (write-through) store and drains it (s_waitcnt vmcnt(0)); after the workgroup's barrier one
// lane adds 1 to done[depth]; the workgroup whose add returns grid-1 is last, acquires once
// (agent scope) and scans the W counts (<= 8 steps of 1024).  Nothing spins; nothing depends
// on dispatch order.
__device__ __forceinline__ void scan_range_counts(const RangeDir &dir, uint32_t *n_out,
                                                  uint32_t *lds_scan /* >= 8 words */) {
    // One step: thread t owns the `per` consecutive entries [t*per, (t+1)*per) (per = ceil(W/256) rounded
    // to a multiple of 4, at most 32 for W <= 8192), loads them with 16-B loads all issued up front,
    // and the 256 partial sums cross through one wave scan + one LDS exchange.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t W = dir.W;
    const uint4 *count4 = reinterpret_cast<const uint4 *>(dir.count());
    uint4 *base4 = reinterpret_cast<uint4 *>(dir.base());
    const uint32_t per4 = ((W + BLOCK - 1) / BLOCK + 3) / 4;          // uint4s per thread, <= 8
    const uint32_t first = threadIdx.x * per4 * 4;                    // first entry of this thread
    uint4 v[8];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t e = first + 4 * k;
        v[k] = make_uint4(0, 0, 0, 0);
        if (k < per4 && e < W) {
            v[k] = count4[e >> 2];                                      // count[] is padded to a multiple of 4
            if (e + 1 >= W) v[k].y = 0;
            if (e + 2 >= W) v[k].z = 0;
            if (e + 3 >= W) v[k].w = 0;
        }
        sum += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off);
        if (lane >= off) incl += u;
    }
    if (lane == 63) lds_scan[wave] = incl;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = lds_scan[w];
        if (w < wave) wave_off += c;
        total += c;
    }
    uint32_t run = wave_off + incl - sum;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
        const uint32_t e = first + 4 * k;
        if (k < per4 && e < W) {
            uint4 b;
            b.x = run; b.y = b.x + v[k].x; b.z = b.y + v[k].y; b.w = b.z + v[k].z;
            base4[e >> 2] = b;                                           // base[] has 4 spare entries
            run = b.w + v[k].w;
        }
    }
    if (threadIdx.x == 0) { dir.base()[W] = total; *n_out = total; }
}

// ---------------------------------------------------------------------------
// material sort (INSTRUCTION.md:78-86; spec 8.0): stable counting sort of the live paths and
// their intersections by key = materialId (misses last), before shading
// ---------------------------------------------------------------------------
// Pass 1 (k_sort_hist): every wave histograms the keys of its run of R tiles (wave64
// match-ballot, per-wave bins in LDS) into table[bin][wave]; the last workgroup out scans the
// bin-major table (nbins * W words) in place into start offsets.  Pass 2 (k_sort_scatter):
// every wave walks its run again and moves path state + intersection to
// offset[key][wave] + (same-key paths already seen in the run) + (same-key lanes below it),
// which is the stable order.  The sorted pool is dense.
constexpr int SORT_MAX_BINS = 256;

struct SortArgs {
    Pool in, out;            // out: dense, sorted
    Isect isect, isect_out;  // logical order in, sorted order out
    RangeDir dir_in;
    Control *ctl;
    uint32_t *table;         // nbins * W words
    int depth, nbins;        // nbins = nmats + 1 (misses)
    uint32_t pool_n;
    int compact;
};

__device__ __forceinline__ uint32_t sort_key(const Isect &is, uint32_t i, int nbins) {
    const float t = is.plane(0)[i];
    const int m = is.mat()[i] & 0x7fffffff;
    return t > 0.0f ? (uint32_t)m : (uint32_t)(nbins - 1);
}

// in-place exclusive scan of `total` words by one workgroup (1024 words per step)
__device__ __forceinline__ void scan_words_inplace(uint32_t *w, uint32_t total, uint32_t *lds_scan) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t steps = (total + 4 * BLOCK - 1) / (4 * BLOCK);
    uint32_t carry = 0;
    for (uint32_t step = 0; step < steps; ++step) {
        const uint32_t e = (step * BLOCK + threadIdx.x) * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (e + k < total) ? w[e + k] : 0u;
        const uint32_t sum = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        uint32_t *slot = lds_scan + (step & 1) * WAVES;
        if (lane == 63) slot[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) {
            const uint32_t c = slot[k];
            if (k < wave) wave_off += c;
            tot += c;
        }
        uint32_t run = carry + wave_off + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e + k < total) w[e + k] = run;
            run += v[k];
        }
        carry += tot;
    }
}

__global__ __launch_bounds__(BLOCK) void k_sort_hist(SortArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *bins = sctl + LDS_CTL_WORDS + (threadIdx.x >> 6) * SORT_MAX_BINS;   // per-wave bins
    const int lane = threadIdx.x & 63;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    const uint32_t n = a.compact ? a.ctl->nlive[a.depth] : a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);
    for (int b = lane; b < a.nbins; b += 64) bins[b] = 0;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t tile = wid * R + r;
        if (tile >= tiles) break;
        const uint32_t i = tile * TILE + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? sort_key(a.isect, i, a.nbins) : 0u;
        uint64_t rem = __ballot(valid);
        while (rem) {                                           // one round per distinct key in the tile
            const int l = __ffsll((unsigned long long)rem) - 1;
            const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, l);
            const uint64_t m = __ballot(valid && key == k);
            if (lane == 0) bins[k] += (uint32_t)__popcll((unsigned long long)m);
            rem &= ~m;
        }
    }
    // publish table[bin][wave] (write-through), then elect the last workgroup to scan it
    for (int b = lane; b < a.nbins; b += 64)
        __hip_atomic_store(&a.table[(size_t)b * W + wid], bins[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = elect_last(a.ctl->bucket[a.depth][1], &a.ctl->done_sort[a.depth]);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        sctl[0] = last ? 1u : 0u;
    }
    __syncthreads();
    if (sctl[0]) scan_words_inplace(a.table, (uint32_t)a.nbins * W, sctl + 2);
}

__global__ __launch_bounds__(BLOCK) void k_sort_scatter(SortArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *bins = sctl + LDS_CTL_WORDS + (threadIdx.x >> 6) * SORT_MAX_BINS;   // per-wave running offsets
    const int lane = threadIdx.x & 63;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    const uint32_t n = a.compact ? a.ctl->nlive[a.depth] : a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);
    const bool packed = a.compact && a.dir_in.mem != nullptr;
    const uint32_t span = packed ? range_tiles(a.ctl->nlive[a.depth - 1], W) * TILE : 0;
    uint32_t cur = 0;
    if (packed && wid * R < tiles) cur = find_range(a.dir_in.base(), W, wid * R * TILE);
    for (int b = lane; b < a.nbins; b += 64) bins[b] = a.table[(size_t)b * W + wid];
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t tile = wid * R + r;
        if (tile >= tiles) break;
        const uint32_t i = tile * TILE + lane;
        const bool valid = i < n;
        uint32_t src = i;
        if (packed) src = resolve_src(a.dir_in, span, cur, i, valid, a.ctl);
        const uint32_t key = valid ? sort_key(a.isect, i, a.nbins) : 0u;
        uint32_t dst = 0;
        uint64_t rem = __ballot(valid);
        while (rem) {
            const int l = __ffsll((unsigned long long)rem) - 1;
            const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, l);
            const uint64_t m = __ballot(valid && key == k);
            const uint32_t base = bins[k];                          // same address for the whole wave
            if (valid && key == k) dst = base + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1)));
            if (lane == 0) bins[k] = base + (uint32_t)__popcll((unsigned long long)m);
            rem &= ~m;
        }
        if (valid) {
            char *qs = a.in.slot(src), *qd = a.out.slot(dst);
#pragma unroll
            for (int k = 0; k < 9; ++k) pf(qd, k) = pf(qs, k);
            ppid(qd) = ppid(qs);
#pragma unroll
            for (int k = 0; k < 4; ++k) a.isect_out.plane(k)[dst] = a.isect.plane(k)[i];
            a.isect_out.mat()[dst] = a.isect.mat()[i];
        }
    }
}

// ---------------------------------------------------------------------------
// the fused bounce kernel
// ---------------------------------------------------------------------------
// MODE_FUSED   : intersect inline (ShadeableIntersection never touches HBM)
// MODE_ISECT   : read the materialised planes written by k_intersect (PT_UNFUSED / sort)
// MODE_CACHE0  : bounce 0 with PT_CACHE_FIRST: the per-pixel intersection cache (INSTRUCTION.md:87-89)
enum { MODE_FUSED = 0, MODE_ISECT = 1, MODE_CACHE0 = 2 };

// per-launch constants of a wave for bounce_tile
struct TileCtx {
    float *mats; const float *gf; float *wq; float *tri_lds;   // LDS carve: materials, per-geom gather records, hit-tail ring, triangle tile
    int lane, iter0;
};

// One 64-path tile of one bounce: load (or generate) the paths, intersect, shade / scatter, write the final
// colour of the paths that end here and append the survivors at dst_base + packed (wave64 ballot + popcount
// rank).  `i` = logical path index (what MODE_ISECT planes and the mesh mask are keyed by), `src` = pool slot.
template <int MODE, bool COMPACT, int MESH>
__device__ __forceinline__ void bounce_tile(const BounceArgs &a, const TileCtx &c, const Pool &in, const Pool &out, int depth,
                                            bool gen_rays, uint32_t tile, uint32_t i, uint32_t src, bool have, bool active,
                                            uint32_t n, uint32_t dst_base, uint32_t &packed, uint32_t &traced) {
    const int lane = c.lane;
    uint32_t pid = DEAD_PID;
    f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1), col = ptd::mk(1.0f, 1.0f, 1.0f);
    if (active) {
        if (gen_rays) {
            pid = i;
        } else {
            // all ten fields of the slot in one burst of loads (one memory latency per tile)
            char *q = in.slot(src);
            pid = ppid(q);
            ro = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
            rd = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
            col = ptd::mk(pf(q, 6), pf(q, 7), pf(q, 8));
            if (pid == DEAD_PID) active = false;
        }
    }
    uint32_t smp = 0;
    int pixel = 0;
    if (active) {
        smp = sample_of(a.map, pid);
        pixel = local_to_pixel(a.map, (int)(pid - smp * (uint32_t)a.map.tile_pixels));
        if (gen_rays) camera_ray(a.cam, a.lens, a.trace_depth, c.iter0 + (int)smp, pixel, a.map.W, ro, rd);
    }
    float t = -1.0f; f3 nrm = ptd::mk(0, 0, 0); int mat = 0; int outside = 1;
    if (MODE == MODE_FUSED) {
        ptd::Hit h;
        const float *gsrc = a.scene.geoms;
        const float4 *pre_hit = nullptr;
        if (MESH == MESH_PRE) {
            // lanes of this tile for which k_mesh found a mesh hit; the mask is consumed (cleared) here
            const unsigned long long mm = a.mesh_mask[tile];
            if (mm) {
                if (lane == 0) a.mesh_mask[tile] = 0ull;
                if (active && ((mm >> lane) & 1ull)) pre_hit = a.mesh_hit + src;
            }
        }
        intersect_scene<MESH>(gsrc, a.scene, c.tri_lds, active, ro, rd, h, c.wq, c.gf, pre_hit);
        if (active) { resolve_hit(gsrc, c.gf, a.scene.tris, h, t, nrm, mat); outside = h.outside; }
    } else if (active) {
        // MODE_ISECT: planes in logical order; MODE_CACHE0: one record per pixel of the tile
        const uint32_t q = (MODE == MODE_CACHE0) ? pid - smp * (uint32_t)a.map.tile_pixels : i;
        t = at(a.isect.plane(0), q);
        nrm = ptd::mk(at(a.isect.plane(1), q), at(a.isect.plane(2), q), at(a.isect.plane(3), q));
        const int m = at(a.isect.mat(), q);
        mat = m & 0x7fffffff; outside = (m < 0) ? 0 : 1;
    }
    bool alive = false;
    ptd::PathState ps;
    ps.o = ro; ps.d = rd; ps.c = col;
    if (active) {
        alive = ptd::shade_scatter(ps, t, nrm, mat, outside, c.mats, c.iter0 + (int)smp, pixel, depth,
                                   depth == a.trace_depth - 1);
        if (!alive) {
            at(a.fin, pid) = ps.c.x; at(a.fin + (size_t)in.cap, pid) = ps.c.y;
            at(a.fin + 2 * (size_t)in.cap, pid) = ps.c.z;
        }
    }
    // ---- survivors append to the wave's packed run (wave64 ballot + popcount rank) ----
    const uint64_t bal = __ballot(alive);
    const uint64_t act = __ballot(active);
    traced += (uint32_t)__popcll((unsigned long long)act);
    uint32_t dst = i;
    if (COMPACT) {
        dst = dst_base + packed + (uint32_t)__popcll((unsigned long long)(bal & ((1ull << lane) - 1)));
        packed += (uint32_t)__popcll((unsigned long long)bal);
    }
    if (alive) {
        char *q = out.slot(dst);
        pf(q, 0) = ps.o.x; pf(q, 1) = ps.o.y; pf(q, 2) = ps.o.z;
        pf(q, 3) = ps.d.x; pf(q, 4) = ps.d.y; pf(q, 5) = ps.d.z;
        pf(q, 6) = ps.c.x; pf(q, 7) = ps.c.y; pf(q, 8) = ps.c.z;
        ppid(q) = pid;
    } else if (!COMPACT && have && i < n) {
        out.pid(dst) = DEAD_PID;
    }
}

template <int MODE, bool COMPACT, int MESH>
__global__ __launch_bounds__(BLOCK, (MESH == MESH_PRE && PT_PRE_WAVES > PT_MIN_WAVES) ? PT_PRE_WAVES : PT_MIN_WAVES) void k_bounce(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    float *mats = lds_raw + LDS_CTL_WORDS;
    TileCtx c;
    c.mats = mats;
    c.gf = mats + ((a.scene.nmats * ptd::MAT_WORDS + 3) & ~3);
    c.wq = mats + scene_lds_words(a.scene.nmats, a.scene.ngeoms) + (threadIdx.x >> 6) * Q_WORDS;
    c.tri_lds = mats + scene_lds_words(a.scene.nmats, a.scene.ngeoms) + WAVES * Q_WORDS;
#ifdef PT_STAMPS
#define STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && a.depth == PT_STAMPS) a.ctl->stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
    STAMP(0);
    stage_scene(mats, a.scene);
    STAMP(1);
    const int lane = threadIdx.x & 63;
    c.lane = lane;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    c.iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;       // graph replay: arguments are frozen
    const uint32_t n = (COMPACT && !a.gen_rays) ? a.ctl->nlive[a.depth] : a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);                        // logical tiles per wave (one contiguous run)
    const bool packed_in = COMPACT && a.dir_in.mem != nullptr;
    const uint32_t span_in = packed_in ? range_tiles(a.ctl->nlive[a.depth - 1], W) * TILE : 0;
    uint32_t traced = 0;
    uint32_t packed = 0;                                         // survivors this wave has written (wave-uniform)
    uint32_t cur = 0;                                            // source range of the run's current position
    if (a.gen_rays && blockIdx.x == 0 && threadIdx.x == 0) a.ctl->nlive[0] = a.pool_n;   // k_raygen's job otherwise
    if (packed_in && wid * R < tiles) cur = find_range(a.dir_in.base(), W, wid * R * TILE);
    STAMP(2);

    // every wave walks its own run of R consecutive 64-path tiles; no workgroup barrier inside
    // the loop unless a mesh needs block-wide triangle staging (then all waves run R iterations)
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t tile = wid * R + r;
        if (MESH != MESH_TILES && tile >= tiles) break;
        const bool have = tile < tiles;
        const uint32_t i = tile * TILE + lane;                    // logical path index
        const bool active = have && i < n;
        uint32_t src = i;
        if (packed_in && have) src = resolve_src(a.dir_in, span_in, cur, i, active, a.ctl);
        bounce_tile<MODE, COMPACT, MESH>(a, c, a.in, a.out, a.depth, a.gen_rays != 0, tile, i, src, have, active, n,
                                         wid * R * TILE, packed, traced);
    }
    STAMP(6);
    // paths traced this bounce: with compaction it is simply the live count; otherwise count the alive
    // slots, one atomic per workgroup (summed through LDS) rather than one per wave on a single address
    if (COMPACT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->alive[a.depth] = n;
    } else {
        if (lane == 0) sctl[8 + (threadIdx.x >> 6)] = traced;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tb = sctl[8] + sctl[9] + sctl[10] + sctl[11];
            if (tb) atomicAdd(&a.ctl->alive[a.depth], tb);
        }
    }

    if (COMPACT) {
        // every wave publishes its range count; the last workgroup out scans them
        if (lane == 0)
            __hip_atomic_store(&a.dir_out.count()[wid], packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's count store has left
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool last = elect_last(a.ctl->bucket[a.depth][0], &a.ctl->done[a.depth]);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sctl[0] = last ? 1u : 0u;
        }
        __syncthreads();
        STAMP(7);
        if (sctl[0]) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            scan_range_counts(a.dir_out, &a.ctl->nlive[a.depth + 1], sctl + 2);
            if (threadIdx.x == 0) a.ctl->scan_ticks[a.depth] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t0);
#ifdef PT_STAMPS
            if (threadIdx.x == 0 && a.depth == PT_STAMPS) { a.ctl->stamp[8] = t0; a.ctl->stamp[9] = __builtin_amdgcn_s_memrealtime(); }
#endif
        }
    }
}

// A whole batch in ONE launch, for small batches (the reference's calling pattern is one iteration per
// call): at 1 spp every bounce kernel is ~20 us of fixed cost (launch, scene staging, directory search,
// last-workgroup scan) around a few microseconds of work.  Here every wave generates the camera rays of its
// run of tiles and then keeps ITS OWN survivors through all the bounces: bounce d+1 reads the span the wave
// packed at bounce d (the two pools ping-pong inside the launch), so there is no exchange between waves, no
// directory and no barrier.  The concatenation of the spans is still the stable partition's order; the paths
// just are not dealt out again after every bounce, which costs load balance (a wave whose pixels live long
// works longer) -- the price that makes this the small-batch path only.  Traced counts go to 32 partial sums
// per bounce (Control::bucket[d][1]; a same-address atomic per wave would serialise), folded by k_gather.
__global__ __launch_bounds__(BLOCK, PT_MIN_WAVES) void k_iteration(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float *mats = lds_raw + LDS_CTL_WORDS;
    TileCtx c;
    c.mats = mats;
    c.gf = mats + ((a.scene.nmats * ptd::MAT_WORDS + 3) & ~3);
    c.wq = mats + scene_lds_words(a.scene.nmats, a.scene.ngeoms) + (threadIdx.x >> 6) * Q_WORDS;
    c.tri_lds = nullptr;
    stage_scene(mats, a.scene);
    const int lane = threadIdx.x & 63;
    c.lane = lane;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    c.iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    const uint32_t n = a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);
    const uint32_t base = wid * R * TILE;                         // this wave's span in both pools
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->nlive[0] = n;
    Pool in = a.in, out = a.out;
    uint32_t count = 0;                                           // paths of this wave entering the bounce (d > 0)
    for (int d = 0; d < a.trace_depth; ++d) {
        uint32_t traced = 0, packed = 0;
        if (d == 0) {
            for (uint32_t r = 0; r < R; ++r) {
                const uint32_t tile = wid * R + r;
                if (tile >= tiles) break;
                const uint32_t i = tile * TILE + lane;
                bounce_tile<MODE_FUSED, true, MESH_NONE>(a, c, in, out, 0, true, tile, i, i, true, i < n, n, base, packed, traced);
            }
        } else {
            for (uint32_t t = 0; t * TILE < count; ++t) {
                const uint32_t k = t * TILE + lane;
                bounce_tile<MODE_FUSED, true, MESH_NONE>(a, c, in, out, d, false, 0, base + k, base + k, true, k < count, n, base,
                                                         packed, traced);
            }
        }
        if (lane == 0 && traced)
            atomicAdd(&a.ctl->bucket[d][1][(wid % ELECT_BUCKETS) * 16], traced);
        count = packed;
        if (count == 0) break;
        const Pool tmp = in; in = out; out = tmp;
        // the span this wave just wrote is read back by its own (other) lanes.  Workgroup scope is enough -- the
        // wave stays on its CU, whose vector L1 sees its own write-through stores -- and costs only the wait; an
        // agent-scope fence writes back / invalidates the L2 and made the launch 4x slower.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------
// Mesh pre-pass (PT_MESH_BVH, fused path).  Walking a hierarchy inside k_bounce keeps a whole
// wave waiting on the few lanes whose rays reach a mesh (a mesh covers a few per cent of the
// directions) while each of their steps is a dependent fetch.  k_mesh runs first instead: every
// wave scans tiles dealt round-robin, tests each ray against the root record of every mesh (two
// boxes, wave-uniform scalar loads) and appends the candidates {slot, path, ray} to a per-wave
// LDS ring; lanes without a walk take ring entries, all lanes walk together, and the triangles of
// the leaves they reach are queued and tested 64 at a time.  Results go
// to mesh_hit[slot] = {t, geom, triangle} and one bit per path in mesh_mask[tile]; k_bounce
// <MESH_PRE> folds them with the geom-index tie-break of pathtrace.cu:192.
// ---------------------------------------------------------------------------
#ifndef PT_SKIP_PAIRS
#define PT_SKIP_PAIRS 2                      // missed-sibling pairs remembered per walk (registers)
#endif
#ifndef PT_MESH_WAVES
#define PT_MESH_WAVES 4                      // waves per SIMD k_mesh is register-budgeted for
#endif
constexpr int MQ_SLOTS = 128;                 // ray ring entries per wave (a tile adds <= 64 while < 64 wait)
constexpr int TQ_SLOTS = 512;                 // triangle ring entries per wave (a step adds <= 64 * 2 * LEAF_MAX while < 64 wait)
constexpr int MQ_RAY_WORDS = 8 * MQ_SLOTS;    // src, path, origin xyz, direction xyz
constexpr int MQ_WORDS = MQ_RAY_WORDS + TQ_SLOTS + 2 * 64;   // + triangle ring + the 64 per-lane best keys (u64)
#ifndef PT_MQ_STEPS
#define PT_MQ_STEPS 8
#endif
#ifndef PT_MQ_LEAVE
#define PT_MQ_LEAVE 56
#endif
constexpr int MQ_STEPS = PT_MQ_STEPS;         // walk steps between two looks at the ray ring
constexpr int MQ_LEAVE = PT_MQ_LEAVE;         // lanes still busy when the wave goes back to scanning
constexpr unsigned long long TRI_KEY_NONE = (0x7f7fffffull << 32) | 0xffffffffull;   // bary.z = FLT_MAX, no triangle
static_assert(2 * PT_LEAF_MAX * 64 + 63 <= TQ_SLOTS, "a step's triangles must fit beside the waiting ones");

// per-lane state of a walk in flight; it survives across the scanning of further tiles
struct MeshWalker {
    bool have;
    uint32_t src, path;
    BvhRay ray;
    int mesh, node, steps;            // position in SceneDev::bvh_meshes, record in that mesh's tree
    int geom, root;                   // of the current mesh
    uint32_t ticket;                  // triangle-ring index past this lane's last queued triangle
    int skip[PT_SKIP_PAIRS], to[PT_SKIP_PAIRS];   // newest (missed far sibling -> where its miss link leads) pairs, newest first
    float best_t; int best_geom, best_tri;   // best over the meshes finished so far (world distance, geom order)
};
struct MeshRings { uint32_t q_head, q_total, t_head, t_total; };   // wave-uniform ring cursors

// the ray in the grid of mesh geom `g` (origin / step sit in the inverse-transform words of its record)
__device__ __forceinline__ BvhRay mesh_ray(const SceneDev &sc, int g, f3 ro, f3 rd) {
    const float *q = sc.geoms + (size_t)g * ptd::GEOM_WORDS + ptd::G_INV;
    return bvh_ray(ro, rd, ptd::mk(q[0], q[1], q[2]), ptd::mk(q[3], q[4], q[5]));
}

// One lane-dense pass over up to 64 queued triangle tests [head, head + count): lane k tests triangle slot
// e >> 6 against the ray of lane e & 63 (fetched from that lane's registers) and folds a hit into the owner's
// best key with an LDS 64-bit min.  key = (bits(bary.z) << 32) | original triangle index: the smallest bary.z,
// the lowest index on ties -- the order of the loop over every triangle (completion spec 8.0).
__device__ __forceinline__ void tri_pass(float *mq, uint32_t head, uint32_t count, const MeshWalker &w, const BounceArgs &a) {
    const int lane = threadIdx.x & 63;
    const uint32_t *tq = reinterpret_cast<const uint32_t *>(mq + MQ_RAY_WORDS);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(mq + MQ_RAY_WORDS + TQ_SLOTS);
    const bool on = (uint32_t)lane < count;
    const uint32_t e = on ? tq[(head + (uint32_t)lane) & (TQ_SLOTS - 1)] : 0u;
    const int owner = (int)(e & 63u);
    const f3 ro = ptd::mk(__shfl(w.ray.ro.x, owner), __shfl(w.ray.ro.y, owner), __shfl(w.ray.ro.z, owner));
    const f3 rd = ptd::mk(__shfl(w.ray.rd.x, owner), __shfl(w.ray.rd.y, owner), __shfl(w.ray.rd.z, owner));
    if (on) {
        const float4 *t4 = reinterpret_cast<const float4 *>(a.scene.bvh_tris + (size_t)(e >> 6) * TRI_WORDS);
        const float4 P = t4[0], Q = t4[1], S = t4[2];
        float tz;
        if (ptd::ray_triangle(ro, rd, ptd::mk(P.x, P.y, P.z), ptd::mk(P.w, Q.x, Q.y), ptd::mk(Q.z, Q.w, S.x), tz) && tz > 0.0f)
            __hip_atomic_fetch_min(&keys[owner], ((unsigned long long)__float_as_uint(tz) << 32) | __float_as_uint(S.y),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Lanes without a walk take the next ray-ring entries; every lane with one advances MQ_STEPS records.  The
// triangles of the leaves a step reaches are not tested by the lane that found them -- a handful of lanes
// would each run the 65-instruction test while the rest of the wave waits -- but queued and tested 64 at a
// time (tri_pass).  A lane whose walk of a mesh is over waits until its last queued triangle has been tested,
// then folds the mesh's winner and moves on to the next mesh or publishes its result.  Returns when the ray
// ring is empty and fewer than `leave` lanes are still busy (0: run dry).
__device__ __forceinline__ void mesh_drain(MeshWalker &w, float *mq, MeshRings &rg, const BounceArgs &a, int leave) {
    const int lane = threadIdx.x & 63;
    const uint32_t *mi = reinterpret_cast<const uint32_t *>(mq);
    uint32_t *tq = reinterpret_cast<uint32_t *>(mq + MQ_RAY_WORDS);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(mq + MQ_RAY_WORDS + TQ_SLOTS);
    const int4 *meshes = a.scene.bvh_meshes;
    const uint64_t below = (1ull << lane) - 1;
    for (;;) {
        const uint64_t idle = __ballot(!w.have);
        const uint32_t avail = rg.q_total - rg.q_head;
        if (idle && avail) {
            const uint32_t rank = (uint32_t)__popcll((unsigned long long)(idle & below));
            if (!w.have && rank < avail) {
                const uint32_t s = (rg.q_head + rank) & (MQ_SLOTS - 1);
                const int4 m = meshes[0];                               // {geom, root record, triangles, -}
                w.src = mi[0 * MQ_SLOTS + s]; w.path = mi[1 * MQ_SLOTS + s];
                w.ray = mesh_ray(a.scene, m.x, ptd::mk(mq[2 * MQ_SLOTS + s], mq[3 * MQ_SLOTS + s], mq[4 * MQ_SLOTS + s]),
                                 ptd::mk(mq[5 * MQ_SLOTS + s], mq[6 * MQ_SLOTS + s], mq[7 * MQ_SLOTS + s]));
                w.mesh = 0; w.geom = m.x; w.root = m.y; w.node = 0; w.steps = 0; w.ticket = rg.t_head;
#pragma unroll
                for (int u = 0; u < PT_SKIP_PAIRS; ++u) { w.skip[u] = -1; w.to[u] = -1; }
                w.best_t = FLT_MAX; w.best_geom = -1; w.best_tri = -1;
                keys[lane] = TRI_KEY_NONE;
                w.have = true;
            }
            rg.q_head += min((uint32_t)__popcll((unsigned long long)idle), avail);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const uint64_t busy = __ballot(w.have);
        if (!busy) return;
        if (rg.q_total == rg.q_head && (int)__popcll((unsigned long long)busy) < leave) return;
#pragma unroll 1
        for (int k = 0; k < MQ_STEPS; ++k) {
            int leaf_l = -1, leaf_r = -1;
            if (w.have && w.node >= 0) {
                const BvhRec rec = bvh_fetch(a.scene.bvh_nodes + (size_t)w.root * BVH_NODE_WORDS, w.node, w.ray.oct);
                // prune against the best bary.z the tested triangles have produced so far (it may lag: conservative)
                const float best = __uint_as_float((uint32_t)(keys[lane] >> 32));
                int skip;
                w.node = bvh_decide(rec, w.ray, best + a.scene.bvh_prune, leaf_l, leaf_r, &skip);
                // A missed far sibling would still be entered through the miss links of the subtree walked first,
                // only to fail both of its box tests.  Its own miss link equals this record's, which is known here:
                // remember the pair and jump over the sibling when the walk arrives at it.  PT_SKIP_PAIRS pairs are kept
                // in registers (the deepest ones, where most visits happen); a forgotten pair only costs the visit.
                if (skip >= 0) {
#pragma unroll
                    for (int u = PT_SKIP_PAIRS - 1; u > 0; --u) { w.skip[u] = w.skip[u - 1]; w.to[u] = w.to[u - 1]; }
                    w.skip[0] = skip; w.to[0] = rec.miss;
                } else {
#pragma unroll
                    for (int u = 0; u < PT_SKIP_PAIRS; ++u)
                        if (w.node >= 0 && w.node == w.skip[0]) {
                            w.node = w.to[0];
#pragma unroll
                            for (int v = 0; v + 1 < PT_SKIP_PAIRS; ++v) { w.skip[v] = w.skip[v + 1]; w.to[v] = w.to[v + 1]; }
                            w.skip[PT_SKIP_PAIRS - 1] = -1;
                        }
                }
#ifdef PT_MESH_STATS
                if (leaf_l < 0 && leaf_r < 0 && w.node == rec.miss) atomicAdd(&a.ctl->keep[12], 1u);   // nothing hit
                if (leaf_l >= 0 || leaf_r >= 0) atomicAdd(&a.ctl->keep[13], 1u);                        // a leaf hit
#endif
                if (++w.steps > a.scene.bvh_guard) w.node = -1;        // NaN rays: every record is "hit"
            }
#ifdef PT_MESH_STATS
            {
                const uint64_t bb = __ballot(w.have && (w.node >= 0 || leaf_l >= 0 || leaf_r >= 0));
                if (lane == 0 && bb) { atomicAdd(&a.ctl->keep[1], (uint32_t)__popcll((unsigned long long)bb)); atomicAdd(&a.ctl->keep[2], 1u); }
                atomicMax(&a.ctl->keep[14], (uint32_t)w.steps);
            }
#endif
            // queue this step's triangles: slot order = lane order (any order gives the same minimum)
            const int nl = leaf_l >= 0 ? (leaf_l >> 24) : 0, nr = leaf_r >= 0 ? (leaf_r >> 24) : 0;
            const int nt = nl + nr;
            if (__ballot(nt > 0)) {
                uint32_t pre = 0, tot = 0;
#pragma unroll
                for (int bit = 0; bit < 4; ++bit) {                      // exclusive prefix of nt (< 16) over the lanes
                    const uint64_t bm = __ballot((nt >> bit) & 1);
                    pre += (uint32_t)__popcll((unsigned long long)(bm & below)) << bit;
                    tot += (uint32_t)__popcll((unsigned long long)bm) << bit;
                }
                const uint32_t pos = rg.t_total + pre;
                for (int j = 0; j < nl; ++j)
                    tq[(pos + (uint32_t)j) & (TQ_SLOTS - 1)] = (uint32_t)lane | ((uint32_t)((leaf_l & 0xffffff) + j) << 6);
                for (int j = 0; j < nr; ++j)
                    tq[(pos + (uint32_t)(nl + j)) & (TQ_SLOTS - 1)] = (uint32_t)lane | ((uint32_t)((leaf_r & 0xffffff) + j) << 6);
                if (nt > 0) w.ticket = pos + (uint32_t)nt;
                rg.t_total += tot;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                while (rg.t_total - rg.t_head >= 64) { tri_pass(mq, rg.t_head, 64, w, a); rg.t_head += 64; }
            }
            // nobody is walking any more but triangles are still queued: test them now, their owners are waiting
            if (rg.t_total != rg.t_head && !__ballot(w.have && w.node >= 0)) {
                tri_pass(mq, rg.t_head, rg.t_total - rg.t_head, w, a);
                rg.t_head = rg.t_total;
            }
            if (w.have && w.node < 0 && (int32_t)(rg.t_head - w.ticket) >= 0) {    // this mesh is done and fully tested
                const unsigned long long key = keys[lane];
                if ((uint32_t)key != 0xffffffffu) {                      // completion spec 8.0: distance to origin + dir * bary.z
                    const float tz = __uint_as_float((uint32_t)(key >> 32));
                    const f3 p = ptd::add(w.ray.ro, ptd::scale(w.ray.rd, tz));
                    const float t = ptd::length(ptd::sub(w.ray.ro, p));
                    if (t > 0.0f && w.best_t > t) { w.best_t = t; w.best_geom = w.geom; w.best_tri = (int)(uint32_t)key; }
                }
                if (++w.mesh < a.scene.bvh_nmesh) {
                    const int4 m = meshes[w.mesh];
                    w.geom = m.x; w.root = m.y;
                    w.ray = mesh_ray(a.scene, m.x, w.ray.ro, w.ray.rd);
                    w.node = 0; w.steps = 0;
#pragma unroll
                    for (int u = 0; u < PT_SKIP_PAIRS; ++u) w.skip[u] = -1;
                    keys[lane] = TRI_KEY_NONE;
                } else {
                    if (w.best_geom >= 0) {
                        a.mesh_hit[w.src] = make_float4(w.best_t, __int_as_float(w.best_geom), __int_as_float(w.best_tri), 0.0f);
                        atomicOr(&a.mesh_mask[w.path >> 6], 1ull << (w.path & 63u));
                    }
                    w.have = false;
                }
            }
        }
    }
}

template <bool COMPACT>
__global__ __launch_bounds__(BLOCK, PT_MESH_WAVES) void k_mesh(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float *mq = lds_raw + (threadIdx.x >> 6) * MQ_WORDS;
    uint32_t *mi = reinterpret_cast<uint32_t *>(mq);
    const int lane = threadIdx.x & 63;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    const int iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    // bounce 0 of a batch: nlive[0] is written by that bounce's own kernel, so the count comes from the host
    const uint32_t n = (COMPACT && !a.gen_rays) ? a.ctl->nlive[a.depth] : a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);
    const bool packed_in = COMPACT && a.dir_in.mem != nullptr;
    const uint32_t Wd = a.dir_in.W;                               // waves of the grid that packed the pool
    const uint32_t span_in = packed_in ? range_tiles(a.ctl->nlive[a.depth - 1], Wd) * TILE : 0;
    MeshRings rg{0, 0, 0, 0};
    MeshWalker w;
    w.have = false; w.src = 0; w.path = 0; w.ray = bvh_ray(ptd::mk(0, 0, 0), ptd::mk(0, 0, 1), ptd::mk(0, 0, 0), ptd::mk(1, 1, 1));
    w.mesh = 0; w.node = -1; w.steps = 0; w.ticket = 0; w.best_t = FLT_MAX; w.best_geom = -1; w.best_tri = -1;
    w.geom = 0; w.root = 0;
#pragma unroll
    for (int u = 0; u < PT_SKIP_PAIRS; ++u) { w.skip[u] = -1; w.to[u] = -1; }
    for (uint32_t r = 0; r < R; ++r) {
        // Tiles are dealt round-robin, not in runs: the pool keeps pixel order through every (stable) compaction,
        // so the rays that reach a mesh -- and the ones that leave its surface -- sit in neighbouring tiles; a run
        // of them would keep one wave walking long after the others are done (measured: waves alive 15 % of the
        // launch on average).  The price is one directory search per tile instead of a cursor.
        const uint32_t tile = r * W + wid;
        if (tile >= tiles) break;
        uint32_t cur = 0;
        if (packed_in) cur = find_range(a.dir_in.base(), Wd, tile * TILE);
        const uint32_t i = tile * TILE + lane;
        bool active = i < n;
        uint32_t src = i;
        if (packed_in) src = resolve_src(a.dir_in, span_in, cur, i, active, a.ctl);
        f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1);
        if (active) {
            if (a.gen_rays) {
                const uint32_t smp = sample_of(a.map, i);
                const int pixel = local_to_pixel(a.map, (int)(i - smp * (uint32_t)a.map.tile_pixels));
                camera_ray(a.cam, a.lens, a.trace_depth, iter0 + (int)smp, pixel, a.map.W, ro, rd);
            } else {
                char *q = a.in.slot(src);
                if (ppid(q) == DEAD_PID) active = false;
                ro = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
                rd = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
            }
        }
        // candidate: the ray reaches one of the two root boxes of some mesh (wave-uniform scalar loads)
        bool cand = false;
#pragma unroll 1
        for (int k = 0; k < a.scene.bvh_nmesh; ++k) {
            const __attribute__((address_space(4))) int *mrec =
                (const __attribute__((address_space(4))) int *)(unsigned long long)(a.scene.bvh_meshes + k);
            cfloat *grid = as_const(a.scene.geoms) + (size_t)mrec[0] * ptd::GEOM_WORDS + ptd::G_INV;
            const __attribute__((address_space(4))) uint32_t *b =
                (const __attribute__((address_space(4))) uint32_t *)(unsigned long long)(a.scene.bvh_nodes + (size_t)mrec[1] * BVH_NODE_WORDS);
            const BvhRay br = bvh_ray(ro, rd, ptd::mk(grid[0], grid[1], grid[2]), ptd::mk(grid[3], grid[4], grid[5]));
            float tn, tf;
            bvh_slab(br, b[0], b[1], b[2], tn, tf);
            cand |= tn <= tf;
            bvh_slab(br, b[3], b[4], b[5], tn, tf);
            cand |= tn <= tf;
        }
        cand = cand && active;
        const uint64_t m = __ballot(cand);
        if (m) {
            if (cand) {
                const uint32_t s = (rg.q_total + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1)))) & (MQ_SLOTS - 1);
                mi[0 * MQ_SLOTS + s] = src; mi[1 * MQ_SLOTS + s] = i;
                mq[2 * MQ_SLOTS + s] = ro.x; mq[3 * MQ_SLOTS + s] = ro.y; mq[4 * MQ_SLOTS + s] = ro.z;
                mq[5 * MQ_SLOTS + s] = rd.x; mq[6 * MQ_SLOTS + s] = rd.y; mq[7 * MQ_SLOTS + s] = rd.z;
            }
            rg.q_total += (uint32_t)__popcll((unsigned long long)m);
#ifdef PT_MESH_STATS
            if (lane == 0) atomicAdd(&a.ctl->keep[0], (uint32_t)__popcll((unsigned long long)m));
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // keep the ring below 64 waiting entries so the next tile always fits
            if (rg.q_total - rg.q_head >= 64 - (uint32_t)__popcll((unsigned long long)__ballot(w.have)))
                mesh_drain(w, mq, rg, a, MQ_LEAVE);
        }
    }
    mesh_drain(w, mq, rg, a, 0);
}

// First-bounce cache (INSTRUCTION.md:87-89): camera rays do not depend on the iteration (no
// jitter, pathtrace.cu:134), so computeIntersections of bounce 0 is evaluated once per pixel and
// camera and reused by every sample.
template <int MESH>
__global__ __launch_bounds__(BLOCK, PT_MIN_WAVES) void k_cache_first(Isect cache, SceneDev sc, pt_camera cam,
                                                                      TileMap map) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float *mats_lds = lds_raw + LDS_CTL_WORDS;
    const float *gf = mats_lds + ((sc.nmats * ptd::MAT_WORDS + 3) & ~3);
    float *wq = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + (threadIdx.x >> 6) * Q_WORDS;
    float *tri_lds = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + WAVES * Q_WORDS;
    stage_scene(mats_lds, sc);
    const float *gsrc = sc.geoms;
    const uint32_t n = (uint32_t)map.tile_pixels;
    const uint32_t tiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t j = tile * BLOCK + threadIdx.x;
        const bool active = j < n;
        f3 ro = ptd::mk(cam.position.x, cam.position.y, cam.position.z), rd = ptd::mk(0, 0, 1);
        if (active) camera_ray(cam, Lens{0, 0.0f, 0.0f}, 0, 0, local_to_pixel(map, (int)j), map.W, ro, rd);   // pinhole only (pt_init)
        ptd::Hit h;
        intersect_scene<MESH>(gsrc, sc, tri_lds, active, ro, rd, h, wq, gf);
        if (active) {
            float t; f3 nrm; int mat;
            resolve_hit(gsrc, gf, sc.tris, h, t, nrm, mat);
            cache.plane(0)[j] = t; cache.plane(1)[j] = nrm.x; cache.plane(2)[j] = nrm.y; cache.plane(3)[j] = nrm.z;
            cache.mat()[j] = mat | (h.outside ? 0 : (int)0x80000000u);
        }
    }
}

// shadeFakeMaterial (pathtrace.cu:224-266): one bounce, never spawns a ray
__global__ __launch_bounds__(BLOCK) void k_shade_fake(Pool p, Isect is, const float *mats_g, TileMap map,
                                                      int iter0, uint32_t n, float *fin) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t pid = p.pid(i);
    const uint32_t s = sample_of(map, pid);
    const int idx = local_to_pixel(map, (int)(pid - s * (uint32_t)map.tile_pixels));
    f3 c = ptd::mk(p.f(i, 6), p.f(i, 7), p.f(i, 8));
    const float t = is.plane(0)[i];
    if (t > 0.0f) {
        uint32_t rng = ptd::seeded_engine(iter0 + (int)s, idx, 0);
        const float *m = mats_g + (is.mat()[i] & 0x7fffffff) * ptd::MAT_WORDS;
        f3 mc = ptd::mk(m[0], m[1], m[2]);
        if (m[9] > 0.0f) {
            c = ptd::mul(c, ptd::scale(mc, m[9]));
        } else {
            f3 nrm = ptd::mk(is.plane(1)[i], is.plane(2)[i], is.plane(3)[i]);
            float lightTerm = ptd::dot(nrm, ptd::mk(0.0f, 1.0f, 0.0f));
            f3 x = ptd::scale(ptd::scale(mc, lightTerm), 0.3f);
            f3 y = ptd::scale(ptd::scale(mc, (1.0f - t * 0.02f)), 0.7f);
            c = ptd::mul(c, ptd::add(x, y));
            c = ptd::scale(c, ptd::u01(rng));
        }
    } else {
        c = ptd::mk(0.0f, 0.0f, 0.0f);
    }
    p.f(i, 6) = c.x; p.f(i, 7) = c.y; p.f(i, 8) = c.z;
    fin[pid] = c.x; fin[(size_t)p.cap + pid] = c.y; fin[2 * (size_t)p.cap + pid] = c.z;
}

// finalGather (pathtrace.cu:269-278): image[pixelIndex] += colour, one add per
// pixel per iteration, samples added in iteration order
__global__ __launch_bounds__(BLOCK) void k_gather(float *image, const float *fin, uint32_t cap, TileMap map,
                                                  int count, Control *ctl, Persist *per, int depths,
                                                  uint32_t fake_rays, int partial_counts) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j == 0) {                    // fold this batch's ray count into the persistent counter
        if (partial_counts)          // k_iteration left 32 partial sums per bounce
            for (int d = 0; d < depths; ++d) {
                uint32_t s = 0;
                for (int k = 0; k < ELECT_BUCKETS; ++k) s += ctl->bucket[d][1][k * 16];
                ctl->alive[d] = s;
            }
        unsigned long long r = fake_rays;
        for (int d = 0; d < depths; ++d) r += ctl->alive[d];
        per->rays += r;
        per->iterations += (unsigned long long)count;
        per->first_rays += depths > 0 ? ctl->alive[0] : fake_rays;
    }
    if (j >= (uint32_t)map.tile_pixels) return;
    const int pix = local_to_pixel(map, (int)j);
    float r = image[3 * pix + 0], g = image[3 * pix + 1], b = image[3 * pix + 2];
    for (int s = 0; s < count; ++s) {
        const size_t k = (size_t)s * map.tile_pixels + j;
        r += fin[k]; g += fin[(size_t)cap + k]; b += fin[2 * (size_t)cap + k];
    }
    image[3 * pix + 0] = r; image[3 * pix + 1] = g; image[3 * pix + 2] = b;
}

// sendImageToPBO (pathtrace.cu:48-68)
__global__ __launch_bounds__(BLOCK) void k_tonemap(uint8_t *pbo, const float *image, int npix, int iter) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= npix) return;
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double v = (double)(image[3 * i + k] / (float)iter) * 255.0;
        int q = (int)v;                      // v_cvt_i32_f64: saturating, NaN -> 0
        c[k] = q < 0 ? 0 : (q > 255 ? 255 : q);
    }
    uchar4 o;
    o.x = (unsigned char)c[0]; o.y = (unsigned char)c[1]; o.z = (unsigned char)c[2]; o.w = 0;
    reinterpret_cast<uchar4 *>(pbo)[i] = o;
}

// pool <-> reference AoS (debug / parity export and pt_intersect_once)
__global__ void k_export_paths(Pool p, TileMap map, uint32_t n_total, uint32_t n_live, int remaining,
                               pt_path_segment *out, RangeDir dir, uint32_t span) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    uint32_t src = i;
    if (dir.mem) {                        // logical -> physical: largest r with base[r] <= i
        const uint32_t *base = dir.base();
        uint32_t lo = 0, hi = dir.W - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (base[mid] <= i) lo = mid; else hi = mid - 1;
        }
        src = lo * span + (i - base[lo]);
    }
    pt_path_segment s;
    s.ray.origin = {p.f(src, 0), p.f(src, 1), p.f(src, 2)};
    s.ray.direction = {p.f(src, 3), p.f(src, 4), p.f(src, 5)};
    s.color = {p.f(src, 6), p.f(src, 7), p.f(src, 8)};
    const uint32_t pid = p.pid(src);
    if (pid == DEAD_PID) { s.pixelIndex = -1; s.remainingBounces = 0; }
    else {
        const uint32_t sm = sample_of(map, pid);
        s.pixelIndex = local_to_pixel(map, (int)(pid - sm * (uint32_t)map.tile_pixels));
        s.remainingBounces = i < n_live ? remaining : 0;
    }
    out[i] = s;
}

__global__ void k_import_paths(Pool p, const pt_path_segment *in, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pt_path_segment s = in[i];
    p.f(i, 0) = s.ray.origin.x; p.f(i, 1) = s.ray.origin.y; p.f(i, 2) = s.ray.origin.z;
    p.f(i, 3) = s.ray.direction.x; p.f(i, 4) = s.ray.direction.y; p.f(i, 5) = s.ray.direction.z;
    p.f(i, 6) = s.color.x; p.f(i, 7) = s.color.y; p.f(i, 8) = s.color.z;
    p.pid(i) = i;
}

__global__ void k_export_isects(Isect is, uint32_t n, pt_shadeable_intersection *out, uint8_t *outside) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pt_shadeable_intersection s;
    const int m = is.mat()[i];
    s.t = is.plane(0)[i];
    if (s.t > 0.0f) { s.surfaceNormal = {is.plane(1)[i], is.plane(2)[i], is.plane(3)[i]}; s.materialId = m & 0x7fffffff; }
    else { s.surfaceNormal = {0, 0, 0}; s.materialId = 0; }      // memset(0) + t = -1 only
    out[i] = s;
    if (outside) outside[i] = (m < 0) ? 0 : 1;
}


}  // namespace
