// pt_kernels.hpp -- the HIP kernels of libptmi355.so (included by ptmi355.hip only), counterparts of
// the reference's src/pathtrace.cu kernels: k_raygen (generateRayFromCamera :122-143), k_intersect
// (computeIntersections :149-213), k_bounce (intersect + shade/scatter + stable compaction, fused),
// k_iteration (all bounces of a small batch in one launch), k_sort_hist / k_shade_sorted_w / k_shade_sorted (material
// sort), k_mesh (triangle-mesh pre-pass), k_cull0_mask (bounce-0 candidate masks), k_shade_fake (shadeFakeMaterial
// :224-266), k_gather (finalGather :269-278), k_tonemap (sendImageToPBO :48-68) and the AoS import/export helpers.
#pragma once

namespace {

// wave64 ballot straight from the compare (HIP's __ballot() goes through select 0/1 + compare-not-equal)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// number of set bits of the wave mask m below this lane: v_mbcnt_lo + v_mbcnt_hi on the scalar mask (the generic
// popcount(m & ((1 << lane) - 1)) compiles to two ands and two bit counts on per-lane copies of the mask)
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// the lanes of a wave mask as a per-lane predicate, for free (the mask becomes the exec mask of the branch)
__device__ __forceinline__ bool lane_of(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// The instruction arbiter serves the OLDEST wave of a SIMD first.  In a persistent grid whose waves all have the same
// amount of work that is the worst order: measured on k_bounce (per-wave start / end times, 5 waves per SIMD), the wave
// in slot 0 -- the first-dispatched fifth of the workgroups -- ended at 0.55-0.7 of the launch, the one in slot 4 at
// 0.92, and every SIMD spent the last third of every launch with fewer and fewer waves to pick instructions from (mean
// residency 0.71-0.81 of the launch).  Rotating the user priority (s_setprio, which the arbiter ranks above age) with
// the wave's tile counter gives every wave the same share of every level: mean residency 0.84-0.94, C2 +10 %.  (No
// effect in k_mesh -- one 16-wave workgroup per CU, whose waves wait on dependent fetches -- and -2 % in the sorted
// shade kernel, eight short-lived workgroups per CU that wait on memory: not used there.)
// `step`: the wave's loop counter (tiles); `slots`: workgroups per CU of the launch (slot = dispatch order).
#ifndef PT_NO_ROTATE_PRIO
__device__ __forceinline__ void set_priority(uint32_t level) {
    switch (level & 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#else
__device__ __forceinline__ void set_priority(uint32_t) {}
#endif
__device__ __forceinline__ void rotate_priority(uint32_t step, uint32_t slots) {
    set_priority((blockIdx.x * slots) / gridDim.x + step);
}

// Final colour of the path that ends here: one 16-B store into final[pid] = {r, g, b, stamp of this batch} -- and only
// when the colour is not zero.  As three planes (round 1) every ending path dirtied three 32-B sectors to deliver
// 12 B; and four paths in five end with colour 0 (they leave the open box or run out of bounces), which adds nothing to
// the sum (x + 0 = x exactly; the sums are never -0): k_gather takes an entry whose stamp is not this batch's as 0.
// Measured on the sorted C3 pipeline: 58 B of HBM writes per ending path before, the 16-B store and its sector.
__device__ __forceinline__ void put_final(float *fin, uint32_t pid, f3 c, uint32_t stamp) {
    if (!(c.x == 0.0f && c.y == 0.0f && c.z == 0.0f))                       // NaN compares false: written
        reinterpret_cast<float4 *>(fin)[pid] = make_float4(c.x, c.y, c.z, __uint_as_float(stamp));
}
// the stamp of the current batch: a launch argument, or (graph replay: arguments are frozen) Control::keep[0]
__device__ __forceinline__ uint32_t batch_stamp(uint32_t arg, const Control *ctl) { return arg ? arg : ctl->keep[0]; }

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
//   [ctl: 16 dwords]
//   [scene block, SLDS only: materials nmats*12 | ginfo ngeoms (padded to 4) | gather records ngeoms*36]
//   [per wave: PW_WORDS]  [per wave: TRQ_WORDS, the triangle candidate ring + best keys (MESH_TILES only)]
// What every lane of a wave reads alike (the cull boxes, mesh records) comes through wave-uniform scalar
// loads from global memory; what lanes gather individually (the matrices of the primitive a candidate names,
// the material of a winner) is staged in LDS when the scene fits (SLDS) and read from global memory through
// the vector cache when it does not (any number of primitives / materials).
constexpr int LDS_CTL_WORDS = 16;    // [0] last-block flag, [2..5] scan scratch, [8..11] traced counts
constexpr int GREC_WORDS = 36;       // gather record per geom: inverseTransform[12] transform[12] invTranspose[12], each
                                     // 4 columns x 3 rows.  Stride 36 words: records of 16 consecutive geoms start in
                                     // distinct 4-bank slots, so a ds_read_b128 by 64 lanes naming different geoms does
                                     // not conflict (the r01 layout, stride 32, put every geom on the same banks: 24 %
                                     // of the LDS cycles were bank conflicts)
// per-wave block: candidate ring + two tiles in flight (rays, best keys, winner records)
constexpr int Q_SLOTS = 128;         // candidate ring entries (a tile's cull adds <= 64 per geom while < 64 wait)
constexpr int PW_RING = 0;                               // u32[128]: lane | parity << 6 | type << 7 | geom << 9
constexpr int PW_BEST = PW_RING + Q_SLOTS;               // u64[2][64]: (bits(t) << 32) | geom << 1 | outside, ~0 = nothing hit
constexpr int PW_WIN = PW_BEST + 2 * 64 * 2;             // float[2][3][64]: winner's normal x, y, z (the outside flag rides in the key)
constexpr int PW_RAYS = PW_WIN + 2 * 64 * 3;             // float[2][6][64]: ro.xyz rd.xyz of the tile's paths
constexpr int PW_WORDS = PW_RAYS + 2 * 6 * 64;           // 1536 dwords = 6 KiB per wave: six workgroups fit a CU's 160 KiB beside a
                                                         // Cornell-sized scene block (round 2: 6.5 KiB, float4 winner records, five)
constexpr int CULL_WORDS = 12;       // per geom, scalar-loaded: lo.x hi.x lo.y hi.y lo.z hi.z | type + (reject mode << 8) | the reject row:
                                     //   m_k0 m_k1 m_k2 m_k3 | spare (48 B: one s_load_dwordx8 + one s_load_dwordx4)

__host__ __device__ constexpr int scene_lds_words(int nmats, int ngeoms) {
    return ((nmats * ptd::MAT_WORDS + 3) & ~3) + ((ngeoms + 3) & ~3) + ngeoms * GREC_WORDS;
}

// where the per-lane gathers of a kernel read from
struct SceneAcc {
    const float *mats;        // MAT_WORDS per material
    const uint32_t *ginfo;    // per geom: materialid | type << 28
    const float *grec;        // GREC_WORDS per geom, 16-B aligned
};

template <bool SLDS>
__device__ __forceinline__ SceneAcc stage_scene(float *lds_scene, const SceneDev &sc) {
    SceneAcc acc;
    if (!SLDS) {
        acc.mats = sc.mats; acc.ginfo = sc.ginfo; acc.grec = sc.grec;
        return acc;
    }
    const int mw = sc.nmats * ptd::MAT_WORDS;
    float *mats = lds_scene;
    uint32_t *ginfo = reinterpret_cast<uint32_t *>(lds_scene + ((mw + 3) & ~3));
    float *grec = lds_scene + ((mw + 3) & ~3) + ((sc.ngeoms + 3) & ~3);
    for (int k = threadIdx.x; k < mw; k += BLOCK) mats[k] = sc.mats[k];
    for (int k = threadIdx.x; k < sc.ngeoms; k += BLOCK) ginfo[k] = sc.ginfo[k];
    {
        const float4 *src = reinterpret_cast<const float4 *>(sc.grec);
        float4 *dst = reinterpret_cast<float4 *>(grec);
        for (int k = threadIdx.x; k < sc.ngeoms * (GREC_WORDS / 4); k += BLOCK) dst[k] = src[k];
    }
    __syncthreads();
    acc.mats = mats; acc.ginfo = ginfo; acc.grec = grec;
    return acc;
}

// Geom-uniform records are read through the CONSTANT address space: the arrays are immutable for the
// lifetime of the launch and the address is wave-uniform, so the loads become s_load_dwordxN
// (scalar cache -> SGPRs) instead of per-lane vector loads.
typedef const __attribute__((address_space(4))) float cfloat;
__device__ __forceinline__ cfloat *as_const(const float *p) {
    return (cfloat *)(unsigned long long)p;
}

// ---------------------------------------------------------------------------
// Intersection of one wave's paths with the scene, in three lane-dense stages.
//
// The reference tests every ray against every primitive in object space (pathtrace.cu:176-199): per cube two
// mat4 * vec4, a normalise, six IEEE divides, then for a hit the shared tail (getPointOnRay, transform back,
// length) and the normal -- ~1550 instructions per ray on Cornell although only ~1.25 primitives per ray are hit.
//
//  1. CULL.  Per primitive a world-space box, computed at pt_init, that contains every ray the reference's own
//     float arithmetic could report a hit for (ptmi355.hip: make_cull_boxes, with the error bound).  All lanes
//     test their ray against it with one v_rcp per axis per RAY and six fused multiply-adds + min/max per
//     primitive (the box comes from wave-uniform scalar loads).  This test only decides which exact tests run,
//     never their outcome, so it may be approximate as long as it errs towards "candidate": rays outside the
//     range the bound was derived for (huge or non-finite origins, odd direction magnitudes) are candidates of
//     everything (`wild`); for the others every slab parameter is finite (cull_ray).
//  2. CANDIDATE RING.  Lanes whose ray reaches the box append (lane, primitive) to a per-wave LDS ring (slot =
//     running total + ballot rank).
//  3. PASS.  Whenever 64 candidates wait, lane k takes candidate k: it fetches that path's ray from the wave's
//     LDS copy, gathers the primitive's matrices (LDS or vector cache), runs the reference's object-space test
//     operation for operation and, on a hit, the tail and the surface normal, and folds the world distance into
//     the owning path's best key with an LDS 64-bit min on (bits(t) << 32) | geom -- positive floats order like
//     their bit patterns, so the minimum is the smallest t with the lowest geom index on ties, exactly
//     pathtrace.cu:192's strict `t_min > t` scan.  The lane whose key is the path's minimum after the pass
//     writes the winner record (normal, outside flag).
//
// A tile's last candidates rarely fill a pass, so two tiles are in flight per wave (parity 0/1 of the per-wave
// LDS block): the leftovers of tile T are tested together with the first candidates of tile T+1, and T is
// shaded after T+1's cull.  Passes therefore run full: ~1.3 per 64 paths on Cornell instead of 2.
// ---------------------------------------------------------------------------
struct CullRay {                      // per path, for stage 1
    float ix, iy, iz, nx, ny, nz;     // slab form: t = plane * i + n   (i = 1/d, n = -o/d)
    bool wild;                        // outside the range the cull bound was derived for: candidate of everything
};
__device__ __forceinline__ CullRay cull_ray(f3 ro, f3 rd, float rmax) {
    CullRay c;
    const float os = (__builtin_fabsf(ro.x) + __builtin_fabsf(ro.y)) + __builtin_fabsf(ro.z);
    const float ds = (__builtin_fabsf(rd.x) + __builtin_fabsf(rd.y)) + __builtin_fabsf(rd.z);
    c.wild = !(os <= rmax) || !(ds >= 9.5367431640625e-07f && ds <= 1048576.0f);      // NaN / inf fail the compares
    // 1/d clamped to +-2^100: a direction component of (nearly) zero would make the planes +-inf and, in the fused
    // form plane * i + n, inf - inf = NaN for every plane on the origin's side of zero -- v_min(NaN, +inf) = +inf
    // would then reject a ray that runs INSIDE the slab.  With the clamp every t of a non-wild ray is finite, an
    // axis-parallel ray inside a slab sees (-huge, +huge), outside it (+-huge, +-huge): the slab test of a ray that
    // is parallel for all purposes (it would need t > 2^46 to cross a pad, far beyond the other axes' exits).
    c.ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(rd.x), -0x1p100f, 0x1p100f);
    c.iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(rd.y), -0x1p100f, 0x1p100f);
    c.iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(rd.z), -0x1p100f, 0x1p100f);
    c.nx = -ro.x * c.ix; c.ny = -ro.y * c.iy; c.nz = -ro.z * c.iz;
    return c;
}
// true unless the ray certainly misses the box [lo, hi] (scalar operands).  NaN-safe towards "true".
__device__ __forceinline__ bool cull_box(const CullRay &c, float lox, float hix, float loy, float hiy, float loz, float hiz) {
    const float t1x = __builtin_fmaf(lox, c.ix, c.nx), t2x = __builtin_fmaf(hix, c.ix, c.nx);
    const float t1y = __builtin_fmaf(loy, c.iy, c.ny), t2y = __builtin_fmaf(hiy, c.iy, c.ny);
    const float t1z = __builtin_fmaf(loz, c.iz, c.nz), t2z = __builtin_fmaf(hiz, c.iz, c.nz);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)),
                                     __builtin_fmaxf(__builtin_fminf(t1z, t2z), 0.0f));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)),
                                     __builtin_fmaxf(t1z, t2z));
    return !(tn > tf);
}

// Is the ray a candidate of the primitive whose cull record is cb[0..10] (wave-uniform)?  The padded world box,
// then an EXACT early miss along one axis of a cube (pt_cull.hpp: reject_row): q_k and v_k = one row of the
// inverseTransform applied to origin and direction in the reference's own operation order; "origin beyond the
// slab and heading away" ( |q_k| > 0.5 and q_k v_k > 0 ) makes both slab parameters of the axis negative: tmax < 0, a
// miss (intersections.h:56-77), whatever the other axes say.  This is what removes a path's OWN surface from
// its candidates: its origin sits 1e-6 above the wall it just left, well inside any box the float error
// allows, and would otherwise cost every bounce ray one object-space test (C2: 0.24 candidates per ray).
// One function for k_bounce / k_intersect and for k_cull0_mask, which memoises "some lane" per camera tile.
// Returns the WAVE MASK of the candidate lanes: every compare goes straight to a scalar register pair and the
// combination -- box and not(early miss) or wild -- is scalar mask arithmetic, not per-lane selects.
__device__ __forceinline__ uint64_t cull_candidates(const CullRay &cr, uint64_t m_wild, f3 ro, f3 rd, float lox, float hix,
                                                    float loy, float hiy, float loz, float hiz, int tw, float m0, float m1,
                                                    float m2, float m3) {
    uint64_t keep = ballot64(cull_box(cr, lox, hix, loy, hiy, loz, hiz));
    const int rmode = (tw >> 8) & 7;                                     // wave-uniform; 0..2 diagonal row, 4 general row, 3 none
    if (rmode != 3) {
        float qk, vk;
        if (rmode == 4) {
            qk = (m0 * ro.x + m1 * ro.y) + (m2 * ro.z + m3);
            vk = (m0 * rd.x + m1 * rd.y) + m2 * rd.z;                    // the reference adds m_k3 * 0.0f = +-0: same value when it matters
        } else if (rmode == 0) {                                          // (scalar branches: selecting the component with
            qk = m0 * ro.x + m3; vk = m0 * rd.x;                          //  wave-uniform v_cndmasks costs nine instructions)
        } else if (rmode == 1) {
            qk = m1 * ro.y + m3; vk = m1 * rd.y;                          // the other products are exact zeros
        } else {
            qk = m2 * ro.z + m3; vk = m2 * rd.z;
        }
        keep &= ~(ballot64(__builtin_fabsf(qk) > 0.5f) & ballot64(qk * vk > 0.0f));
    }
    return keep | m_wild;
}

struct WaveQ {                        // wave-uniform ring cursors + the wave's LDS block
    float *pw;
    uint32_t head, total;
    __device__ __forceinline__ uint32_t *ring() const { return reinterpret_cast<uint32_t *>(pw + PW_RING); }
    __device__ __forceinline__ unsigned long long *best(int par) const {
        return reinterpret_cast<unsigned long long *>(pw + PW_BEST) + par * 64;
    }
    __device__ __forceinline__ float *win(int par) const { return pw + PW_WIN + par * 3 * 64; }
    __device__ __forceinline__ float *rays(int par) const { return pw + PW_RAYS + par * 6 * 64; }
};

#ifdef PT_CULL_STATS
__device__ unsigned long long g_cull_stats[8];     // tiles, candidates, passes, pass lanes, hits, wild paths, active paths
#define CULL_STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_cull_stats[k], (unsigned long long)(v)); } while (0)
#else
#define CULL_STAT(k, v) do {} while (0)
#endif

// stage 3: candidates [head, head + count), count <= 64
__device__ __forceinline__ void cand_pass1(const WaveQ &q, const SceneAcc &acc, uint32_t head, uint32_t count);
__device__ __forceinline__ void cand_pass(const WaveQ &q, const SceneAcc &acc, uint32_t head, uint32_t count) {
    cand_pass1(q, acc, head, count);
#ifdef PT_DBG_PASS2          // cost measurement: every pass twice (idempotent: same keys, same records)
    cand_pass1(q, acc, head, count);
#endif
}
__device__ __forceinline__ void cand_pass1(const WaveQ &q, const SceneAcc &acc, uint32_t head, uint32_t count) {
    const int lane = threadIdx.x & 63;
    CULL_STAT(2, 1); CULL_STAT(3, count);
    if ((uint32_t)lane < count) {
        const uint32_t e = q.ring()[(head + (uint32_t)lane) & (Q_SLOTS - 1)];
        const int origin = (int)(e & 63u), par = (int)((e >> 6) & 1u), type = (int)((e >> 7) & 3u);
        const uint32_t g = e >> 9;
        const float *ry = q.rays(par) + origin;
        const f3 ro = ptd::mk(ry[0], ry[64], ry[128]);
        const f3 rd = ptd::mk(ry[192], ry[256], ry[320]);
        const float4 *rec4 = reinterpret_cast<const float4 *>(acc.grec + (size_t)g * GREC_WORDS);
        float m[12];
        {
            const float4 a = rec4[0], b = rec4[1], c = rec4[2];
            m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
            m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
        }
        f3 qo, v;
        float x;
        ptd::object_ray(m, ro, rd, qo, v, x);
        const f3 qd = ptd::normalize_with(v, x, ptd::norm_fast_ok(x));
        float t_obj = 0.0f;
        int code = 7, outside = 1;
        bool hit = false;
        if (type == PT_CUBE) {
            hit = ptd::cube_slabs(qo, qd, ptd::cube_fast_ok(qo, v, x), t_obj, code, outside);
        } else {
            hit = ptd::sphere_roots(qo, qd, t_obj, outside);
        }
        CULL_STAT(4, __popcll((unsigned long long)ballot64(hit)));
        if (hit) {
            // shared tail of both tests (intersections.h:85-87,136-143)
            {
                const float4 a = rec4[3], b = rec4[4], c = rec4[5];
                m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
                m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
            }
            f3 obj_p;
            const float t = ptd::world_distance(m, ro, qo, qd, t_obj, obj_p);
            // normal: cube = normalize(transform * (face, 0)), sphere = +-normalize(invTranspose * (objP, 0))
            f3 nv = obj_p;
            if (type == PT_CUBE) {
                nv = ptd::face_from_code(code);
            } else {
                const float4 a = rec4[6], b = rec4[7], c = rec4[8];
                m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
                m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
            }
            f3 n = ptd::normalize(ptd::mv_dir(m, nv));
            if (type != PT_CUBE && !outside) n = ptd::neg(n);
            if (t > 0.0f) {                                                // pathtrace.cu:192
                // positive floats order like their bit patterns; geom << 1 | outside: the lowest geom wins a tie in t
                // (the flag belongs to the geom: it cannot reorder two different geoms)
                const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | (g << 1) | (uint32_t)(outside & 1);
                unsigned long long *bk = q.best(par) + origin;
                __hip_atomic_fetch_min(bk, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // LDS operations of one wave execute in order: every min of this pass precedes this read
                if (*bk == key) { float *w = q.win(par) + origin; w[0] = n.x; w[64] = n.y; w[128] = n.z; }
            }
        }
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
template <typename P>
__device__ __forceinline__ void bvh_walk(const float *__restrict__ nodes, const float *__restrict__ btris, P grid,
                                         float prune, int guard, f3 ro, f3 rd, float &best, int &best_i) {
    const BvhRay r = bvh_ray(ro, rd, ptd::mk(grid[0], grid[1], grid[2]), ptd::mk(grid[3], grid[4], grid[5]));
    int node = 0;
    for (int it = 0; it < guard && node >= 0; ++it) node = bvh_step(nodes, btris, prune, r, node, best, best_i);
}

// nearest mesh hit of a path so far (meshes fold in geom order: strict `>` keeps the first on ties)
struct MeshBest { float t; int geom, tri; };

// ---------------------------------------------------------------------------
// MESH_TILES: the loop over EVERY triangle of a mesh for every ray (completion spec 8.0 "Triangles"; BASELINE's
// "naive triangle loop (no BVH)"; INSTRUCTION.md:123-128).  No hierarchy, no grouping: each (ray, triangle) pair is
// visited.  Like the cubes and spheres (stages 1-3 above) a pair is visited in two stages:
//
//  1. BOUND.  pt_init computes per triangle a sphere (centre c, radius Rs) that contains every point a hit the spec
//     accepts can report: the spec's hit-point test (tri_point_ok) only counts a triangle whose reported point
//     P = fl(o + fl(d * tz)) lies inside the triangle's box widened by the mesh's pad, P lies within
//     sqrt3 * 2^-23 (|o| + |P|) of the ray's line, and the test below misplaces that line by less than
//     2^-20 (R + |c|) (R = the |origin|_1 bound of the non-wild rays: ptmi355.hip, tri_bounds, with the error budget)
//     -- so a ray whose line passes the centre at more than Rs cannot be accepted for this triangle, whatever
//     glm::intersectRayTriangle's float arithmetic returns for it.  All lanes test their ray against it:
//     q = c x d' - o x d' (d' = d scaled to unit length, o x d' hoisted per ray), |q|^2 > Rs^2 -> skip: six fused
//     multiply-adds, a three-term dot and one compare per (ray, triangle), the triangle's four floats coming from the
//     wave's own LDS strip by ONE wave-uniform ds_read_b128.  Round 2 ran the exact test on every pair with the
//     triangle read from LDS by three wave-uniform ds_read_b128: bound by the LDS pipe at 41 cycles per (wave, triangle).  Rays the bound was not derived for (non-finite, huge, odd direction magnitudes: `wild`) are
//     candidates of every triangle; NaNs fail the compare towards "candidate".
//  2. EXACT.  Candidates (lane, triangle) queue in a per-wave LDS ring; whenever 64 wait, lane k runs
//     glm::intersectRayTriangle (operation for operation, ptd::ray_triangle) + the hit-point test for candidate k
//     -- the ray from the wave's LDS copy, the triangle record gathered from global memory -- and folds
//     (bits(bary.z) << 32) | triangle index into the owner's key with an LDS 64-bit min: the smallest bary.z, the
//     lowest index on ties, i.e. the loop's strict `best > tz` scan in index order.
// ---------------------------------------------------------------------------
constexpr unsigned long long TRI_KEY_NONE = (0x7f7fffffull << 32) | 0xffffffffull;   // bary.z = FLT_MAX, no triangle
#ifndef PT_SWEEP_AHEAD
#define PT_SWEEP_AHEAD 4                       // spheres read from LDS ahead of the tests that use them
#endif
constexpr int TRQ_SLOTS = 128;                 // triangle candidates waiting per wave (a triangle adds <= 64 while < 64 wait)
constexpr int TRQ_WORDS = TRQ_SLOTS + 2 * 64 + 2 * 64 * 4;   // ring + the 64 per-lane best keys (u64) + two groups of 64 spheres: 3 KiB per wave

__device__ __forceinline__ void tri_cand_pass(const float *ry0, const uint32_t *ring, unsigned long long *keys,
                                              const float *__restrict__ tris, uint32_t head, uint32_t count) {
    const int lane = threadIdx.x & 63;
    if ((uint32_t)lane < count) {
        const uint32_t e = ring[(head + (uint32_t)lane) & (TRQ_SLOTS - 1)];
        const int owner = (int)(e & 63u);
        const uint32_t idx = e >> 6;
        const float *ry = ry0 + owner;
        const f3 ro = ptd::mk(ry[0], ry[64], ry[128]);
        const f3 rd = ptd::mk(ry[192], ry[256], ry[320]);
        const float4 *t4 = reinterpret_cast<const float4 *>(tris + (size_t)idx * TRI_WORDS);
        const float4 A = t4[0], B = t4[1], C = t4[2];
        const f3 v0 = ptd::mk(A.x, A.y, A.z), e1 = ptd::mk(A.w, B.x, B.y), e2 = ptd::mk(B.z, B.w, C.x);
        float tz;
        if (ptd::ray_triangle(ro, rd, v0, e1, e2, tz) && tz > 0.0f && ptd::tri_point_ok(ro, rd, tz, v0, e1, e2, C.z))
            __hip_atomic_fetch_min(&keys[owner], ((unsigned long long)__float_as_uint(tz) << 32) | idx, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// nearest accepted triangle of the mesh [first, first + count) for this lane's ray: best = bary.z, best_i = index
__device__ __forceinline__ void mesh_sweep(const SceneDev &sc, const WaveQ &q, int par, float *trq, int first, int count, int boff,
                                           f3 ro, f3 rd, uint64_t m_act, uint64_t m_wild, float &best, int &best_i) {
    const int lane = threadIdx.x & 63;
    uint32_t *ring = reinterpret_cast<uint32_t *>(trq);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(trq + TRQ_SLOTS);
    const float *ry0 = q.rays(par);
    keys[lane] = TRI_KEY_NONE;
    uint32_t head = 0, total = 0;
    // the ray's line in Pluecker form, direction scaled to unit length (v_rsq: the scale only has to be about right)
    const float sc1 = __builtin_amdgcn_rsqf((rd.x * rd.x + rd.y * rd.y) + rd.z * rd.z);
    const float dx = rd.x * sc1, dy = rd.y * sc1, dz = rd.z * sc1;
    const float mx = __builtin_fmaf(ro.y, dz, -(ro.z * dy)), my = __builtin_fmaf(ro.z, dx, -(ro.x * dz)),
                mz = __builtin_fmaf(ro.x, dy, -(ro.y * dx));
    const float4 *__restrict__ tb = reinterpret_cast<const float4 *>(sc.tri_bound) + (size_t)boff;
    float4 *stage = reinterpret_cast<float4 *>(trq + TRQ_SLOTS + 2 * 64);       // [2][64] spheres, this wave's own
    const uint64_t m_all = m_act & m_wild;                        // candidates of everything
    auto one = [&](float4 t, int k) {
        const float qx = __builtin_fmaf(t.y, dz, __builtin_fmaf(-t.z, dy, -mx));
        const float qy = __builtin_fmaf(t.z, dx, __builtin_fmaf(-t.x, dz, -my));
        const float qz = __builtin_fmaf(t.x, dy, __builtin_fmaf(-t.y, dx, -mz));
        const float qq = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
        const uint64_t m = (m_act & ~ballot64(qq > t.w)) | m_all;  // NaN: not greater, a candidate
        if (__builtin_expect(m != 0, 0)) {                         // rare: ~1e-5 of the pairs
            if (k >= count) return;                                // (a padding sphere and a wild ray)
            if (lane_of(m)) ring[(total + rank_below(m)) & (TRQ_SLOTS - 1)] = (uint32_t)lane | ((uint32_t)(first + k) << 6);
            total += (uint32_t)__popcll((unsigned long long)m);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (total - head >= 64) { tri_cand_pass(ry0, ring, keys, sc.tris, head, 64); head += 64; }
        }
    };
    // 64 spheres per group: one coalesced 16-B load per lane (the next group's is in flight while this one is tested),
    // parked in the wave's own LDS strip and read back as wave-uniform ds_read_b128 -- one LDS read per (wave,
    // triangle), four in flight ahead of the tests that use them.  (Wave-uniform scalar loads straight from memory
    // were measured first: s_load returns out of order, so only one batch can be in flight, and 81 cycles per pair
    // went by waiting on the scalar cache; the array is padded to a multiple of 64 with spheres nothing reaches.)
    const int ngroups = (count + 63) >> 6;
    float4 g_next = ngroups > 0 ? tb[lane] : make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    for (int g = 0; g < ngroups; ++g) {
        float4 *buf = stage + (g & 1) * 64;
        buf[lane] = g_next;
        if (g + 1 < ngroups) g_next = tb[(size_t)(g + 1) * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float4 cur[PT_SWEEP_AHEAD];
#pragma unroll
        for (int u = 0; u < PT_SWEEP_AHEAD; ++u) cur[u] = buf[u];
#pragma unroll 2
        for (int j = 0; j < 64; j += PT_SWEEP_AHEAD) {
            const int jn = (j + PT_SWEEP_AHEAD) & 63;             // the last step re-reads the first entries: harmless
            float4 nxt[PT_SWEEP_AHEAD];
#pragma unroll
            for (int u = 0; u < PT_SWEEP_AHEAD; ++u) nxt[u] = buf[jn + u];
#pragma unroll
            for (int u = 0; u < PT_SWEEP_AHEAD; ++u) one(cur[u], g * 64 + j + u);
#pragma unroll
            for (int u = 0; u < PT_SWEEP_AHEAD; ++u) cur[u] = nxt[u];
        }
    }
    while (total != head) {
        const uint32_t cnt = min(64u, total - head);
        tri_cand_pass(ry0, ring, keys, sc.tris, head, cnt);
        head += cnt;
    }
    const unsigned long long key = keys[lane];
    if ((uint32_t)key != 0xffffffffu) { best = __uint_as_float((uint32_t)(key >> 32)); best_i = (int)(uint32_t)key; }
}

// Stages 1 + 2 for one tile (parity `par` of the wave's LDS block): store the rays, reset the best keys, test every
// primitive's cull box and queue the candidates; passes run as the ring fills.  Triangle meshes keep their own
// paths (every triangle through LDS tiles / the hierarchy inline / the k_mesh pre-pass) and fold into `mb`.
template <int MESH>
__device__ __forceinline__ void cull_scene(const SceneDev &sc, const SceneAcc &acc, WaveQ &q, int par, float *tri_lds,
                                           bool active, f3 ro, f3 rd, MeshBest &mb, const float4 *pre_hit,
                                           bool masked = false, unsigned long long gmask = 0) {
    const int lane = threadIdx.x & 63;
    {
        float *ry = q.rays(par) + lane;
        ry[0] = ro.x; ry[64] = ro.y; ry[128] = ro.z; ry[192] = rd.x; ry[256] = rd.y; ry[320] = rd.z;
        q.best(par)[lane] = ~0ull;
    }
    mb.t = FLT_MAX; mb.geom = -1; mb.tri = -1;
    if (MESH == MESH_PRE && pre_hit) {                           // this lane's nearest mesh hit, found by k_mesh
        const float4 m = *pre_hit;
        mb.t = m.x; mb.geom = __float_as_int(m.y); mb.tri = __float_as_int(m.z);
    }
    const CullRay cr = cull_ray(ro, rd, sc.rmax);
    const uint64_t m_act = ballot64(active), m_wild = ballot64(cr.wild);
    CULL_STAT(0, 1); CULL_STAT(5, __popcll((unsigned long long)ballot64(active && cr.wild))); CULL_STAT(6, __popcll((unsigned long long)ballot64(active)));
    const uint32_t tag = (uint32_t)lane | ((uint32_t)par << 6);
    const int ngeoms = sc.ngeoms;
    // the records come through wave-uniform scalar loads (s_load_dwordx8 + x4).  Round 2 requested the next primitive's
    // record before using this one's (its latency then overlaps the test); by round 3 the eleven scalar registers that
    // keeps alive across the loop cost more than the latency -- the kernel spilled 47 scalar values into VGPR lanes and
    // reloaded 28 of them per tile; without the prefetch it spills 34, and every configuration gained 2-6 %
    // (profiles/r03/variants_cull_prefetch.log).  -DPT_CULL_PREFETCH brings it back.
#ifdef PT_CULL_PREFETCH
    float nxt[11];
    {
        cfloat *c0 = as_const(sc.cull);
#pragma unroll
        for (int k = 0; k < 11; ++k) nxt[k] = ngeoms > 0 ? c0[k] : 0.0f;
    }
#endif
    for (int g = 0; g < ngeoms; ++g) {
        float cb[11];
#ifndef PT_CULL_PREFETCH
        {
            cfloat *cc = as_const(sc.cull) + g * CULL_WORDS;
#pragma unroll
            for (int k = 0; k < 11; ++k) cb[k] = cc[k];
        }
#else
#pragma unroll
        for (int k = 0; k < 11; ++k) cb[k] = nxt[k];
        if (g + 1 < ngeoms) {
            cfloat *cn = as_const(sc.cull) + (g + 1) * CULL_WORDS;
#pragma unroll
            for (int k = 0; k < 11; ++k) nxt[k] = cn[k];
        }
#endif
        // bounce 0: primitives no camera ray of this tile is a candidate of (k_cull0_mask, bit g of the tile's word)
        if (masked && !((gmask >> (g & 63)) & 1ull)) continue;
        const int tw = __float_as_int(cb[6]);
        const int type = tw & 0xff;
        if (MESH != MESH_NONE && type == PT_TRIANGLE_MESH) {
            if (MESH == MESH_PRE) continue;                             // k_mesh already walked every mesh
            cfloat *rec = as_const(sc.geoms) + g * ptd::GEOM_WORDS;
            float best = FLT_MAX;
            int best_i = -1;
            if (MESH == MESH_BVH) {
                // same winner as the loop below (smallest bary.z, lowest triangle index on ties), found by
                // walking the mesh's bounding-volume hierarchy instead of testing every triangle
                const int root = __float_as_int(rec[2]);
                const int count = __float_as_int(rec[3]);
                if (active && count > 0)
                    bvh_walk(sc.bvh_nodes + (size_t)root * BVH_NODE_WORDS, sc.bvh_tris, rec + ptd::G_INV, sc.bvh_prune,
                             sc.bvh_guard, ro, rd, best, best_i);
            } else {
                // every triangle of the mesh, for every ray (the completion spec's loop, 8.0): mesh_sweep
                const int first = __float_as_int(rec[2]);
                const int count = __float_as_int(rec[3]);
                const int boff = __float_as_int(rec[ptd::G_INV + 6]);
                mesh_sweep(sc, q, par, tri_lds, first, count, boff, ro, rd, m_act, m_wild, best, best_i);
            }
            if (active && best_i >= 0) {
                f3 p = ptd::add(ro, ptd::scale(rd, best));
                const float t = ptd::length(ptd::sub(ro, p));
                if (t > 0.0f && mb.t > t) { mb.t = t; mb.geom = g; mb.tri = best_i; }
            }
            continue;
        }
        const uint64_t m = m_act & cull_candidates(cr, m_wild, ro, rd, cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], tw, cb[7], cb[8], cb[9], cb[10]);
        if (m) {
            if (lane_of(m)) {
                const uint32_t s = (q.total + rank_below(m)) & (Q_SLOTS - 1);
                q.ring()[s] = tag | ((uint32_t)type << 7) | ((uint32_t)g << 9);
            }
            q.total += (uint32_t)__popcll((unsigned long long)m);
            CULL_STAT(1, __popcll((unsigned long long)m));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (q.total - q.head >= 64) {                // a full wave of candidates is waiting
                cand_pass(q, acc, q.head, 64);
                q.head += 64;
            }
        }
    }
}

// every candidate queued before `ticket` has been tested when this returns
__device__ __forceinline__ void drain_to(WaveQ &q, const SceneAcc &acc, uint32_t ticket) {
    while ((int32_t)(ticket - q.head) > 0) {
        const uint32_t cnt = min(64u, q.total - q.head);
        cand_pass(q, acc, q.head, cnt);
        q.head += cnt;
    }
}

// the winner of lane's path of the tile with parity `par`: t (-1: miss), normal, materialId, outside flag
__device__ __forceinline__ void tile_result(const WaveQ &q, int par, const SceneAcc &acc, const float *__restrict__ tris,
                                            const MeshBest &mb, float &t, f3 &n, int &mat, int &outside) {
    const int lane = threadIdx.x & 63;
    const unsigned long long key = q.best(par)[lane];
    t = -1.0f; n = ptd::mk(0, 0, 0); mat = 0; outside = 1;
    int geom = -1;
    if (key != ~0ull) {
        const float *w = q.win(par) + lane;
        t = __uint_as_float((uint32_t)(key >> 32)); geom = (int)((uint32_t)key >> 1); outside = (int)((uint32_t)key & 1u);
        n = ptd::mk(w[0], w[64], w[128]);
    }
    if (mb.geom >= 0 && (geom < 0 || t > mb.t || (t == mb.t && mb.geom < geom))) {     // pathtrace.cu:192 across all geoms
        const float *tv = tris + (size_t)mb.tri * TRI_WORDS;
        t = mb.t; geom = mb.geom; outside = 1;
        n = ptd::normalize(ptd::cross(ptd::mk(tv[3], tv[4], tv[5]), ptd::mk(tv[6], tv[7], tv[8])));
    }
    if (geom >= 0) mat = (int)(acc.ginfo[geom] & 0x0fffffffu);
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
        const uint64_t ok = ballot64(v <= P);       // base[] is non-decreasing: a prefix of the lanes
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
#ifndef PT_NO_RESOLVE_FAST
    {
        // A range holds the survivors of a whole run of tiles (thousands of paths), so a tile almost always lies
        // inside the range the previous tile ended in: two wave-uniform loads and one subtraction then replace the
        // windowed search below.
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p) - (uint32_t)__builtin_amdgcn_readfirstlane(lane);
        const uint32_t b0 = base[cur], b1 = base[cur + 1];              // cur < W always (base[] has W + 1 entries)
        if (p0 >= b0 && p0 + 63u < b1) return cur * span + (p - b0);
    }
#endif
    bool resolved = !active;
    uint32_t src = 0, rng = cur;
    uint32_t s = cur;
    // bounded: every window resolves at least the first unresolved lane; every wave reaches the exit
    for (uint32_t guard = 0;; ++guard) {
        if (guard > 66) {
            if (lane == 0) atomicOr(&ctl->error, 2u);
            break;
        }
        const uint32_t t = s + (uint32_t)lane;
        const uint32_t w = t <= dir.nr ? base[t] : 0xffffffffu;
        int lo = 0, hi = 63;                        // w(lane 0) <= p always holds for unresolved lanes
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi + 1) >> 1;
            const uint32_t wm = (uint32_t)__shfl((int)w, mid);
            if (wm <= p) lo = mid; else hi = mid - 1;
        }
        const uint32_t wl = (uint32_t)__shfl((int)w, lo);
        if (!resolved && lo < 63) { resolved = true; rng = s + (uint32_t)lo; src = rng * span + (p - wl); }
        const uint64_t un = ballot64(!resolved);
        if (!un) break;
        // The next window starts at the range that holds the first unresolved path (paths ascend with the lane), found
        // by the 64-ary search -- not 63 ranges further on: between two keys of a sorted pool lie the W ranges of every
        // material nobody survived on (the light: thousands of empty ranges), and a tile that straddles them walked
        // them window by window -- 80 windows of one dependent load each, 60-150 us at the end of every sorted launch.
        const uint32_t pmin = (uint32_t)__builtin_amdgcn_readlane((int)p, __ffsll((unsigned long long)un) - 1);
        const uint32_t nxt = find_range(base, dir.nr, pmin);
        s = nxt > s ? nxt : s + 63;                 // (always ahead: the first unresolved lane lies past this window)
    }
    // the highest active lane holds the tile's last path
    const uint64_t act = ballot64(active);
    if (act) cur = (uint32_t)__builtin_amdgcn_readlane((int)rng, 63 - __builtin_clzll((unsigned long long)act));
    return src;
}

// the LDS carve of a kernel that intersects
struct LdsCarve { float *scene, *pw, *tri; };
__device__ __forceinline__ LdsCarve carve_lds(float *lds_raw, const SceneDev &sc, bool slds) {
    LdsCarve c;
    c.scene = lds_raw + LDS_CTL_WORDS;
    float *after = c.scene + (slds ? scene_lds_words(sc.nmats, sc.ngeoms) : 0);
    c.pw = after + (threadIdx.x >> 6) * PW_WORDS;
    c.tri = after + WAVES * PW_WORDS + (threadIdx.x >> 6) * TRQ_WORDS;        // this wave's triangle queue (MESH_TILES)
    return c;
}

// standalone computeIntersections: materialises the ShadeableIntersection planes
// (indexed by LOGICAL path index).  Two tiles in flight per wave, as in k_bounce.
template <int MESH, bool SLDS, bool GEN = false>
__global__ __launch_bounds__(BLOCK, PT_ISECT_WAVES) void k_intersect(Pool in, Isect out, SceneDev sc,
                                                                    const uint32_t *n_ptr, uint32_t n_fixed,
                                                                    RangeDir dir_in, const uint32_t *nprev_ptr,
                                                                    Control *ctl, const unsigned long long *cull0,
                                                                    uint32_t cull0_tiles, RayGen gen) {
    // cull0 != nullptr: the pool is k_raygen's output for a pinhole camera (bounce 0 of the unfused / sorted
    // pipelines): tile t holds the pixels of camera tile t mod cull0_tiles (k_cull0_mask).
    // GEN (bounce 0 of a sorted batch): path i's camera ray is generated here, in registers -- k_raygen does
    // not run and `in` is not read; k_shade_sorted_w generates the same ray again when it shades the path.
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const LdsCarve lc = carve_lds(lds_raw, sc, SLDS);
    const SceneAcc acc = stage_scene<SLDS>(lc.scene, sc);
    WaveQ q{lc.pw, 0, 0};
    const int lane = threadIdx.x & 63;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);
    const bool packed = dir_in.mem && nprev_ptr;
    const uint32_t span = packed ? range_tiles(*nprev_ptr, dir_in.W) * TILE : 0;
    uint32_t cur = 0;
    if (packed && wid * R < tiles) cur = find_range(dir_in.base(), dir_in.nr, wid * R * TILE);
    auto finish = [&](uint32_t i, int par, const MeshBest &mb) {
        if (i < n) {
            float t; f3 nrm; int mat, outside;
            tile_result(q, par, acc, sc.tris, mb, t, nrm, mat, outside);
            // a miss writes only t; the other fields read as the zeros of pathtrace.cu:343's memset
            out.plane(0)[i] = t; out.plane(1)[i] = nrm.x; out.plane(2)[i] = nrm.y; out.plane(3)[i] = nrm.z;
            out.mat()[i] = mat | (outside ? 0 : (int)0x80000000u);
        }
    };
    bool pending = false;
    uint32_t prev_i = 0, prev_ticket = 0;
    MeshBest prev_mb{FLT_MAX, -1, -1};
    int par = 0;
    const bool masked = cull0 != nullptr;
    uint32_t mtile = masked ? (wid * R) % cull0_tiles : 0u;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t tile = wid * R + r;
        if (tile >= tiles) break;
        rotate_priority(r, PT_ISECT_WAVES + 1);
        const bool have = tile < tiles;
        const uint32_t i = tile * TILE + lane;
        bool active = have && i < n;
        uint32_t src = i;
        if (packed && have) src = resolve_src(dir_in, span, cur, i, active, ctl);
        unsigned long long gmask = 0;
        if (masked) {
            gmask = ((const __attribute__((address_space(4))) unsigned long long *)(unsigned long long)cull0)[mtile];
            if (++mtile == cull0_tiles) mtile = 0;
        }
        f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1);
        if (GEN && active) {
            const uint32_t smp = sample_of(gen.map, i);
            const int pixel = local_to_pixel(gen.map, (int)(i - smp * (uint32_t)gen.map.tile_pixels));
            const int it0 = gen.iter0 >= 0 ? gen.iter0 : (int)ctl->iter0;
            camera_ray(gen.cam, gen.lens, gen.trace_depth, it0 + (int)smp, pixel, gen.map.W, ro, rd);
        } else if (!GEN && active) {
            char *p = in.slot(src);
            if (ppid(p) == DEAD_PID) active = false;
            ro = ptd::mk(pf(p, 0), pf(p, 1), pf(p, 2));
            rd = ptd::mk(pf(p, 3), pf(p, 4), pf(p, 5));
        }
        MeshBest mb;
        cull_scene<MESH>(sc, acc, q, par, lc.tri, active, ro, rd, mb, nullptr, masked, gmask);
        const uint32_t ticket = q.total;
        if (pending) { drain_to(q, acc, prev_ticket); finish(prev_i, par ^ 1, prev_mb); }
        prev_i = have ? i : 0xffffffffu; prev_mb = mb; prev_ticket = ticket; pending = true; par ^= 1;
    }
    if (pending) { drain_to(q, acc, prev_ticket); finish(prev_i, par ^ 1, prev_mb); }
}

// Bounce 0 of a pinhole camera without jitter traces the same rays every iteration, tile by tile: the primitives
// that at least one ray of a 64-pixel camera tile is a candidate of are found once per camera (one wave per tile,
// the very arithmetic of cull_scene) and written down as one bit per primitive; bounce 0 then skips the cull test of
// the others for the whole wave (C2: five or six of the seven).  Only the conservative candidate decision is
// memoised -- every exact test, every hit and every random number is computed per ray and per iteration as before.
// Scenes of up to 64 primitives; meshes are always "candidates" (they have their own paths).
__global__ __launch_bounds__(BLOCK) void k_cull0_mask(SceneDev sc, pt_camera cam, TileMap map, int trace_depth,
                                                     unsigned long long *mask, uint32_t ntiles) {
    const int lane = threadIdx.x & 63;
    const uint32_t tile = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const Lens pinhole{0, 0.0f, 0.0f};
    f3 ro, rd;
    camera_ray(cam, pinhole, trace_depth, 0, local_to_pixel(map, (int)(tile * TILE + lane)), map.W, ro, rd);
    const CullRay cr = cull_ray(ro, rd, sc.rmax);
    const uint64_t m_wild = ballot64(cr.wild);
    unsigned long long bits = 0;
    for (int g = 0; g < sc.ngeoms; ++g) {
        float cb[11];
        cfloat *cn = as_const(sc.cull) + g * CULL_WORDS;
#pragma unroll
        for (int k = 0; k < 11; ++k) cb[k] = cn[k];
        const bool mesh = (__float_as_int(cb[6]) & 0xff) == PT_TRIANGLE_MESH;
        if (mesh || cull_candidates(cr, m_wild, ro, rd, cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], __float_as_int(cb[6]), cb[7],
                                    cb[8], cb[9], cb[10]) != 0)
            bits |= 1ull << (g & 63);
    }
    if (lane == 0) mask[tile] = bits;
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
    // One step covers 8192 entries: thread t owns the `per` consecutive entries [t*per, (t+1)*per) of the step (per =
    // a multiple of 4, at most 32), loads them with 16-B loads all issued up front, and the 256 partial sums cross
    // through one wave scan + one LDS exchange.  W <= 8192 waves: one step; K * W ranges (survivors placed by
    // material): K steps at most, the running total carried from step to step.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t NR = dir.nr;
    const uint4 *count4 = reinterpret_cast<const uint4 *>(dir.count());
    uint4 *base4 = reinterpret_cast<uint4 *>(dir.base());
    uint32_t carry = 0;
    for (uint32_t s0 = 0, step = 0; s0 < NR; s0 += 8192u, ++step) {
        const uint32_t W = min(8192u, NR - s0);                           // entries of this step
        const uint32_t per4 = ((W + BLOCK - 1) / BLOCK + 3) / 4;          // uint4s per thread, <= 8
        const uint32_t first = threadIdx.x * per4 * 4;                    // first entry of this thread within the step
        uint4 v[8];
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t e = first + 4 * k;
            v[k] = make_uint4(0, 0, 0, 0);
            if (k < per4 && e < W) {
                v[k] = count4[(s0 + e) >> 2];                                // count[] is padded to a multiple of 4
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
        uint32_t *slot = lds_scan + (step & 1u) * WAVES;
        if (lane == 63) slot[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = slot[w];
            if (w < wave) wave_off += c;
            total += c;
        }
        uint32_t run = carry + wave_off + incl - sum;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t e = first + 4 * k;
            if (k < per4 && e < W) {
                uint4 b;
                b.x = run; b.y = b.x + v[k].x; b.z = b.y + v[k].y; b.w = b.z + v[k].z;
                base4[(s0 + e) >> 2] = b;                                    // base[] has 4 spare entries; steps start at multiples of 8192
                run = b.w + v[k].w;
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) { dir.base()[NR] = carry; *n_out = carry; }
}

// ---------------------------------------------------------------------------
// material sort (INSTRUCTION.md:78-86; spec 8.0): stable sort of the live paths and their
// intersections by key = materialId (misses last) before shading; the pool order after the
// bounce is the stable partition of that sorted order
// ---------------------------------------------------------------------------
// k_intersect materialises the intersections of the (dense) pool.  Then
//   k_sort_hist   : every WORKGROUP histograms the keys of its contiguous run of 512-path chunks into
//                   table[key][workgroup] -- only the keys whose paths go on (with compaction a key either survives
//                   as a whole or not at all: miss, emissive material and the last bounce end a path, nothing
//                   else does); the last workgroup out scans the table in place: table[key][g] becomes the position
//                   in the OUTPUT pool of workgroup g's first path with that key.
//   k_shade_sorted: every workgroup walks its run again, chunk by chunk: a stable counting sort of the chunk's 512
//                   keys in LDS (per-wave counts -> starts, no data moves), then each wave takes 128 consecutive
//                   SORTED positions, gathers their state and intersection from the chunk's 30 KB of pool rows (every
//                   line the gathers touch is consumed by the same workgroup), shades them -- lanes of a wave run the
//                   same material's code except where two keys meet -- and writes the survivors straight to their
//                   place in the globally sorted, compacted output pool (runs of consecutive slots per key).
// The sort therefore costs one extra read of the keys (8 B per path); nothing is moved to be sorted.  r01 moved
// state + intersection (60 B per path) with fifteen scattered 4-B stores, then read them back: 2.5 TB/s, 43 % of
// the time of a C3 step.
constexpr int SORT_MAX_BINS = 2048;          // one bin per material + misses; per-wave chunk counts live in LDS (32 KiB at the limit)
constexpr int SORT_TPW = 2;                  // 64-path tiles per wave and chunk (1 and 4 measured: -2 % / -5 %)
constexpr int SORT_CHUNK_TILES = SORT_TPW * WAVES;
constexpr int SORT_CHUNK = SORT_CHUNK_TILES * TILE;

__device__ __forceinline__ uint32_t sort_key(const Isect &is, uint32_t i, int nbins) {
    const float t = is.plane(0)[i];
    const int m = is.mat()[i] & 0x7fffffff;
    return t > 0.0f ? (uint32_t)m : (uint32_t)(nbins - 1);
}
// does a path whose intersection has this key go on to the next bounce?  (ptd::shade_scatter's three exits)
__device__ __forceinline__ bool key_survives(const float *__restrict__ mats, uint32_t key, int nbins, bool last_bounce) {
    if (last_bounce || key >= (uint32_t)(nbins - 1)) return false;
    return !(mats[key * ptd::MAT_WORDS + 9] > 0.0f);
}

// in-place exclusive scan of `total` words by one workgroup (1024 words per step); returns the sum
__device__ __forceinline__ uint32_t scan_words_inplace(uint32_t *w, uint32_t total, uint32_t *lds_scan) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (total <= BLOCK * 128u) {
        // small tables (the usual case: 8 keys x 2048 workgroups = 16 K words): every thread owns one contiguous
        // segment, sums it with all its 16-B loads in flight, the 256 sums cross through one wave scan + one LDS
        // exchange, and the segment is read again (L2) and written as prefixes -- one barrier instead of one per
        // 1024 words with a carried dependency (16 steps of ~1.5 us: half of k_sort_hist's 50 us)
        const uint32_t per4 = ((total + BLOCK - 1) / BLOCK + 3) / 4;       // uint4s per thread, <= 32
        const uint32_t first = threadIdx.x * per4 * 4;
        const uint4 *w4 = reinterpret_cast<const uint4 *>(w);
        uint32_t sum = 0;
        for (uint32_t k = 0; k < per4; ++k) {
            const uint32_t e = first + 4 * k;
            if (e < total) {
                const uint4 v = w4[e >> 2];                                  // the table is padded to a multiple of 4 words
                sum += v.x + (e + 1 < total ? v.y : 0u) + (e + 2 < total ? v.z : 0u) + (e + 3 < total ? v.w : 0u);
            }
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == 63) lds_scan[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) {
            const uint32_t c = lds_scan[k];
            if (k < wave) wave_off += c;
            tot += c;
        }
        uint32_t run = wave_off + incl - sum;
        for (uint32_t k = 0; k < per4; ++k) {
            const uint32_t e = first + 4 * k;
            if (e < total) {
                const uint4 v = w4[e >> 2];
                uint4 o;
                o.x = run; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
                run = o.w + v.w;
                if (e + 3 < total) reinterpret_cast<uint4 *>(w)[e >> 2] = o;
                else { w[e] = o.x; if (e + 1 < total) w[e + 1] = o.y; if (e + 2 < total) w[e + 2] = o.z; }
            }
        }
        return tot;
    }
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
    return carry;
}

// chunks of the pool a workgroup owns in the sort kernels: [first, first + count)
__device__ __forceinline__ void sort_run(uint32_t n, uint32_t &first, uint32_t &count) {
    const uint32_t chunks = (n + SORT_CHUNK - 1) / SORT_CHUNK;
    const uint32_t per = (chunks + gridDim.x - 1) / gridDim.x;
    first = min(chunks, blockIdx.x * per);
    count = min(chunks - first, per);
}

// one round per distinct key among the valid lanes: f(key, ballot of the lanes holding it)
template <typename F>
__device__ __forceinline__ void for_each_key(bool valid, uint32_t key, F f) {
    uint64_t rem = ballot64(valid);
    while (rem) {
        const int l = __ffsll((unsigned long long)rem) - 1;
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, l);
        const uint64_t m = ballot64(valid && key == k);
        f(k, m);
        rem &= ~m;
    }
}

template <bool COMPACT>
__global__ __launch_bounds__(BLOCK) void k_sort_hist(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *bins = sctl + LDS_CTL_WORDS;                            // the workgroup's bins
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool compact = COMPACT;
    const uint32_t n = (compact && a.depth > 0) ? a.ctl->nlive[a.depth] : a.pool_n;
    uint32_t first, count;
    sort_run(n, first, count);
    for (int b = threadIdx.x; b < a.nbins; b += BLOCK) bins[b] = 0;
    __syncthreads();
    for (uint32_t c = 0; c < count; ++c) {
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t i = ((first + c) * SORT_CHUNK_TILES + wave * SORT_TPW + s) * TILE + lane;
            const bool valid = i < n;
            const uint32_t key = valid ? sort_key(a.isect, i, a.nbins) : 0u;
            for_each_key(valid, key, [&](uint32_t k, uint64_t m) {
                if (lane == 0) atomicAdd(&bins[k], (uint32_t)__popcll((unsigned long long)m));
            });
        }
    }
    __syncthreads();
    // publish table[bin][workgroup] (write-through), then elect the last workgroup to scan it
    const bool last_bounce = a.depth == a.trace_depth - 1;
    for (int b = threadIdx.x; b < a.nbins; b += BLOCK) {
        const uint32_t cnt = (!compact || key_survives(a.scene.mats, (uint32_t)b, a.nbins, last_bounce)) ? bins[b] : 0u;
        __hip_atomic_store(&a.sort_table[(size_t)b * gridDim.x + blockIdx.x], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
    if (sctl[0]) {
        const uint32_t total = scan_words_inplace(a.sort_table, (uint32_t)a.nbins * gridDim.x, sctl + 2);
        if (threadIdx.x == 0) {
            if (compact) a.ctl->nlive[a.depth + 1] = total;
            if (a.depth == 0) a.ctl->nlive[0] = a.pool_n;
        }
    }
}

template <bool COMPACT>
__global__ __launch_bounds__(BLOCK) void k_shade_sorted(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    const int nb = (a.nbins + 3) & ~3;
    uint32_t *gbase = sctl + LDS_CTL_WORDS;          // [nb] output position of this workgroup's next path per key
    uint32_t *ktot = gbase + nb;                     // [nb] paths per key in the chunk
    uint32_t *kstart = ktot + nb;                    // [nb] first sorted position of the key in the chunk
    uint32_t *wcount = kstart + nb;                  // [WAVES][nb] per-wave counts, then running sorted positions
    uint32_t *order = wcount + WAVES * nb;           // [SORT_CHUNK] sorted position -> element of the chunk
    uint32_t *keyl = order + SORT_CHUNK;             // [SORT_CHUNK] element -> key
    float *mats = reinterpret_cast<float *>(keyl + SORT_CHUNK);       // materials (when they fit: a.nbins <= 64)
    const bool mats_lds = a.nbins <= 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    const uint32_t stamp = batch_stamp(a.fin_stamp, a.ctl);
    const uint32_t n = (COMPACT && a.depth > 0) ? a.ctl->nlive[a.depth] : a.pool_n;
    const bool last_bounce = a.depth == a.trace_depth - 1;
    uint32_t first, count;
    sort_run(n, first, count);
    for (int b = threadIdx.x; b < a.nbins; b += BLOCK) gbase[b] = a.sort_table[(size_t)b * gridDim.x + blockIdx.x];
    if (mats_lds)
        for (int k = threadIdx.x; k < a.scene.nmats * ptd::MAT_WORDS; k += BLOCK) mats[k] = a.scene.mats[k];
    const float *mat_src = mats_lds ? mats : a.scene.mats;
    uint32_t traced = 0;
    for (uint32_t c = 0; c < count; ++c) {
        const uint32_t chunk_base = (first + c) * SORT_CHUNK;
        // ---- A: keys of the wave's two tiles, per-wave counts ----
        for (int b = lane; b < a.nbins; b += 64) wcount[wave * nb + b] = 0;
        uint32_t key2[SORT_TPW];
        bool valid2[SORT_TPW];
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t e = (uint32_t)(wave * SORT_TPW + s) * TILE + lane;
            const uint32_t i = chunk_base + e;
            valid2[s] = i < n;
            key2[s] = valid2[s] ? sort_key(a.isect, i, a.nbins) : 0u;
            keyl[e] = key2[s];
            for_each_key(valid2[s], key2[s], [&](uint32_t k, uint64_t m) {
                if (lane == 0) wcount[wave * nb + k] += (uint32_t)__popcll((unsigned long long)m);
            });
        }
        __syncthreads();
        // ---- B: per key, counts -> starts of each wave's share; chunk totals; starts of the keys ----
        for (int b = threadIdx.x; b < a.nbins; b += BLOCK) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { const uint32_t v = wcount[w * nb + b]; wcount[w * nb + b] = run; run += v; }
            ktot[b] = run;
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t carry = 0;
            for (int base = 0; base < a.nbins; base += 64) {
                const uint32_t v = (base + lane < a.nbins) ? ktot[base + lane] : 0u;
                uint32_t incl = v;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t u = __shfl_up(incl, off);
                    if (lane >= off) incl += u;
                }
                if (base + lane < a.nbins) kstart[base + lane] = carry + incl - v;
                carry += (uint32_t)__shfl((int)incl, 63);
            }
        }
        __syncthreads();
        // ---- C: sorted position of every element (stable: tiles in order, lanes in order) ----
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t e = (uint32_t)(wave * SORT_TPW + s) * TILE + lane;
            for_each_key(valid2[s], key2[s], [&](uint32_t k, uint64_t m) {
                const uint32_t base = kstart[k] + wcount[wave * nb + k];           // same address for the whole wave
                if (valid2[s] && key2[s] == k) order[base + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1)))] = e;
                if (lane == 0) wcount[wave * nb + k] += (uint32_t)__popcll((unsigned long long)m);
            });
        }
        __syncthreads();
        // ---- D: shade 128 consecutive sorted positions per wave ----
        const uint32_t chunk_n = min((uint32_t)SORT_CHUNK, n - chunk_base);
#pragma unroll                       // both tiles' gathers in flight together: +5 % (profiles/r02/variants_sort.log)
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t p = (uint32_t)(wave * SORT_TPW + s) * TILE + lane;
            bool active = p < chunk_n;
            uint32_t key = 0, i = 0, pid = DEAD_PID, dst = 0;
            f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1), col = ptd::mk(1, 1, 1);
            if (active) {
                const uint32_t e = order[p];
                key = keyl[e];
                i = chunk_base + e;
                dst = gbase[key] + (p - kstart[key]);
                char *q = a.in.slot(i);
                pid = ppid(q);
                ro = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
                rd = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
                col = ptd::mk(pf(q, 6), pf(q, 7), pf(q, 8));
            }
            const bool have = active;
            if (pid == DEAD_PID) active = false;
            bool alive = false;
            ptd::PathState ps;
            ps.o = ro; ps.d = rd; ps.c = col;
            if (active) {
                const float t = at(a.isect.plane(0), i);
                const f3 nrm = ptd::mk(at(a.isect.plane(1), i), at(a.isect.plane(2), i), at(a.isect.plane(3), i));
                const int m = at(a.isect.mat(), i);
                const uint32_t smp = sample_of(a.map, pid);
                const int pixel = local_to_pixel(a.map, (int)(pid - smp * (uint32_t)a.map.tile_pixels));
                alive = ptd::shade_scatter(ps, t, nrm, m & 0x7fffffff, (m < 0) ? 0 : 1, mat_src, iter0 + (int)smp, pixel,
                                           a.depth, last_bounce);
                if (!alive) {
                    put_final(a.fin, pid, ps.c, stamp);
                }
            }
            traced += (uint32_t)__popcll((unsigned long long)ballot64(active));
            if (alive) {
                char *q = a.out.slot(dst);
                pf(q, 0) = ps.o.x; pf(q, 1) = ps.o.y; pf(q, 2) = ps.o.z;
                pf(q, 3) = ps.d.x; pf(q, 4) = ps.d.y; pf(q, 5) = ps.d.z;
                pf(q, 6) = ps.c.x; pf(q, 7) = ps.c.y; pf(q, 8) = ps.c.z;
                ppid(q) = pid;
            } else if (!COMPACT && have) {
                a.out.pid(dst) = DEAD_PID;
            }
        }
        __syncthreads();
        // ---- E: this workgroup's output positions move on ----
        for (int b = threadIdx.x; b < a.nbins; b += BLOCK)
            if (!COMPACT || key_survives(mat_src, (uint32_t)b, a.nbins, last_bounce)) gbase[b] += ktot[b];
        __syncthreads();
    }
    if (COMPACT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->alive[a.depth] = n;
    } else {
        if (lane == 0) sctl[8 + wave] = traced;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tb = sctl[8] + sctl[9] + sctl[10] + sctl[11];
            if (tb) atomicAdd(&a.ctl->alive[a.depth], tb);
        }
    }
}

// k_shade_sorted_w: the same result with WAVE-PRIVATE sorting, for up to 64 keys (lane k of a wave holds key k's
// counters in registers).  k_shade_sorted above spends its time between six workgroup barriers per 512-path chunk
// (per-wave counts -> per-key prefix -> key starts -> positions -> shade -> advance), each phase waiting for the
// slowest wave's memory latency: 34 us per chunk and workgroup on C3, of which ~2 us are instructions.  Here a wave
// sorts and shades ITS OWN 128 paths of the chunk (two tiles: stable counting sort through a 128-word LDS strip that
// only this wave touches, so LDS program order replaces the barriers), and the four waves of the workgroup meet once
// per chunk, to exchange their per-key counts: the output position of wave w's first key-k path is
//     gbase[k] + sum over w' < w of count_w'[k],
// the order of the workgroup-wide sort (chunks in order, elements in order), so the global result -- pool order after
// the bounce = stable partition of the stable sort by key -- is unchanged and k_sort_hist's per-workgroup table too.
// The exchange slots alternate by chunk parity: a wave that has passed barrier c cannot still be reading the slots of
// chunk c - 1, so one barrier per chunk is enough.  The gathers of a wave touch only its own two tiles' rows (at most
// four 128-B lines per instruction), so nothing is staged.
constexpr int SORTW_MAX_BINS = 64;
__host__ __device__ constexpr size_t shade_sorted_w_lds_words(int nmats) {
    return (size_t)LDS_CTL_WORDS + 2 * WAVES * 64 + (size_t)((nmats * ptd::MAT_WORDS + 3) & ~3);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off);
        if (lane >= off) v += u;
    }
    return v;
}

template <bool COMPACT, bool GEN = false>
__global__ __launch_bounds__(BLOCK, GEN ? 6 : 8) void k_shade_sorted_w(BounceArgs a) {
    static_assert(SORT_TPW == 2, "a wave handles two tiles per chunk");
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *xch = sctl + LDS_CTL_WORDS;                    // [2][WAVES][64]: per-key counts of each wave, by chunk parity
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mats = reinterpret_cast<float *>(xch + 2 * WAVES * 64);
    const int iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    const uint32_t stamp = batch_stamp(a.fin_stamp, a.ctl);
    const uint32_t n = (COMPACT && a.depth > 0) ? a.ctl->nlive[a.depth] : a.pool_n;
    const bool last_bounce = a.depth == a.trace_depth - 1;
    uint32_t first, count;
    sort_run(n, first, count);
    // lane k: where this workgroup's next path with key k goes, and whether paths with key k go on at all
    uint32_t gbase = lane < a.nbins ? a.sort_table[(size_t)lane * gridDim.x + blockIdx.x] : 0u;
    const bool key_lives = lane < a.nbins && (!COMPACT || key_survives(a.scene.mats, (uint32_t)lane, a.nbins, last_bounce));
    for (int k = threadIdx.x; k < a.scene.nmats * ptd::MAT_WORDS; k += BLOCK) mats[k] = a.scene.mats[k];
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1;
    uint32_t traced = 0;
    for (uint32_t c = 0; c < count; ++c) {
        const uint32_t sub_base = (first + c) * SORT_CHUNK + (uint32_t)wave * (SORT_TPW * TILE);
        // ---- the wave's two tiles, whole rows: state + intersection (every load coalesced, all in flight together) ----
        uint32_t idx[SORT_TPW], pid[SORT_TPW], key[SORT_TPW];
        bool valid[SORT_TPW];
        f3 ro[SORT_TPW], rd[SORT_TPW], col[SORT_TPW], nrm[SORT_TPW];
        float th[SORT_TPW];
        int mh[SORT_TPW];
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            idx[s] = sub_base + (uint32_t)s * TILE + lane;
            valid[s] = idx[s] < n;
            pid[s] = DEAD_PID; th[s] = -1.0f; mh[s] = 0;
            ro[s] = ptd::mk(0, 0, 0); rd[s] = ptd::mk(0, 0, 1); col[s] = ptd::mk(1, 1, 1); nrm[s] = ptd::mk(0, 0, 0);
            if (valid[s]) {
                th[s] = at(a.isect.plane(0), idx[s]);
                mh[s] = at(a.isect.mat(), idx[s]);
                nrm[s] = ptd::mk(at(a.isect.plane(1), idx[s]), at(a.isect.plane(2), idx[s]), at(a.isect.plane(3), idx[s]));
                if (GEN) {                                         // bounce 0 of a batch: the ray k_intersect<GEN> generated
                    pid[s] = idx[s];
                    const uint32_t smp = sample_of(a.map, pid[s]);
                    const int pixel = local_to_pixel(a.map, (int)(pid[s] - smp * (uint32_t)a.map.tile_pixels));
                    camera_ray(a.cam, a.lens, a.trace_depth, iter0 + (int)smp, pixel, a.map.W, ro[s], rd[s]);
                } else {
                    char *q = a.in.slot(idx[s]);
                    pid[s] = ppid(q);
                    ro[s] = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
                    rd[s] = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
                    col[s] = ptd::mk(pf(q, 6), pf(q, 7), pf(q, 8));
                }
            }
            key[s] = valid[s] ? (th[s] > 0.0f ? (uint32_t)(mh[s] & 0x7fffffff) : (uint32_t)(a.nbins - 1)) : 0u;
        }
        // ---- lane k counts key k over the wave's two tiles ----
        uint32_t cnt0 = 0, cnt1 = 0;
        for_each_key(valid[0], key[0], [&](uint32_t k, uint64_t m) { if ((uint32_t)lane == k) cnt0 = (uint32_t)__popcll((unsigned long long)m); });
        for_each_key(valid[1], key[1], [&](uint32_t k, uint64_t m) { if ((uint32_t)lane == k) cnt1 = (uint32_t)__popcll((unsigned long long)m); });
        uint32_t *slot = xch + (c & 1u) * (WAVES * 64);
        slot[wave * 64 + lane] = cnt0 + cnt1;
        __syncthreads();                                                   // the only barrier of the chunk: the waves' counts
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t v = slot[w * 64 + lane];
            if (w < wave) before += v;
            all += v;
        }
        const uint32_t g0 = gbase + before;                                // lane k: output slot of the wave's first key-k path
        const uint32_t g1 = g0 + cnt0;                                     //         ... of tile 1's first key-k path
        if (key_lives) gbase += all;
        // ---- output slots: stable within a key (tile 0's paths, then tile 1's, lanes in order) ----
        uint32_t dst[SORT_TPW] = {0u, 0u};
        for_each_key(valid[0], key[0], [&](uint32_t k, uint64_t m) {
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)g0, (int)k);
            if (valid[0] && key[0] == k) dst[0] = base + (uint32_t)__popcll((unsigned long long)(m & lt));
        });
        for_each_key(valid[1], key[1], [&](uint32_t k, uint64_t m) {
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)g1, (int)k);
            if (valid[1] && key[1] == k) dst[1] = base + (uint32_t)__popcll((unsigned long long)(m & lt));
        });
        // ---- shade in place (the order of shading is not observable; the output order is) ----
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const bool have = valid[s];
            const bool active = have && pid[s] != DEAD_PID;
            bool alive = false;
            ptd::PathState ps;
            ps.o = ro[s]; ps.d = rd[s]; ps.c = col[s];
            if (active) {
                const uint32_t smp = sample_of(a.map, pid[s]);
                const int pixel = local_to_pixel(a.map, (int)(pid[s] - smp * (uint32_t)a.map.tile_pixels));
                alive = ptd::shade_scatter(ps, th[s], nrm[s], mh[s] & 0x7fffffff, (mh[s] < 0) ? 0 : 1, mats, iter0 + (int)smp, pixel,
                                           a.depth, last_bounce);
                if (!alive) {
                    put_final(a.fin, pid[s], ps.c, stamp);
                }
            }
            traced += (uint32_t)__popcll((unsigned long long)ballot64(active));
            if (alive) {
                char *q = a.out.slot(dst[s]);
                pf(q, 0) = ps.o.x; pf(q, 1) = ps.o.y; pf(q, 2) = ps.o.z;
                pf(q, 3) = ps.d.x; pf(q, 4) = ps.d.y; pf(q, 5) = ps.d.z;
                pf(q, 6) = ps.c.x; pf(q, 7) = ps.c.y; pf(q, 8) = ps.c.z;
                ppid(q) = pid[s];
            } else if (!COMPACT && have) {
                a.out.pid(dst[s]) = DEAD_PID;
            }
        }
    }
    if (COMPACT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->alive[a.depth] = n;
    } else {
        if (lane == 0) sctl[8 + wave] = traced;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tb = sctl[8] + sctl[9] + sctl[10] + sctl[11];
            if (tb) atomicAdd(&a.ctl->alive[a.depth], tb);
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

// per-launch constants of a wave
// A field of the kernel's argument block read again where it is used (k_bounce: BounceArgs is the one kernel
// argument, so the field sits at its offset in the kernarg segment).  The pools' and the final-colour buffer's base
// pointers are used once per tile; kept in scalar registers across the tile loop they were spilled to VGPR lanes and
// came back through v_readlane -- vector-issue slots the kernel is bound by -- whereas a scalar load costs this wave
// a wait and the vector pipe nothing.  The empty asm hides the pointer's origin from the compiler, which would
// otherwise hoist the load out of the loop and keep the value alive again.
template <typename T>
__device__ __forceinline__ T karg_field(size_t offset) {
    const __attribute__((address_space(4))) char *kp = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const __attribute__((address_space(4))) T *)(kp + offset);
}
// a plain struct of the argument block (camera, lens, tile map), word by word behind ONE hidden pointer: the compiler
// merges the words into s_load_dwordx4 / x8 / x16
template <typename T>
__device__ __forceinline__ T karg_struct(size_t offset) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    const __attribute__((address_space(4))) char *kp = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; ++k) w[k] = *(const __attribute__((address_space(4))) uint32_t *)(kp + offset + 4 * k);
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}
// a Pool (base pointer + capacity) of the argument block
__device__ __forceinline__ Pool karg_pool(size_t offset) {
    return Pool{karg_field<float *>(offset + offsetof(Pool, base)), karg_field<uint32_t>(offset + offsetof(Pool, cap))};
}

struct TileCtx {
    bool kargs = false;         // k_bounce: pools and final colours through karg_field (a compile-time constant after inlining)
    bool kmisc = false;         // k_iteration: final colours, camera, lens, tile map through karg_field / karg_struct (its pools are locals)
    SceneAcc acc;               // per-lane gathers: materials, geom info, matrices (LDS or global)
    float *tri_lds;             // triangle tile (MESH_TILES)
    int lane, iter0;
    uint32_t stamp;             // of this batch's final colours (put_final)
};

// a tile in flight: what its shading needs besides the wave's LDS block (rays, best keys, winner records)
struct TileRegs {
    bool have, active;
    uint32_t i, src, tile, pid, smp;
    int pixel;
    f3 col;
    MeshBest mb;
};

// First half of one 64-path tile of one bounce: load (or generate) the paths.  `i` = logical path index (what
// MODE_ISECT planes and the mesh mask are keyed by), `src` = pool slot.
template <bool GEN>
__device__ __forceinline__ void tile_load(const BounceArgs &a, const TileCtx &c, const Pool &in,
                                          uint32_t tile, uint32_t i, uint32_t src, bool have, bool active,
                                          TileRegs &tr, f3 &ro, f3 &rd) {
    constexpr bool gen_rays = GEN;
    tr.have = have; tr.i = i; tr.src = src; tr.tile = tile;
    tr.pid = DEAD_PID; tr.smp = 0; tr.pixel = 0;
    tr.col = ptd::mk(1.0f, 1.0f, 1.0f);
    tr.mb.t = FLT_MAX; tr.mb.geom = -1; tr.mb.tri = -1;
    ro = ptd::mk(0, 0, 0); rd = ptd::mk(0, 0, 1);
    if (active) {
        if (gen_rays) {
            tr.pid = i;
        } else {
            // all ten fields of the slot in one burst of loads (one memory latency per tile)
            char *p = (c.kargs ? karg_pool(offsetof(BounceArgs, in)) : in).slot(src);
            tr.pid = ppid(p);
            ro = ptd::mk(pf(p, 0), pf(p, 1), pf(p, 2));
            rd = ptd::mk(pf(p, 3), pf(p, 4), pf(p, 5));
            tr.col = ptd::mk(pf(p, 6), pf(p, 7), pf(p, 8));
            if (tr.pid == DEAD_PID) active = false;
        }
    }
    if (active) {
        if (c.kmisc) {
            const TileMap map = karg_struct<TileMap>(offsetof(BounceArgs, map));
            tr.smp = sample_of(map, tr.pid);
            tr.pixel = local_to_pixel(map, (int)(tr.pid - tr.smp * (uint32_t)map.tile_pixels));
            if (gen_rays) camera_ray(karg_struct<pt_camera>(offsetof(BounceArgs, cam)), karg_struct<Lens>(offsetof(BounceArgs, lens)),
                                     karg_field<int>(offsetof(BounceArgs, trace_depth)), c.iter0 + (int)tr.smp, tr.pixel, map.W, ro, rd);
        } else {
            tr.smp = sample_of(a.map, tr.pid);
            tr.pixel = local_to_pixel(a.map, (int)(tr.pid - tr.smp * (uint32_t)a.map.tile_pixels));
            if (gen_rays) camera_ray(a.cam, a.lens, a.trace_depth, c.iter0 + (int)tr.smp, tr.pixel, a.map.W, ro, rd);
        }
    }
    tr.active = active;
}

// Second half: shade / scatter with the intersection (t, nrm, mat, outside), write the final colour of the paths
// that end here and append the survivors at dst_base + packed (wave64 ballot + popcount rank).
// does the ray reach one of the two root boxes of some mesh?  (wave-uniform scalar loads of the roots; the same
// conservative box arithmetic the walk uses)
__device__ __forceinline__ bool mesh_root_candidate(const int4 *bvh_meshes, int bvh_nmesh, const float *geoms, const float *bvh_nodes, f3 ro, f3 rd) {
    bool cand = false;
#pragma unroll 1
    for (int k = 0; k < bvh_nmesh; ++k) {
        const __attribute__((address_space(4))) int *mrec =
            (const __attribute__((address_space(4))) int *)(unsigned long long)(bvh_meshes + k);
        cfloat *grid = as_const(geoms) + (size_t)mrec[0] * ptd::GEOM_WORDS + ptd::G_INV;
        const __attribute__((address_space(4))) uint32_t *b =
            (const __attribute__((address_space(4))) uint32_t *)(unsigned long long)(bvh_nodes + (size_t)mrec[1] * BVH_NODE_WORDS);
        const BvhRay br = bvh_ray(ro, rd, ptd::mk(grid[0], grid[1], grid[2]), ptd::mk(grid[3], grid[4], grid[5]));
        float tn, tf;
        bvh_slab(br, b[0], b[1], b[2], tn, tf);
        cand |= tn <= tf;
        bvh_slab(br, b[3], b[4], b[5], tn, tf);
        cand |= tn <= tf;
    }
    return cand;
}
__device__ __forceinline__ bool mesh_root_candidate(const SceneDev &sc, f3 ro, f3 rd) {
    return mesh_root_candidate(sc.bvh_meshes, sc.bvh_nmesh, sc.geoms, sc.bvh_nodes, ro, rd);
}

// SORT (PT_SORT_MATERIAL, fused form): the survivors of key (= material hit) k go to the wave's span of range
// k * W + w -- `key_stride` slots further per key -- and `packed` is per LANE: lane k counts the wave's key-k survivors.
template <bool COMPACT, int MESH = MESH_NONE, bool SORT = false>
__device__ __forceinline__ void tile_shade(const BounceArgs &a, const TileCtx &c, const Pool &in, const Pool &out, int depth,
                                           const TileRegs &tr, f3 ro, f3 rd, float t, f3 nrm, int mat, int outside,
                                           uint32_t n, uint32_t dst_base, uint32_t &packed, uint32_t &traced,
                                           uint32_t key_stride = 0) {
    const int lane = c.lane;
    bool alive = false;
    ptd::PathState ps;
    ps.o = ro; ps.d = rd; ps.c = tr.col;
    if (tr.active) {
        alive = ptd::shade_scatter(ps, t, nrm, mat, outside, c.acc.mats, c.iter0 + (int)tr.smp, tr.pixel, depth,
                                   depth == a.trace_depth - 1);
        if (!alive) {
            put_final((c.kargs || c.kmisc) ? karg_field<float *>(offsetof(BounceArgs, fin)) : a.fin, tr.pid, ps.c, c.stamp);
        }
    }
    // ---- survivors append to the wave's packed run (wave64 ballot + popcount rank) ----
    const uint64_t bal = ballot64(alive);
    const uint64_t act = ballot64(tr.active);
    traced += (uint32_t)__popcll((unsigned long long)act);
    uint32_t dst = tr.i;
    if (COMPACT && SORT) {
        // one round per material among the tile's survivors (two to four on Cornell): stable within a key -- lanes in
        // order, tiles in order, waves in order (the directory is key-major)
        for_each_key(alive, (uint32_t)mat, [&](uint32_t k, uint64_t m) {
            const uint32_t have = (uint32_t)__builtin_amdgcn_readlane((int)packed, (int)k);
            if (alive && (uint32_t)mat == k) dst = k * key_stride + dst_base + have + rank_below(m);
            if ((uint32_t)lane == k) packed += (uint32_t)__popcll((unsigned long long)m);
        });
    } else if (COMPACT) {
        dst = dst_base + packed + rank_below(bal);
        packed += (uint32_t)__popcll((unsigned long long)bal);
    }
    if (alive) {
        char *p = (c.kargs ? karg_pool(offsetof(BounceArgs, out)) : out).slot(dst);
        pf(p, 0) = ps.o.x; pf(p, 1) = ps.o.y; pf(p, 2) = ps.o.z;
        pf(p, 3) = ps.d.x; pf(p, 4) = ps.d.y; pf(p, 5) = ps.d.z;
        pf(p, 6) = ps.c.x; pf(p, 7) = ps.c.y; pf(p, 8) = ps.c.z;
        ppid(p) = tr.pid;
        // mesh pre-pass of the NEXT bounce: flag the slot when the new ray can reach a mesh at all (~11 % of them on
        // C4), so that k_mesh neither scans nor loads the other 89 %
        if (MESH == MESH_PRE) {
            constexpr size_t SC = offsetof(BounceArgs, scene);
            const bool reach = c.kargs
                ? mesh_root_candidate(karg_field<const int4 *>(SC + offsetof(SceneDev, bvh_meshes)), karg_field<int>(SC + offsetof(SceneDev, bvh_nmesh)),
                                      karg_field<const float *>(SC + offsetof(SceneDev, geoms)), karg_field<const float *>(SC + offsetof(SceneDev, bvh_nodes)), ps.o, ps.d)
                : mesh_root_candidate(a.scene, ps.o, ps.d);
            if (reach)
                atomicOr(&(c.kargs ? karg_field<unsigned long long *>(offsetof(BounceArgs, mesh_flags_out)) : a.mesh_flags_out)[dst >> 6], 1ull << (dst & 63u));
        }
    } else if (!COMPACT && tr.have && tr.i < n) {
        out.pid(dst) = DEAD_PID;
    }
}

// the tile with parity `par` has been fully tested: read its rays back from the wave's LDS block, fold the
// winner and shade
template <bool COMPACT, int MESH, bool SORT = false>
__device__ __forceinline__ void tile_finish(const BounceArgs &a, const TileCtx &c, const WaveQ &q, int par, const Pool &in,
                                            const Pool &out, int depth, const TileRegs &tr, uint32_t n, uint32_t dst_base,
                                            uint32_t &packed, uint32_t &traced, uint32_t key_stride = 0) {
    const float *ry = q.rays(par) + c.lane;
    const f3 ro = ptd::mk(ry[0], ry[64], ry[128]);
    const f3 rd = ptd::mk(ry[192], ry[256], ry[320]);
    float t = -1.0f; f3 nrm = ptd::mk(0, 0, 0); int mat = 0, outside = 1;
    if (tr.active) tile_result(q, par, c.acc, a.scene.tris, tr.mb, t, nrm, mat, outside);
    tile_shade<COMPACT, MESH, SORT>(a, c, in, out, depth, tr, ro, rd, t, nrm, mat, outside, n, dst_base, packed, traced, key_stride);
}

// The tiles [first, first + count) of one wave's run at one bounce, two in flight (see the intersection stages
// above): tile T+1 is loaded and culled before tile T is shaded, so T's last candidates share a pass with T+1's
// first.  `logical0` = logical index of the run's first path; with `own_span` (k_iteration) the paths sit densely
// in the wave's own span and `live` of them exist.
// k_iteration, bounces >= 1: the survivors of a WORKGROUP's four waves, each packed at the front of its wave's span,
// read as one sequence -- wave s holds the workgroup-logical paths [p[s], p[s+1]) (p[0] = 0, p[4] = total) in the
// slots b[s] + (L - p[s]).  Wave-uniform.
struct WgSpans {
    uint32_t p1, p2, p3, total;
    uint32_t b0, b1, b2, b3;
};

// GEN: bounce 0 of a batch generates the camera rays in registers (a compile-time switch: the camera, the lens and
// the candidate masks then never occupy scalar registers in the kernels of the other bounces, and the pool's input
// side never does in bounce 0's)
template <int MODE, bool COMPACT, int MESH, bool GEN, bool SORT = false>
__device__ __forceinline__ void run_tiles(const BounceArgs &a, const TileCtx &c, WaveQ &q, const Pool &in, const Pool &out,
                                          int depth, uint32_t first_tile, uint32_t count, uint32_t tiles,
                                          uint32_t n, bool packed_in, uint32_t span_in, uint32_t &cur, uint32_t dst_base,
                                          bool own_span, const WgSpans &ws, uint32_t &packed, uint32_t &traced,
                                          uint32_t key_stride = 0) {
    const int lane = c.lane;
    bool pending = false;
    TileRegs prev{};
    uint32_t prev_ticket = 0;
    int par = 0;
    // bounce 0 of a pinhole camera: pool tile t holds the pixels of camera tile t mod (tiles per sample), whose
    // candidate primitives k_cull0_mask has written down
    const bool masked = MODE == MODE_FUSED && GEN && a.cull0 != nullptr;
    uint32_t mtile = masked ? first_tile % a.cull0_tiles : 0u;
    for (uint32_t r = 0; r < count; ++r) {
        const uint32_t tile = first_tile + r;
        if (!own_span && tile >= tiles) break;
        rotate_priority(r + (uint32_t)depth, PT_MIN_WAVES + 1);
        unsigned long long gmask = 0;
        if (masked) {
            gmask = ((const __attribute__((address_space(4))) unsigned long long *)(unsigned long long)a.cull0)[mtile];
            if (++mtile == a.cull0_tiles) mtile = 0;
        }
        bool have, active;
        uint32_t i, src;
        if (own_span) {                                   // k_iteration: tile `tile` of the workgroup's survivors
            const uint32_t L = tile * TILE + lane;
            have = true; active = L < ws.total;
            uint32_t off = L, b = ws.b0;                      // the span that holds L: three compares, wave s's span
            if (L >= ws.p1) { off = L - ws.p1; b = ws.b1; }
            if (L >= ws.p2) { off = L - ws.p2; b = ws.b2; }
            if (L >= ws.p3) { off = L - ws.p3; b = ws.b3; }
            src = b + off; i = src;
        } else {
            have = tile < tiles;
            i = tile * TILE + lane;                        // logical path index
            active = have && i < n;
            src = i;
            if (packed_in && have) src = resolve_src(a.dir_in, span_in, cur, i, active, a.ctl);
        }
        TileRegs tr;
        f3 ro, rd;
        tile_load<GEN>(a, c, in, tile, i, src, have, active, tr, ro, rd);
        if (MODE == MODE_FUSED) {
            const float4 *pre_hit = nullptr;
            if (MESH == MESH_PRE && tr.active) {
                // slots whose flag is set carry a mesh result from k_mesh (a hit, or "walked, nothing hit")
                const unsigned long long *fl = c.kargs ? karg_field<unsigned long long *>(offsetof(BounceArgs, mesh_flags_in)) : a.mesh_flags_in;
                if ((fl[src >> 6] >> (src & 63u)) & 1ull) pre_hit = (c.kargs ? karg_field<float4 *>(offsetof(BounceArgs, mesh_hit)) : a.mesh_hit) + src;
            }
            cull_scene<MESH>(a.scene, c.acc, q, par, c.tri_lds, tr.active, ro, rd, tr.mb, pre_hit, masked, gmask);
            const uint32_t ticket = q.total;
            if (pending) {
                drain_to(q, c.acc, prev_ticket);
                tile_finish<COMPACT, MESH, SORT>(a, c, q, par ^ 1, in, out, depth, prev, n, dst_base, packed, traced, key_stride);
            }
            prev = tr; prev_ticket = ticket; pending = true; par ^= 1;
        } else {
            // MODE_ISECT: planes in logical order; MODE_CACHE0: one record per pixel of the tile
            float t = -1.0f; f3 nrm = ptd::mk(0, 0, 0); int mat = 0, outside = 1;
            if (tr.active) {
                const uint32_t k = (MODE == MODE_CACHE0) ? tr.pid - tr.smp * (uint32_t)a.map.tile_pixels : i;
                t = at(a.isect.plane(0), k);
                nrm = ptd::mk(at(a.isect.plane(1), k), at(a.isect.plane(2), k), at(a.isect.plane(3), k));
                const int m = at(a.isect.mat(), k);
                mat = m & 0x7fffffff; outside = (m < 0) ? 0 : 1;
            }
            tile_shade<COMPACT>(a, c, in, out, depth, tr, ro, rd, t, nrm, mat, outside, n, dst_base, packed, traced);
        }
    }
    if (pending) {
        drain_to(q, c.acc, prev_ticket);
        tile_finish<COMPACT, MESH, SORT>(a, c, q, par ^ 1, in, out, depth, prev, n, dst_base, packed, traced, key_stride);
    }
}

#ifdef PT_WAVE_TIMES
// diagnostic build (profiles/wave_times.py): start / end time (100 MHz ticks) and hardware slot of every wave of
// k_bounce, per bounce -- what showed the arbiter's oldest-first order (rotate_priority)
__device__ unsigned long long g_wave_times[8][8192][2];
__device__ uint32_t g_wave_hw[8][8192];
#endif

template <int MODE, bool COMPACT, int MESH, bool SLDS, bool GEN = false, bool SORT = false>
__global__ __launch_bounds__(BLOCK, MESH == MESH_TILES ? PT_LOOP_WAVES : (MESH == MESH_PRE && PT_PRE_WAVES > PT_MIN_WAVES) ? PT_PRE_WAVES : (SORT && MODE == MODE_FUSED && MESH == MESH_NONE) ? PT_SORT_WAVES : (MODE == MODE_FUSED && COMPACT && MESH == MESH_NONE && !SORT && PT_FUSED_WAVES > PT_MIN_WAVES) ? PT_FUSED_WAVES : PT_MIN_WAVES) void k_bounce(BounceArgs a) {
#ifdef PT_WAVE_TIMES
    const unsigned long long wt0 = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    const LdsCarve lc = carve_lds(lds_raw, a.scene, SLDS);
    TileCtx c;
#ifndef PT_NO_KARG_RELOAD
    // C2 +0.9 %, C3 +0.5 %, C3 sorted +2.2 % (ten scalar spills fewer); the every-triangle loop measured 1 % slower with it
    // (profiles/r03/variants_karg_reload.log)
    c.kargs = MESH != MESH_TILES;
#endif
    c.tri_lds = lc.tri;
#ifdef PT_STAMPS
#define STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && a.depth == PT_STAMPS) a.ctl->stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
    STAMP(0);
    c.acc = stage_scene<SLDS>(lc.scene, a.scene);
    STAMP(1);
    WaveQ q{lc.pw, 0, 0};
    const int lane = threadIdx.x & 63;
    c.lane = lane;
    const uint32_t Wp = gridDim.x * WAVES;                        // waves of the grid
    // runs of tiles the pool is cut into: one per wave -- or, with the material sort, several (RangeDir::W = S * Wp).
    // Consecutive logical tiles of a sorted pool hold paths that all hit the SAME material at the last bounce, and what a
    // path costs depends on where it has just been (a run of paths that left the glass ball is all sphere candidates):
    // with one run per wave the launch waited 60-150 us for its slowest wave.  Wave w takes the runs w, Wp + w, ...:
    // a share of every part of the key space.
    const uint32_t W = (SORT && COMPACT) ? a.dir_out.W : Wp;
    const uint32_t runs_per_wave = (SORT && COMPACT) ? W / Wp : 1u;
    const uint32_t wid0 = run_id();
    c.iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;       // graph replay: arguments are frozen
    c.stamp = batch_stamp(a.fin_stamp, a.ctl);
    const uint32_t n = (COMPACT && !GEN) ? a.ctl->nlive[a.depth] : a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);                        // logical tiles per run (contiguous)
    const bool packed_in = COMPACT && !GEN && a.dir_in.mem != nullptr;
    const uint32_t span_in = packed_in ? range_tiles(a.ctl->nlive[a.depth - 1], a.dir_in.W) * TILE : 0;
    uint32_t traced = 0;
    if (GEN && blockIdx.x == 0 && threadIdx.x == 0) a.ctl->nlive[0] = a.pool_n;   // k_raygen's job otherwise
    for (uint32_t j = 0; j < runs_per_wave; ++j) {
        const uint32_t wid = j * Wp + wid0;
        uint32_t packed = 0;                                     // survivors of this run written so far (wave-uniform; SORT: lane k counts key k)
        uint32_t cur = 0;                                        // source range of the run's current position
        if (packed_in && wid * R < tiles) cur = find_range(a.dir_in.base(), a.dir_in.nr, wid * R * TILE);
        STAMP(2);
        // the run's R consecutive 64-path tiles; no workgroup barrier inside the loop
        run_tiles<MODE, COMPACT, MESH, GEN, SORT>(a, c, q, a.in, a.out, a.depth, wid * R, R, tiles, n, packed_in, span_in,
                                                  cur, wid * R * TILE, false, WgSpans{}, packed, traced, W * R * TILE);
        if (COMPACT) {
            // every run publishes its range count(s); the last workgroup out scans them
            if (SORT) {
                if ((uint32_t)lane * W < a.dir_out.nr)                    // lane k: the run's key-k survivors, range k * W + run
                    __hip_atomic_store(&a.dir_out.count()[(uint32_t)lane * W + wid], packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (lane == 0)
                __hip_atomic_store(&a.dir_out.count()[wid], packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const uint32_t wid = wid0;
    STAMP(6);
#ifdef PT_WAVE_TIMES
    if (lane == 0 && a.depth < 8 && wid < 8192) {
        g_wave_times[a.depth][wid][0] = wt0; g_wave_times[a.depth][wid][1] = __builtin_amdgcn_s_memrealtime();
        uint32_t xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_wave_hw[a.depth][wid] = (xcc & 0xf) | (hwid << 4);
    }
#endif
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's count stores have left
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

// k_iteration's traced counts, iter_counts[bounce][workgroup] (plain stores, nothing cleared beforehand), added up by ONE
// workgroup of BLOCK threads: Control::alive[bounce], the session's persistent counters and -- synchronous calls -- the
// page-locked pt_stats block.  Eight bounces per pass: thread t takes bounce t / 32 and every 32nd workgroup from t % 32
// on, the 32 partial sums of a bounce meet in a half-wave shuffle.  (A serial loop over the bounces with two barriers
// each, tried first at the end of k_iteration, cost every launch ~30 us of tail: 1 spp per call 27.3 -> 24.8 Grays/s.)
__device__ __forceinline__ void fold_iter_counts(const uint32_t *counts, uint32_t G, int depth, Control *ctl, Persist *per, HostStats *hs,
                                                 uint32_t iterations, uint32_t serial, uint32_t *lds /* >= BLOCK / 32 words */) {
    constexpr int PER_PASS = BLOCK / 32;
    unsigned long long rays = 0;
    uint32_t first = 0;
    for (int d0 = 0; d0 < depth; d0 += PER_PASS) {
        const int d = d0 + (int)(threadIdx.x >> 5);
        uint32_t sum = 0;
        if (d < depth)
            for (uint32_t b = threadIdx.x & 31u; b < G; b += 32u)
                sum += __hip_atomic_load(&counts[(uint32_t)d * G + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int off = 16; off > 0; off >>= 1) sum += __shfl_down(sum, off, 32);
        if ((threadIdx.x & 31u) == 0) lds[threadIdx.x >> 5] = sum;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < PER_PASS && d0 + k < depth; ++k) {
                const uint32_t tot = lds[k];
                ctl->alive[d0 + k] = tot;
                if (hs) hs->alive[d0 + k] = tot;
                rays += tot;
                if (d0 + k == 0) first = tot;
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&per->rays, rays);
        atomicAdd(&per->iterations, (unsigned long long)iterations);
        atomicAdd(&per->first_rays, (unsigned long long)first);
        if (hs) {
            for (int d = depth; d <= MAX_DEPTH; ++d) hs->alive[d] = 0;
            hs->error = 0;
            __hip_atomic_store(&hs->serial, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// A whole batch in ONE launch, for small batches (the reference's calling pattern is one iteration per
// call): at 1 spp every bounce kernel is ~20 us of fixed cost (launch, scene staging, directory search,
// last-workgroup scan) around a few microseconds of work.  Here every wave generates the camera rays of its
// run of tiles; from then on the survivors stay inside the WORKGROUP: every wave packs its survivors at the front
// of its own span (the two pools ping-pong inside the launch), and bounce d+1 deals the four spans of the workgroup
// out again to its four waves (one barrier per bounce; WgSpans).  No exchange between workgroups, no directory.
// The paths are not dealt out again across the whole grid after every bounce, which costs load balance (a
// workgroup whose pixels live long works longer) -- the price that makes this the small-batch path only.  Traced
// counts go to 32 partial sums per bounce (Control::bucket[d][1]; a same-address atomic per wave would
// serialise), folded by k_gather.
//
// A wave's stores of bounce d are read back by the waves of its workgroup at bounce d+1 through the CU's vector
// L1, which the write-through stores update: workgroup scope is enough for that, on the condition that the
// workgroup runs in CU mode (not tgsplit: a workgroup's waves then share one CU and one L1) -- the mode hipcc
// compiles for by default and the only one this library is built in (build.py passes no -mtgsplit).
template <bool SLDS>
__global__ __launch_bounds__(BLOCK, PT_ITER_WAVES) void k_iteration(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const LdsCarve lc = carve_lds(lds_raw, a.scene, SLDS);
    TileCtx c;
#ifndef PT_NO_KARG_RELOAD
    // 71 -> 28 scalar spills, 92 -> 81 VGPRs: 1 spp 26.0 -> 27.3, 4 spp 35.5 -> 36.8 Grays/s (profiles/r03/variants_karg_iter.log)
    c.kmisc = true;
#endif
    c.tri_lds = nullptr;
    c.acc = stage_scene<SLDS>(lc.scene, a.scene);
    WaveQ q{lc.pw, 0, 0};
    const int lane = threadIdx.x & 63;
    c.lane = lane;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    c.iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    c.stamp = batch_stamp(a.fin_stamp, a.ctl);
    const uint32_t n = a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);
    const uint32_t base = wid * R * TILE;                         // this wave's span in both pools
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->nlive[0] = n;
    Pool in = a.in, out = a.out;
    uint32_t cur = 0;
    // Bounces >= 1 deal the WORKGROUP's survivors out again: every wave packs its survivors at the front of its own span
    // (no exchange inside a bounce), the four counts cross through LDS at one barrier per bounce, and wave w then takes
    // the w-th quarter of the tiles of the four spans read as one sequence (WgSpans).  Left with its own survivors only,
    // a wave ran half-empty tiles from bounce 2 on (36 paths in a tile of 64 at bounce 5): 11 tile passes per wave and
    // iteration at 1 spp instead of 8.5.  Slots alternate by bounce parity, so one barrier per bounce is enough.
    uint32_t *xcnt = reinterpret_cast<uint32_t *>(lds_raw);      // [2][survivors of WAVES | traced by WAVES] (the 16 LDS control words)
    const int wave = threadIdx.x >> 6;
    WgSpans ws{};
    ws.b0 = (0u * gridDim.x + blockIdx.x) * R * TILE; ws.b1 = (1u * gridDim.x + blockIdx.x) * R * TILE;
    ws.b2 = (2u * gridDim.x + blockIdx.x) * R * TILE; ws.b3 = (3u * gridDim.x + blockIdx.x) * R * TILE;
    static_assert(WAVES == 4, "four spans per workgroup");
    for (int d = 0; d < a.trace_depth; ++d) {
        uint32_t traced = 0, packed = 0;
        if (d == 0) {
            run_tiles<MODE_FUSED, true, MESH_NONE, true>(a, c, q, in, out, 0, wid * R, R, tiles, n, false, 0, cur, base, false, ws,
                                                         packed, traced);
        } else {
            const uint32_t wg_tiles = (ws.total + TILE - 1) / TILE;
            const uint32_t per = (wg_tiles + WAVES - 1) / WAVES;              // <= R: a wave's output still fits its span
            const uint32_t first = (uint32_t)wave * per;
            const uint32_t mine = first < wg_tiles ? min(per, wg_tiles - first) : 0u;
            run_tiles<MODE_FUSED, true, MESH_NONE, false>(a, c, q, in, out, d, first, mine, tiles, n, false, 0, cur, base, true, ws,
                                                          packed, traced);
        }
        // this wave's survivors are read by the workgroup's other waves at the next bounce, through the CU's vector L1
        // that the write-through stores went through: workgroup scope (CU mode, see above); an agent-scope fence writes
        // back / invalidates the L2 and made the launch 4x slower
        uint32_t *slot = xcnt + (d & 1) * (2 * WAVES);
        if (lane == 0) { slot[wave] = packed; slot[WAVES + wave] = traced; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                                     // every wave, every bounce: exits are uniform
        const uint32_t c0 = slot[0], c1 = slot[1], c2 = slot[2], c3 = slot[3];
        ws.p1 = c0; ws.p2 = c0 + c1; ws.p3 = c0 + c1 + c2; ws.total = c0 + c1 + c2 + c3;
        // paths this workgroup traced at bounce d: one plain (write-through) store into its own word of
        // iter_counts[bounce][workgroup] -- nothing to clear before the launch, no same-address atomics; the launch's
        // last workgroup adds the columns up.  A workgroup that runs out of paths writes the zeros of its later bounces.
        // (the pointers this kernel needs once per bounce or once at its end are read from the kernel-argument segment
        // where they are used, like the camera: kept in scalar registers across the tile loops they were spilled)
        if (threadIdx.x == 0)
            __hip_atomic_store(&karg_field<uint32_t *>(offsetof(BounceArgs, iter_counts))[(uint32_t)d * gridDim.x + blockIdx.x],
                               slot[WAVES] + slot[WAVES + 1] + slot[WAVES + 2] + slot[WAVES + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ws.total == 0) {
            if ((int)threadIdx.x > d && (int)threadIdx.x < karg_field<int>(offsetof(BounceArgs, trace_depth)))
                __hip_atomic_store(&karg_field<uint32_t *>(offsetof(BounceArgs, iter_counts))[threadIdx.x * gridDim.x + blockIdx.x], 0u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const Pool tmp = in; in = out; out = tmp;
    }
    // pathtrace() per call with a host image (the reference's pattern, pathtrace.cu:380-392): at 1 spp a wave owns the
    // pixels of its run of tiles through every bounce, so when it is done their final colours are all its own stores
    // and it can do finalGather for them itself -- image[pixel] += colour -- and write the new sums straight into the
    // caller's page-locked image (mapped into the device's address space), while other waves still trace: the 7.68 MB
    // that used to cross PCIe AFTER the iteration now cross during it.  The 192 dwords of a tile's 64 float3 pixels
    // are transposed through the wave's LDS block so that every store instruction writes 256 contiguous bytes
    // (whole lines for the PCIe write combiner), not 64 dwords 12 bytes apart.
    float *const epi_image = karg_field<float *>(offsetof(BounceArgs, epi_image));
    if (epi_image) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's final colours have left the CU
        float *const epi_host = karg_field<float *>(offsetof(BounceArgs, epi_host));
        const float *const fin = karg_field<float *>(offsetof(BounceArgs, fin));
        const TileMap map = karg_struct<TileMap>(offsetof(BounceArgs, map));
        float *tr = lc.pw;                                          // the wave's LDS block is free now
        for (uint32_t r = 0; r < R; ++r) {
            const uint32_t tile = wid * R + r;
            if (tile >= tiles) break;
            const uint32_t j = tile * TILE + lane;                  // one sample: pid == local pixel
            float cx = 0.0f, cy = 0.0f, cz = 0.0f;
            if (j < n) {                                            // agent-scope loads: from the L2 the stores went to
                const float *f = fin + (size_t)j * 4;
                if (__float_as_uint(__hip_atomic_load(f + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == c.stamp) {
                    cx = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cy = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cz = __hip_atomic_load(f + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            tr[3 * lane] = cx; tr[3 * lane + 1] = cy; tr[3 * lane + 2] = cz;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t w = (uint32_t)k * TILE + lane;       // dword of the tile's 192
                const uint32_t jl = w / 3u;
                const uint32_t jj = tile * TILE + jl;
                if (jj < n) {
                    const size_t idx = (size_t)local_to_pixel(map, (int)jj) * 3 + (w - jl * 3u);
                    const float v = epi_image[idx] + tr[w];
                    epi_image[idx] = v;
                    if (epi_host) epi_host[idx] = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    // ---- a launch that did its own finalGather also folds its traced counts: its last workgroup out adds the columns
    // of iter_counts up (fold_iter_counts) -- no k_gather runs behind it.  Otherwise k_gather's first workgroup does.
    if (epi_image) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's count and image stores have left
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool last = elect_last_self_clearing(karg_field<Control *>(offsetof(BounceArgs, ctl))->ticket);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            xcnt[0] = last ? 1u : 0u;
        }
        __syncthreads();
        if (xcnt[0])
            fold_iter_counts(karg_field<uint32_t *>(offsetof(BounceArgs, iter_counts)), gridDim.x, karg_field<int>(offsetof(BounceArgs, trace_depth)),
                             karg_field<Control *>(offsetof(BounceArgs, ctl)), karg_field<Persist *>(offsetof(BounceArgs, persist)),
                             karg_field<HostStats *>(offsetof(BounceArgs, host_stats)),
                             n / (uint32_t)karg_field<int>(offsetof(BounceArgs, map) + offsetof(TileMap, tile_pixels)), c.stamp, xcnt + 4);
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
// to mesh_hit[slot] = {t, geom, triangle} with one flag per pool slot (BounceArgs::mesh_flags_*); k_bounce
// <MESH_PRE> folds them with the geom-index tie-break of pathtrace.cu:192 and flags, among the survivors it
// writes, the ones whose new ray can reach a mesh: the next bounce's k_mesh touches only those.
// ---------------------------------------------------------------------------
#ifndef PT_SKIP_PAIRS
#define PT_SKIP_PAIRS 2                      // missed-sibling pairs remembered per walk (registers)
#endif
#ifndef PT_MESH_WAVES
#define PT_MESH_WAVES 4                      // waves per SIMD k_mesh is register-budgeted for
#endif
#ifndef PT_MESH_BLOCK
#define PT_MESH_BLOCK 1024
#endif
constexpr int MESH_BLOCK = PT_MESH_BLOCK;     // k_mesh: ONE workgroup of 16 waves per CU, so that the CU's waves share one LDS copy
constexpr int MESH_WG_WAVES = MESH_BLOCK / 64;   // of the tops of the trees
constexpr int MQ_SLOTS = 128;                 // ray ring entries per wave (a tile adds <= 64 while < 64 wait)
constexpr int TQ_SLOTS = 512;                 // triangle ring entries per wave (a step adds <= 64 * 2 * LEAF_MAX while < 64 wait)
constexpr int MQ_RAY_WORDS = 8 * MQ_SLOTS;    // src, path, origin xyz, direction xyz
constexpr int MQ_WORDS = MQ_RAY_WORDS + TQ_SLOTS + 2 * 64;   // + triangle ring + the 64 per-lane best keys (u64)
constexpr int MESH_TAB = 8;                   // meshes whose {geom, root, top, grid} sit in LDS: starting a walk then costs no global load
constexpr int MESH_TAB_WORDS = 12;            //   geom root top - | origin xyz step x | step yz - -
constexpr size_t MESH_LDS_BYTES = ((size_t)MESH_WG_WAVES * MQ_WORDS + (size_t)BVH_TOP * BVH_TOP_STRIDE + MESH_TAB * MESH_TAB_WORDS) * 4;   // 147 840 of 163 840
static_assert(MESH_LDS_BYTES <= 160 * 1024, "k_mesh: per-wave rings + tree tops must fit one CU's LDS");
#ifndef PT_MQ_STEPS
#define PT_MQ_STEPS 8
#endif
#ifndef PT_MQ_LEAVE
#define PT_MQ_LEAVE 56
#endif
constexpr int MQ_STEPS = PT_MQ_STEPS;         // walk steps between two looks at the ray ring
constexpr int MQ_LEAVE = PT_MQ_LEAVE;         // lanes still busy when the wave goes back to scanning
static_assert(2 * PT_LEAF_MAX * 64 + 63 <= TQ_SLOTS, "a step's triangles must fit beside the waiting ones");
constexpr int NT_BITS = 2 * PT_LEAF_MAX < 2 ? 1 : 2 * PT_LEAF_MAX < 4 ? 2 : 2 * PT_LEAF_MAX < 8 ? 3 : 4;   // bits of a step's triangle count per lane

// per-lane state of a walk in flight; it survives across the scanning of further tiles
struct MeshWalker {
    bool have;
    uint32_t src, path;
    BvhRay ray;
    int mesh, node, steps;            // position in SceneDev::bvh_meshes, record in that mesh's tree
    int geom, root;                   // of the current mesh
    uint32_t top;                     // its records [0, top >> 16) sit in LDS from record slot (top & 0xffff) on
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
        const f3 v0 = ptd::mk(P.x, P.y, P.z), e1 = ptd::mk(P.w, Q.x, Q.y), e2 = ptd::mk(Q.z, Q.w, S.x);
        if (ptd::ray_triangle(ro, rd, v0, e1, e2, tz) && tz > 0.0f && ptd::tri_point_ok(ro, rd, tz, v0, e1, e2, S.z))
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
// Point walker `w` at mesh number k (position in SceneDev::bvh_meshes) for the ray (ro, rd).  The first MESH_TAB
// meshes' entries and grids are read from the LDS table k_mesh stages; the rest from the scene buffers.
__device__ __forceinline__ void mesh_begin(MeshWalker &w, const float *mtab, const BounceArgs &a, int k, f3 ro, f3 rd) {
    if (k < MESH_TAB) {
        // (read through the LDS address space: with generic pointers the compiler merges this branch and the other into
        // flat loads on a selected pointer, see bvh_fetch_top)
        const uint32_t off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)mtab + (uint32_t)k * (uint32_t)(MESH_TAB_WORDS * 4);
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x4_t h = *(__attribute__((address_space(3))) const f32x4_t *)(size_t)off;
        const f32x4_t g0 = *(__attribute__((address_space(3))) const f32x4_t *)(size_t)(off + 16);
        const f32x2_t g1 = *(__attribute__((address_space(3))) const f32x2_t *)(size_t)(off + 32);
        w.geom = __float_as_int(h.x); w.root = __float_as_int(h.y); w.top = __float_as_uint(h.z);
        w.ray = bvh_ray(ro, rd, ptd::mk(g0.x, g0.y, g0.z), ptd::mk(g0.w, g1.x, g1.y));
    } else {
        const int4 m = a.scene.bvh_meshes[k];
        w.geom = m.x; w.root = m.y; w.top = (uint32_t)m.w;
        w.ray = mesh_ray(a.scene, m.x, ro, rd);
    }
    w.mesh = k; w.node = 0; w.steps = 0;
}

// Lanes without a walk take the next ray-ring entries.  Returns the ballot of the lanes that have one.
__device__ __forceinline__ uint64_t mesh_refill(MeshWalker &w, float *mq, const float *mtab, MeshRings &rg, const BounceArgs &a) {
    const int lane = threadIdx.x & 63;
    const uint32_t *mi = reinterpret_cast<const uint32_t *>(mq);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(mq + MQ_RAY_WORDS + TQ_SLOTS);
    const uint64_t below = (1ull << lane) - 1;
    const uint64_t idle = ballot64(!w.have);
    const uint32_t avail = rg.q_total - rg.q_head;
    if (idle && avail) {
        const uint32_t rank = rank_below(idle);
        if (!w.have && rank < avail) {
            const uint32_t s = (rg.q_head + rank) & (MQ_SLOTS - 1);
            w.src = mi[0 * MQ_SLOTS + s]; w.path = mi[1 * MQ_SLOTS + s];
            mesh_begin(w, mtab, a, 0, ptd::mk(mq[2 * MQ_SLOTS + s], mq[3 * MQ_SLOTS + s], mq[4 * MQ_SLOTS + s]),
                       ptd::mk(mq[5 * MQ_SLOTS + s], mq[6 * MQ_SLOTS + s], mq[7 * MQ_SLOTS + s]));
            w.ticket = rg.t_head;
#pragma unroll
            for (int u = 0; u < PT_SKIP_PAIRS; ++u) { w.skip[u] = -1; w.to[u] = -1; }
            w.best_t = FLT_MAX; w.best_geom = -1; w.best_tri = -1;
            keys[lane] = TRI_KEY_NONE;
            w.have = true;
        }
        rg.q_head += min((uint32_t)__popcll((unsigned long long)idle), avail);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    return ballot64(w.have);
}

// Every lane with a walk advances MQ_STEPS records (see mesh_drain).
__device__ __forceinline__ void mesh_steps(MeshWalker &w, float *mq, const float *tops, const float *mtab, MeshRings &rg, const BounceArgs &a) {
    const int lane = threadIdx.x & 63;
    const uint32_t *mi = reinterpret_cast<const uint32_t *>(mq);
    uint32_t *tq = reinterpret_cast<uint32_t *>(mq + MQ_RAY_WORDS);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(mq + MQ_RAY_WORDS + TQ_SLOTS);
    const uint64_t below = (1ull << lane) - 1;
    const uint32_t tops_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float *)tops;   // LDS byte address
    // A walk that is over (and whose queued triangles have been tested) folds its mesh's winner and publishes its
    // result or moves on to the next mesh.  Lanes get new walks only between blocks of MQ_STEPS steps (mesh_drain), so
    // with ONE mesh this runs once per block, for all the lanes that finished during it together -- per step it ran a
    // couple of lanes wide on most steps (72 % of them had some lane finishing).  Several meshes: per step, so that a
    // lane's next mesh starts at once.
    const bool multi = a.scene.bvh_nmesh > 1;
    auto finish = [&]() {
        if (w.have && w.node < 0 && (int32_t)(rg.t_head - w.ticket) >= 0) {    // this mesh is done and fully tested
            const unsigned long long key = keys[lane];
            if ((uint32_t)key != 0xffffffffu) {                      // completion spec 8.0: distance to origin + dir * bary.z
                const float tz = __uint_as_float((uint32_t)(key >> 32));
                const f3 p = ptd::add(w.ray.ro, ptd::scale(w.ray.rd, tz));
                const float t = ptd::length(ptd::sub(w.ray.ro, p));
                if (t > 0.0f && w.best_t > t) { w.best_t = t; w.best_geom = w.geom; w.best_tri = (int)(uint32_t)key; }
            }
            if (w.mesh + 1 < a.scene.bvh_nmesh) {
                mesh_begin(w, mtab, a, w.mesh + 1, w.ray.ro, w.ray.rd);
#pragma unroll
                for (int u = 0; u < PT_SKIP_PAIRS; ++u) w.skip[u] = -1;
                keys[lane] = TRI_KEY_NONE;
            } else {
                // flagged slots (marked by the previous bounce) always get a record, a hit or "nothing"; in scan
                // mode only hits are recorded and flagged here
                if (w.best_geom >= 0 || !a.mesh_scan)
                    a.mesh_hit[w.src] = make_float4(w.best_t, __int_as_float(w.best_geom), __int_as_float(w.best_tri), 0.0f);
                if (w.best_geom >= 0 && a.mesh_scan) atomicOr(&a.mesh_flags_in[w.src >> 6], 1ull << (w.src & 63u));
                w.have = false;
            }
        }
    };
#pragma unroll 1
    for (int k = 0; k < MQ_STEPS; ++k) {
#ifdef PT_STEP_REFILL
        if (k > 0 && rg.q_total != rg.q_head) mesh_refill(w, mq, mtab, rg, a);
#endif
        int leaf_l = -1, leaf_r = -1;
        if (w.have && w.node >= 0) {
            const BvhRec rec = bvh_fetch_top(a.scene.bvh_nodes, w.root, tops_lds, w.top, w.node, w.ray.oct);
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
            const uint64_t bb = ballot64(w.have && (w.node >= 0 || leaf_l >= 0 || leaf_r >= 0));
            const uint64_t wt = ballot64(w.have && w.node < 0 && leaf_l < 0 && leaf_r < 0);     // walk over, waiting for its queued triangles
            const uint64_t id = ballot64(!w.have);
            if (lane == 0 && bb) { atomicAdd(&a.ctl->keep[1], (uint32_t)__popcll((unsigned long long)bb)); atomicAdd(&a.ctl->keep[2], 1u); }
            if (lane == 0) { atomicAdd(&a.ctl->keep[8], (uint32_t)__popcll((unsigned long long)wt)); atomicAdd(&a.ctl->keep[9], (uint32_t)__popcll((unsigned long long)id)); atomicAdd(&a.ctl->keep[10], 1u); }
            atomicMax(&a.ctl->keep[14], (uint32_t)w.steps);
        }
#endif
        // queue this step's triangles: slot order = lane order (any order gives the same minimum)
        const int nl = leaf_l >= 0 ? (leaf_l >> 24) : 0, nr = leaf_r >= 0 ? (leaf_r >> 24) : 0;
        const int nt = nl + nr;
        if (ballot64(nt > 0)) {
            uint32_t pre = 0, tot = 0;
#pragma unroll
            for (int bit = 0; bit < NT_BITS; ++bit) {                // exclusive prefix of nt (<= 2 * LEAF_MAX) over the lanes
                const uint64_t bm = ballot64((nt >> bit) & 1);
                pre += rank_below(bm) << bit;
                tot += (uint32_t)__popcll((unsigned long long)bm) << bit;
            }
            const uint32_t pos = rg.t_total + pre;
#pragma unroll
            for (int j = 0; j < PT_LEAF_MAX; ++j) {                  // predicated stores, no per-lane loops
                if (j < nl) tq[(pos + (uint32_t)j) & (TQ_SLOTS - 1)] = (uint32_t)lane | ((uint32_t)((leaf_l & 0xffffff) + j) << 6);
                if (j < nr) tq[(pos + (uint32_t)(nl + j)) & (TQ_SLOTS - 1)] = (uint32_t)lane | ((uint32_t)((leaf_r & 0xffffff) + j) << 6);
            }
            if (nt > 0) w.ticket = pos + (uint32_t)nt;
            rg.t_total += tot;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef PT_TRI_FLUSH
            // test queued triangles before a whole pass has gathered: the walks' prune distance lags less (experiments)
            while (rg.t_total - rg.t_head >= PT_TRI_FLUSH) {
                const uint32_t cnt = min(64u, rg.t_total - rg.t_head);
                tri_pass(mq, rg.t_head, cnt, w, a); rg.t_head += cnt;
            }
#else
            while (rg.t_total - rg.t_head >= 64) { tri_pass(mq, rg.t_head, 64, w, a); rg.t_head += 64; }
#endif
        }
        // nobody is walking any more but triangles are still queued: test them now, their owners are waiting
        if (rg.t_total != rg.t_head && !ballot64(w.have && w.node >= 0)) {
            tri_pass(mq, rg.t_head, rg.t_total - rg.t_head, w, a);
            rg.t_head = rg.t_total;
        }
        if (multi) finish();
#ifdef PT_MESH_BREAK
        else if (!ballot64(w.have && w.node >= 0) && rg.t_total == rg.t_head) break;     // every walk of the block is over
#endif
    }
    if (!multi) finish();
}

__device__ __forceinline__ void mesh_drain(MeshWalker &w, float *mq, const float *tops, const float *mtab, MeshRings &rg, const BounceArgs &a, int leave) {
    for (;;) {
        const uint64_t busy = mesh_refill(w, mq, mtab, rg, a);
        if (!busy) return;
        if (rg.q_total == rg.q_head && (int)__popcll((unsigned long long)busy) < leave) return;
        mesh_steps(w, mq, tops, mtab, rg, a);
    }
}

// position of the r-th (0-based) set bit of w, r < popcount(w)
__device__ __forceinline__ uint32_t kth_set_bit(unsigned long long w, uint32_t r) {
    const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
    const uint32_t c = (uint32_t)__popc(lo);
    uint32_t pos = 0, x = lo;
    if (r >= c) { r -= c; x = hi; pos = 32; }
#pragma unroll
    for (int width = 16; width >= 1; width >>= 1) {
        const uint32_t cc = (uint32_t)__popc(x & ((1u << width) - 1u));
        if (r >= cc) { r -= cc; x >>= width; pos += (uint32_t)width; }
    }
    return pos;
}
constexpr uint32_t FLAG_GROUP = 8;            // tiles of a wave whose flag words are read together (flagged launches)

template <bool COMPACT>
__global__ __launch_bounds__(MESH_BLOCK, PT_MESH_WAVES) void k_mesh(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float *mq = lds_raw + (threadIdx.x >> 6) * MQ_WORDS;
    float *tops = lds_raw + MESH_WG_WAVES * MQ_WORDS;
    float *mtab = tops + BVH_TOP * BVH_TOP_STRIDE;
    if ((int)threadIdx.x < MESH_TAB && (int)threadIdx.x < a.scene.bvh_nmesh) {
        const int4 m = a.scene.bvh_meshes[threadIdx.x];
        const float *g = a.scene.geoms + (size_t)m.x * ptd::GEOM_WORDS + ptd::G_INV;
        float *e = mtab + threadIdx.x * MESH_TAB_WORDS;
        e[0] = __int_as_float(m.x); e[1] = __int_as_float(m.y); e[2] = __int_as_float(m.w); e[3] = 0.0f;
        for (int k = 0; k < 6; ++k) e[4 + k] = g[k];
    }
    for (int k = threadIdx.x; k < a.scene.bvh_top_n * 4; k += MESH_BLOCK)       // 64-B records -> 80-B slots, 16 B per thread per step
        reinterpret_cast<uint4 *>(tops + (k >> 2) * BVH_TOP_STRIDE)[k & 3] = reinterpret_cast<const uint4 *>(a.scene.bvh_top)[k];
    __syncthreads();
    uint32_t *mi = reinterpret_cast<uint32_t *>(mq);
    const int lane = threadIdx.x & 63;
    const uint32_t wid = run_id();
    const int iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    // bounce 0 of a batch: nlive[0] is written by that bounce's own kernel, so the count comes from the host
    const uint32_t n = (COMPACT && !a.gen_rays) ? a.ctl->nlive[a.depth] : a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    // (Measured with an instrumented build on C4: 42 of 64 lanes step on average at 64 spp per step, 27 at 16 -- a wave
    // only has ~100-400 walks per launch to refill its lanes with, and ends with a tail as long as its longest walk.
    // Giving the work to fewer, fuller waves was tried and is strictly slower -- 13.5 -> 13.0 / 11.3 / 7.7 Grays/s at
    // 1/2, 1/4, 1/8 of the waves: the walk is bound by the latency of its dependent record fetches, which only waves
    // in flight hide.)
    const uint32_t W = gridDim.x * MESH_WG_WAVES;
    const uint32_t R = range_tiles(n, W);
    const bool packed_in = COMPACT && a.dir_in.mem != nullptr;
    const uint32_t Wd = a.dir_in.W;                               // waves of the grid that packed the pool
    const uint32_t span_in = packed_in ? range_tiles(a.ctl->nlive[a.depth - 1], Wd) * TILE : 0;
    MeshRings rg{0, 0, 0, 0};
    MeshWalker w;
    w.have = false; w.src = 0; w.path = 0; w.ray = bvh_ray(ptd::mk(0, 0, 0), ptd::mk(0, 0, 1), ptd::mk(0, 0, 0), ptd::mk(1, 1, 1));
    w.mesh = 0; w.node = -1; w.steps = 0; w.ticket = 0; w.best_t = FLT_MAX; w.best_geom = -1; w.best_tri = -1;
    w.geom = 0; w.root = 0; w.top = 0;
#pragma unroll
    for (int u = 0; u < PT_SKIP_PAIRS; ++u) { w.skip[u] = -1; w.to[u] = -1; }
    // Tiles are dealt round-robin, not in runs: the pool keeps pixel order through every (stable) compaction, so the
    // rays that reach a mesh -- and the ones that leave its surface -- sit in neighbouring tiles; a run of them would
    // keep one wave walking long after the others are done (measured: waves alive 15 % of the launch on average).
    // mesh_scan = 0 (every bounce but the first): the previous bounce flagged the slots whose ray reaches a mesh's
    // root boxes; this kernel walks the PHYSICAL 64-slot tiles of the pool, skips the unflagged ones after one scalar
    // load -- no directory search, no ray loads, no root tests for the ~89 % of the paths that cannot hit a mesh -- and
    // loads only the flagged lanes' rays.
    const uint32_t phys_tiles = packed_in ? Wd * (span_in / TILE) : tiles;
    // Flagged launches.  A flagged tile holds a handful of candidates (7-20 % of its lanes), and finding them costs a
    // flag load plus a dependent round of ray loads; tile by tile, a wave waited on those about as long as it walked.
    // So: (1) the flags of FLAG_GROUP of the wave's tiles (dealt round-robin as in a scan, tile = round * W + wave) are
    // read by one load, lane j holding the word of round j, and the NEXT group's word is already in flight; (2) lane k
    // takes the k-th set bit of the group, so the rays of up to 64 candidates are fetched by ONE round of loads; (3) that
    // round is issued before a block of walk steps and its rays are appended to the ring after it -- the loads complete
    // under the block's own record fetches, and the ring is restocked before it runs dry.
    if (!a.mesh_scan) {
        const uint32_t rounds = (phys_tiles + W - 1) / W;
        const uint32_t groups = (rounds + FLAG_GROUP - 1) / FLAG_GROUP;
        auto load_group = [&](uint32_t g) -> unsigned long long {
            const uint32_t tile = (g * FLAG_GROUP + (uint32_t)lane) * W + wid;
            return ((uint32_t)lane < FLAG_GROUP && g < groups && tile < phys_tiles) ? a.mesh_flags_in[tile] : 0ull;
        };
        unsigned long long f_next = load_group(0), f_cur = 0;
        uint32_t g_next = 0, g_cur = 0, done = 0, total = 0;
        uint32_t p_take = 0, p_src = 0;                              // the batch in flight: candidates, slot, ray
        f3 p_ro = ptd::mk(0, 0, 0), p_rd = ptd::mk(0, 0, 1);
        for (;;) {
            if (p_take) {                                            // its loads were issued a block ago
                if ((uint32_t)lane < p_take) {
                    const uint32_t s = (rg.q_total + (uint32_t)lane) & (MQ_SLOTS - 1);
                    mi[0 * MQ_SLOTS + s] = p_src; mi[1 * MQ_SLOTS + s] = p_src;
                    mq[2 * MQ_SLOTS + s] = p_ro.x; mq[3 * MQ_SLOTS + s] = p_ro.y; mq[4 * MQ_SLOTS + s] = p_ro.z;
                    mq[5 * MQ_SLOTS + s] = p_rd.x; mq[6 * MQ_SLOTS + s] = p_rd.y; mq[7 * MQ_SLOTS + s] = p_rd.z;
                }
                rg.q_total += p_take;
#ifdef PT_MESH_STATS
                if (lane == 0) atomicAdd(&a.ctl->keep[0], p_take);
#endif
                p_take = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            if (rg.q_total - rg.q_head <= 64) {                      // room for a whole batch: pick and fetch the next one
                while (done == total && g_next < groups) {
                    f_cur = f_next; g_cur = g_next; ++g_next;
                    f_next = load_group(g_next);
                    total = 0; done = 0;
#pragma unroll
                    for (uint32_t j = 0; j < FLAG_GROUP; ++j)
                        total += (uint32_t)__popc(__builtin_amdgcn_readlane((int)(uint32_t)f_cur, j)) +
                                 (uint32_t)__popc(__builtin_amdgcn_readlane((int)(uint32_t)(f_cur >> 32), j));
                }
                if (done < total) {
                    const uint32_t i = done + (uint32_t)lane;
                    unsigned long long word = 0;
                    uint32_t base = 0, tj = 0, run = 0;
#pragma unroll
                    for (uint32_t j = 0; j < FLAG_GROUP; ++j) {      // the word that holds candidate i, and what came before it
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f_cur, j);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(f_cur >> 32), j);
                        if (i >= run) { word = ((unsigned long long)hi << 32) | lo; base = run; tj = j; }
                        run += (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
                    }
                    p_take = min(64u, total - done);
                    if ((uint32_t)lane < p_take) {
                        p_src = (((g_cur * FLAG_GROUP + tj) * W + wid) << 6) + kth_set_bit(word, i - base);
                        char *q = a.in.slot(p_src);
                        p_ro = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
                        p_rd = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
                    }
                    done += p_take;
                }
            }
            const uint64_t busy = mesh_refill(w, mq, mtab, rg, a);
            if (busy) mesh_steps(w, mq, tops, mtab, rg, a);
            else if (!p_take && done == total && g_next >= groups) break;
        }
        return;
    }
    const uint32_t rounds = R;
    // camera rays: whole 64-pixel tiles whose pixels cannot see a mesh are skipped after one mask bit (the tile's
    // position inside its sample is tracked incrementally: tile = r * W + wid, modulo the tiles of one sample)
    const bool masked = a.gen_rays && a.cam_mask != nullptr;
    const uint32_t tps = masked ? (uint32_t)a.map.tile_pixels / TILE : 1u;      // tiles per sample (tile_pixels % 64 == 0 when masked)
    uint32_t lt = masked ? wid % tps : 0u;
    const uint32_t lt_step = masked ? W % tps : 0u;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t tile = r * W + wid;
        uint32_t src = tile * TILE + lane;
        bool cand = false;
        f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1);
        {
            if (tile >= tiles) break;
            if (masked) {
                const uint32_t t = lt;
                lt += lt_step; if (lt >= tps) lt -= tps;
                if (!((a.cam_mask[t >> 6] >> (t & 63u)) & 1ull)) continue;
            }
            uint32_t cur = 0;
            if (packed_in) cur = find_range(a.dir_in.base(), a.dir_in.nr, tile * TILE);
            const uint32_t i = tile * TILE + lane;
            bool active = i < n;
            src = i;
            if (packed_in) src = resolve_src(a.dir_in, span_in, cur, i, active, a.ctl);
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
            // candidate: the ray reaches one of the two root boxes of some mesh
            cand = active && mesh_root_candidate(a.scene, ro, rd);
        }
#ifdef PT_MESH_SCAN_ONLY
        cand = cand && __float_as_uint(ro.x) == 0x7fc12345u;          // timing experiments: scan, load, walk nothing
#endif
        const uint64_t m = ballot64(cand);
        if (m) {
            if (cand) {
                const uint32_t s = (rg.q_total + rank_below(m)) & (MQ_SLOTS - 1);
                mi[0 * MQ_SLOTS + s] = src; mi[1 * MQ_SLOTS + s] = src;
                mq[2 * MQ_SLOTS + s] = ro.x; mq[3 * MQ_SLOTS + s] = ro.y; mq[4 * MQ_SLOTS + s] = ro.z;
                mq[5 * MQ_SLOTS + s] = rd.x; mq[6 * MQ_SLOTS + s] = rd.y; mq[7 * MQ_SLOTS + s] = rd.z;
            }
            rg.q_total += (uint32_t)__popcll((unsigned long long)m);
#ifdef PT_MESH_STATS
            if (lane == 0) atomicAdd(&a.ctl->keep[0], (uint32_t)__popcll((unsigned long long)m));
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // keep the ring below 64 waiting entries so the next tile always fits
            if (rg.q_total - rg.q_head >= 64 - (uint32_t)__popcll((unsigned long long)ballot64(w.have)))
                mesh_drain(w, mq, tops, mtab, rg, a, MQ_LEAVE);
        }
    }
    mesh_drain(w, mq, tops, mtab, rg, a, 0);
}

// First-bounce cache (INSTRUCTION.md:87-89): camera rays do not depend on the iteration (no
// jitter, pathtrace.cu:134), so computeIntersections of bounce 0 is evaluated once per pixel and
// camera and reused by every sample.
template <int MESH, bool SLDS>
__global__ __launch_bounds__(BLOCK, PT_MIN_WAVES) void k_cache_first(Isect cache, SceneDev sc, pt_camera cam,
                                                                      TileMap map) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const LdsCarve lc = carve_lds(lds_raw, sc, SLDS);
    const SceneAcc acc = stage_scene<SLDS>(lc.scene, sc);
    WaveQ q{lc.pw, 0, 0};
    const uint32_t n = (uint32_t)map.tile_pixels;
    const uint32_t tiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t j = tile * BLOCK + threadIdx.x;
        const bool active = j < n;
        f3 ro = ptd::mk(cam.position.x, cam.position.y, cam.position.z), rd = ptd::mk(0, 0, 1);
        if (active) camera_ray(cam, Lens{0, 0.0f, 0.0f}, 0, 0, local_to_pixel(map, (int)j), map.W, ro, rd);   // pinhole only (pt_init)
        MeshBest mb;
        cull_scene<MESH>(sc, acc, q, 0, lc.tri, active, ro, rd, mb, nullptr);
        drain_to(q, acc, q.total);
        if (active) {
            float t; f3 nrm; int mat, outside;
            tile_result(q, 0, acc, sc.tris, mb, t, nrm, mat, outside);
            cache.plane(0)[j] = t; cache.plane(1)[j] = nrm.x; cache.plane(2)[j] = nrm.y; cache.plane(3)[j] = nrm.z;
            cache.mat()[j] = mat | (outside ? 0 : (int)0x80000000u);
        }
    }
}

// shadeFakeMaterial (pathtrace.cu:224-266): one bounce, never spawns a ray
__global__ __launch_bounds__(BLOCK) void k_shade_fake(Pool p, Isect is, const float *mats_g, TileMap map,
                                                      int iter0, uint32_t n, float *fin, uint32_t stamp_arg, const Control *ctl) {
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
    put_final(fin, pid, c, batch_stamp(stamp_arg, ctl));
}

// finalGather (pathtrace.cu:269-278): image[pixelIndex] += colour, one add per
// pixel per iteration, samples added in iteration order
__global__ __launch_bounds__(BLOCK) void k_gather(float *image, const float *fin, uint32_t cap, TileMap map,
                                                  int count, Control *ctl, Persist *per, int depths,
                                                  uint32_t fake_rays, int partial_counts, int counters_only, uint32_t stamp_arg,
                                                  const uint32_t *iter_counts, uint32_t iter_grid, HostStats *host_stats) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (partial_counts) {            // the batch ran as ONE launch (k_iteration): its per-workgroup counts are added up here
        __shared__ uint32_t fold_lds[BLOCK / 32];
        if (blockIdx.x == 0)
            fold_iter_counts(iter_counts, iter_grid, depths, ctl, per, host_stats, (uint32_t)count, batch_stamp(stamp_arg, ctl), fold_lds);
    } else if (j == 0) {             // fold this batch's ray count into the persistent counters (batches of different lanes may
                                     // run side by side, hence atomics)
        unsigned long long r = fake_rays;
        for (int d = 0; d < depths; ++d) r += ctl->alive[d];
        atomicAdd(&per->rays, r);
        atomicAdd(&per->iterations, (unsigned long long)count);
        atomicAdd(&per->first_rays, (unsigned long long)(depths > 0 ? ctl->alive[0] : fake_rays));
    }
    if (counters_only || j >= (uint32_t)map.tile_pixels) return;
    const int pix = local_to_pixel(map, (int)j);
    float r = image[3 * pix + 0], g = image[3 * pix + 1], b = image[3 * pix + 2];
    // samples are added in iteration order (one add per pixel per iteration, as the reference does); the loads of
    // eight samples are issued together, the adds stay in order
    // an entry counts when it carries this batch's stamp; the others are paths that ended with colour 0 (put_final)
    const uint32_t stamp = batch_stamp(stamp_arg, ctl);
    const float4 *f4 = reinterpret_cast<const float4 *>(fin) + j;
    int s = 0;
    for (; s + 8 <= count; s += 8) {
        float4 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = f4[(size_t)(s + u) * map.tile_pixels];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (__float_as_uint(c[u].w) == stamp) { r += c[u].x; g += c[u].y; b += c[u].z; }
    }
    for (; s < count; ++s) {
        const float4 c = f4[(size_t)s * map.tile_pixels];
        if (__float_as_uint(c.w) == stamp) { r += c.x; g += c.y; b += c.z; }
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
        uint32_t lo = 0, hi = dir.nr - 1;
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
