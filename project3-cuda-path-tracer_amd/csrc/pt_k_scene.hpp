// pt_k_scene.hpp -- scene staging and the first two stages of computeIntersections (pathtrace.cu:149-213): cull boxes -> candidate ring -> lane-dense exact passes
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

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
constexpr int LDS_CTL_WORDS = 32;    // [0] last-block flag, [8..11] traced counts, [16..31] the last workgroup's scan scratch (two steps x two sums x four waves)
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
constexpr int CULL_WORDS = 12;       // per geom (word 11 = type << 7 | geom << 9, the candidate ring's entry), scalar-loaded: centre.x half.x centre.y half.y centre.z half.z (of the padded world box) | type + (reject mode << 8) | the reject row:
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
//     float arithmetic could report a hit for (pt_h_scene.hpp: upload_cull / pt_cull.hpp, with the error bound).  All lanes
//     test their ray against it with one v_rcp per axis per RAY and, per primitive, nine fused multiply-adds, a v_max3,
//     a v_min3, a v_max and one compare (the box as centre and half extent, from wave-uniform scalar loads; round 5).  This test only decides which exact tests run,
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
// true unless the ray certainly misses the box centre +- half (scalar operands, pt_cull.hpp: centre_half).  NaN-safe
// towards "true".  Per axis the entry / exit parameters are  t_mid -+ half * |1/d|  with  t_mid = centre * (1/d) + n:
// three fused multiply-adds (|.| and the sign are source modifiers) where lo / hi planes took two plus a v_min and a
// v_max -- and a min / max issues in 4.2-4.4 SIMD cycles against 2.3 for an fma (profiles/r04/costs_r04.json): 38 issue
// cycles per (primitive, wave) instead of 57.  Two roundings per parameter instead of one; pt_cull.hpp (4) prices them.
__device__ __forceinline__ bool cull_box(const CullRay &c, float cx, float hx, float cy, float hy, float cz, float hz) {
    const float tmx = __builtin_fmaf(cx, c.ix, c.nx), tmy = __builtin_fmaf(cy, c.iy, c.ny), tmz = __builtin_fmaf(cz, c.iz, c.nz);
    const float tnx = __builtin_fmaf(-hx, __builtin_fabsf(c.ix), tmx), tfx = __builtin_fmaf(hx, __builtin_fabsf(c.ix), tmx);
    const float tny = __builtin_fmaf(-hy, __builtin_fabsf(c.iy), tmy), tfy = __builtin_fmaf(hy, __builtin_fabsf(c.iy), tmy);
    const float tnz = __builtin_fmaf(-hz, __builtin_fabsf(c.iz), tmz), tfz = __builtin_fmaf(hz, __builtin_fabsf(c.iz), tmz);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, 0.0f));
    const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
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
#ifndef PT_CULL_ROW
#define PT_CULL_ROW 1
#endif
__device__ __forceinline__ uint64_t cull_candidates(const CullRay &cr, uint64_t m_wild, f3 ro, f3 rd, float cx, float hx,
                                                    float cy, float hy, float cz, float hz, int tw, float m0, float m1,
                                                    float m2, float m3) {
    uint64_t keep = ballot64(cull_box(cr, cx, hx, cy, hy, cz, hz));
#if PT_CULL_ROW == 2
    // one straight-line form for every mode (the rows of modes 0..2 hold exact zeros off the diagonal, mode 3's row is all
    // zeros): no scalar dispatch per primitive, eight vector instructions more for a diagonal row
    {
        (void)tw;
        const float qk = (m0 * ro.x + m1 * ro.y) + (m2 * ro.z + m3);
        const float vk = (m0 * rd.x + m1 * rd.y) + m2 * rd.z;
        keep &= ~(ballot64(__builtin_fabsf(qk) > 0.5f) & ballot64(qk * vk > 0.0f));
    }
#elif PT_CULL_ROW == 1
    // Two forms instead of five.  A GENERAL row (mode 4) is evaluated in the reference's order.  Every other row has at most
    // one non-zero entry beside the translation (modes 0..2: m_kk; mode 3: none, all zeros), so in
    //   fma(m2, o_z, fma(m1, o_y, m0 o_x))
    // two of the three products are exact zeros and the value is fl(m_kk o_k) exactly -- what the reference's
    // m_kk o_k + 0 + 0 rounds to -- whichever k it is: no dispatch on the mode (round 5: the scalar pipe of k_bounce is as
    // full as its vector pipe; the dispatch was a dozen scalar instructions per primitive), four vector instructions more.
    {
        float qk, vk;
        if (tw & 0x400) {                                                 // mode 4
            qk = (m0 * ro.x + m1 * ro.y) + (m2 * ro.z + m3);
            vk = (m0 * rd.x + m1 * rd.y) + m2 * rd.z;                    // the reference adds m_k3 * 0.0f = +-0: same value when it matters
        } else {
            qk = __builtin_fmaf(m2, ro.z, __builtin_fmaf(m1, ro.y, m0 * ro.x)) + m3;
            vk = __builtin_fmaf(m2, rd.z, __builtin_fmaf(m1, rd.y, m0 * rd.x));
        }
        keep &= ~(ballot64(__builtin_fabsf(qk) > 0.5f) & ballot64(qk * vk > 0.0f));
    }
#else
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
#endif
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

}  // namespace
