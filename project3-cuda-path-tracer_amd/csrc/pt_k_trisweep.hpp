// pt_k_trisweep.hpp -- MESH_TILES: the loop over every triangle of a mesh (BASELINE C4 as stated), bounding spheres + lane-dense exact tests
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

// ---------------------------------------------------------------------------
// MESH_TILES: the loop over EVERY triangle of a mesh for every ray (completion spec 8.0 "Triangles"; BASELINE's
// "naive triangle loop (no BVH)"; INSTRUCTION.md:123-128).  No hierarchy, no grouping: each (ray, triangle) pair is
// visited.  Like the cubes and spheres (stages 1-3 above) a pair is visited in two stages:
//
//  1. BOUND.  pt_init computes per triangle a sphere (centre c, radius Rs) that contains every point a hit the spec
//     accepts can report: the spec's hit-point test (tri_point_ok) only counts a triangle whose reported point
//     P = fl(o + fl(d * tz)) lies inside the triangle's box widened by the mesh's pad, P lies within
//     sqrt3 * 2^-23 (|o| + |P|) of the ray's line, and a single-precision evaluation of that line misplaces it by less
//     than 2^-20 (R + |c|) (R = the |origin|_1 bound of the non-wild rays: pt_h_scene.hpp, make_tri_bounds, with the error
//     budget) -- so a ray whose line passes the centre at more than Rs cannot be accepted for this triangle, whatever
//     glm::intersectRayTriangle's float arithmetic returns for it.
//     Round 6: the test runs on the MATRIX pipe.  In the mesh's own frame (centre g, scaled by 1 / Rm so that every sphere
//     lies in the unit ball) the squared distance of the line (unit direction d, moment m = o x d) to c is
//         |c x d - m|^2 = |c|^2 - sum_ij c_i c_j d_i d_j - 2 c . (d x m) + |m|^2,
//     so  v = |c x d - m|^2 - Rs^2  is ONE bilinear form of a per-triangle vector
//         [-cx^2 -cy^2 -cz^2 -2cxcy -2cxcz -2cycz | cx cy cz | K = |c|^2 - Rs^2 | 1]   (pt_init: make_tri_records) and a per-ray vector
//         [ dx^2  dy^2  dz^2   dxdy   dxdz   dydz | -2wx -2wy -2wz | 1 | M = |m|^2 - E]   (w = d x m; ray_slots below).
//     Every term is carried as a binary16 pair (hi, lo) and three of its four cross products (hi hi, hi lo, lo hi): 6 x 3 +
//     3 x 3 + 2 + 2 = 31 of the 32 K-slots of v_mfma_f32_16x16x32_f16, products exact, accumulated in binary32.  One
//     MFMA = 16 triangles x 16 rays; four of them cover the wave's 64 rays; what is left for the vector pipe is the OR of
//     the results' sign bits: a pair is a candidate when v < 0.  Measured (profiles/microbench/tri_reject_mfma.hip, the same
//     mesh size): 1.42e13 pairs/s against 4.4e12 for round 5's form (six fma, a three-term dot, a compare per pair).
//     ERROR BUDGET of the form itself (its inputs' rounding is inside Rs already -- make_tri_bounds reserves 2^-17 of the
//     reach for a single-precision line, this form needs 2^-21): for rays whose line comes near the unit ball
//     (|m|^2 <= 1.21; the others are classified FAR and never candidates: spheres lie in the unit ball) the terms' absolute
//     values sum to at most (sum |c_i||d_i|)^2 + 2 |c||w| + |K| + M <= 1 + 2.2 + 1 + 1.21 = 5.41; dropping the lo lo products
//     and rounding the operands to 22 bits loses at most 3 x 2^-22 of that, the binary16 subnormal step adds 62 x 2^-25, and
//     31 binary32 additions in any order lose at most 31 x 2^-24 x 5.41: together < 1.5e-5.  E = 4e-5 is subtracted on the
//     ray side, so v_computed < 0 whenever the exact v <= 0 (tests/test_tri_bounds_cpu.py models the form; the whole C4
//     frame is compared with the oracle's, tests/test_gpu_mesh.py).  Rays the bound was not derived for (non-finite, huge,
//     odd direction magnitudes: `wild`) get a ray vector that makes every real triangle a candidate; non-finite triangles a
//     triangle vector that makes them everybody's candidate (the exact test never accepts them); padding records one that
//     nobody reaches.
//  2. EXACT.  Candidates (lane, triangle) queue in a per-wave LDS ring; whenever 64 wait, lane k runs
//     glm::intersectRayTriangle (operation for operation, ptd::ray_triangle) + the hit-point test for candidate k
//     -- the ray from the wave's LDS copy, the triangle record gathered from global memory -- and folds
//     (bits(bary.z) << 32) | triangle index into the owner's key with an LDS 64-bit min: the smallest bary.z, the
//     lowest index on ties, i.e. the loop's strict `best > tz` scan in index order.
// ---------------------------------------------------------------------------
constexpr unsigned long long TRI_KEY_NONE = (0x7f7fffffull << 32) | 0xffffffffull;   // bary.z = FLT_MAX, no triangle
#ifndef PT_SWEEP_AHEAD
#define PT_SWEEP_AHEAD 4                       // groups of 16 triangle records fetched ahead of the MFMAs that use them
#endif
constexpr int TRQ_SLOTS = 128;                 // triangle candidates waiting per wave (a step adds <= 64 while < 64 wait)
constexpr int TRQ_WORDS = TRQ_SLOTS + 2 * 64 + 2 * 64 * 4;   // ring + the 64 per-lane best keys (u64) + 2 KiB of scratch (the rays' slots, 32 rays at a time): 3 KiB per wave
typedef _Float16 pt_half8 __attribute__((ext_vector_type(8)));
typedef float pt_float4v __attribute__((ext_vector_type(4)));
constexpr float TRI_FORM_E = 4.0e-5f;          // the bilinear form's error budget (above), subtracted on the ray side
constexpr float TRI_FAR_M2 = 1.21f;            // |m|^2 beyond which a line cannot touch a sphere of the unit ball

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

// binary16 pair of a value: hi = nearest, lo = nearest of the remainder
__device__ __forceinline__ void half_pair(float v, _Float16 &hi, _Float16 &lo) { hi = (_Float16)v; lo = (_Float16)(v - (float)hi); }

// the ray's 32 K-slots (slot pairing with make_tri_records: a term's three slots are (hi, lo, hi) here against (hi, hi, lo) there)
__device__ __forceinline__ void ray_slots(f3 d, f3 m, float M, bool far, bool wild, bool active, pt_half8 out[4]) {
    const float wx = d.y * m.z - d.z * m.y, wy = d.z * m.x - d.x * m.z, wz = d.x * m.y - d.y * m.x;
    const float v[9] = {d.x * d.x, d.y * d.y, d.z * d.z, d.x * d.y, d.x * d.z, d.y * d.z, -2.0f * wx, -2.0f * wy, -2.0f * wz};
    _Float16 b[32];
#pragma unroll
    for (int t = 0; t < 9; ++t) { _Float16 h, l; half_pair(v[t], h, l); b[3 * t] = h; b[3 * t + 1] = l; b[3 * t + 2] = h; }
    b[27] = (_Float16)1.0f; b[28] = (_Float16)1.0f;                    // x K (hi, lo)
    { _Float16 h, l; half_pair(M - TRI_FORM_E, h, l); b[29] = h; b[30] = l; }
    b[31] = (_Float16)0.0f;
    // far lines and idle lanes: v = K + 1000 > 0 for every record; wild rays: v = K - 1000 < 0 for every real triangle
    // (padding records carry K = 30000).  Selected, not computed: a wild ray's own numbers may be anything.
    const bool plain = active && !far && !wild;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const _Float16 c = (k == 29) ? ((wild && active) ? (_Float16)-1000.0f : (_Float16)1000.0f) : ((k == 27 || k == 28) ? (_Float16)1.0f : (_Float16)0.0f);
        b[k] = plain ? b[k] : c;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) out[q][k] = b[8 * q + k];
}

// nearest accepted triangle of the mesh [first, first + count) for this lane's ray: best = bary.z, best_i = index
// (gx, gy, gz, ir) = the mesh's frame {g, 1 / Rm} (geom record words G_INV + 7 .. + 10)
__device__ __forceinline__ void mesh_sweep(const SceneDev &sc, const WaveQ &q, int par, float *trq, int first, int count, int boff,
                                           float gx, float gy, float gz, float ir, f3 ro, f3 rd, uint64_t m_act, uint64_t m_wild, float &best, int &best_i) {
    const int lane = threadIdx.x & 63;
    uint32_t *ring = reinterpret_cast<uint32_t *>(trq);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(trq + TRQ_SLOTS);
    const float *ry0 = q.rays(par);
    keys[lane] = TRI_KEY_NONE;
    uint32_t head = 0, total = 0;
    // the ray's line in the mesh's frame: unit direction (v_rsq: 2^-22 of |d|^2, inside the budget), moment about the centre
    const float sc1 = __builtin_amdgcn_rsqf((rd.x * rd.x + rd.y * rd.y) + rd.z * rd.z);
    const f3 d = ptd::mk(rd.x * sc1, rd.y * sc1, rd.z * sc1);
    const f3 o = ptd::mk((ro.x - gx) * ir, (ro.y - gy) * ir, (ro.z - gz) * ir);
    const f3 m = ptd::mk(__builtin_fmaf(o.y, d.z, -(o.z * d.y)), __builtin_fmaf(o.z, d.x, -(o.x * d.z)), __builtin_fmaf(o.x, d.y, -(o.y * d.x)));
    const float M = __builtin_fmaf(m.z, m.z, __builtin_fmaf(m.y, m.y, m.x * m.x));
    const bool active = (m_act >> lane) & 1ull, wild = (m_wild >> lane) & 1ull;
    pt_half8 mine[4];
    ray_slots(d, m, M, !(M <= TRI_FAR_M2), wild, active, mine);          // (NaN: not <=, far -- unless wild, which such a ray is)
    // B operands: ray group gI = the wave's rays 16 gI .. 16 gI + 15; lane l holds column l & 15, K-slots 8 (l >> 4) .. + 7 --
    // through the wave's LDS scratch, 32 rays (2 KiB) at a time
    pt_half8 *scratch = reinterpret_cast<pt_half8 *>(trq + TRQ_SLOTS + 2 * 64);           // [32 rays][4 blocks of 8 slots]
    pt_half8 bf[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) scratch[(lane & 31) * 4 + qd] = mine[qd];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bf[2 * h] = scratch[(lane & 15) * 4 + (lane >> 4)];
        bf[2 * h + 1] = scratch[(16 + (lane & 15)) * 4 + (lane >> 4)];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // A operands: lane l fetches row (triangle) l & 15, K-slots 8 (l >> 4) .. + 7 of the group's sixteen 64-byte records: 16 B
    // per lane, 1 KiB per group, straight from the L2 (every wave of the device streams the same records); PT_SWEEP_AHEAD in flight
    const pt_half8 *__restrict__ rec = reinterpret_cast<const pt_half8 *>(sc.tri_rec) + (size_t)boff * 4 + (size_t)(lane & 15) * 4 + (lane >> 4);
    const int ngroups = (count + 15) >> 4;                            // (records are padded to a multiple of 64)
    pt_half8 af[PT_SWEEP_AHEAD];
#pragma unroll
    for (int u = 0; u < PT_SWEEP_AHEAD; ++u) af[u] = rec[(size_t)min(u, max(ngroups - 1, 0)) * 64];
    for (int g = 0; g < ngroups; g += PT_SWEEP_AHEAD) {
        pt_half8 nx[PT_SWEEP_AHEAD];
#pragma unroll
        for (int u = 0; u < PT_SWEEP_AHEAD; ++u) nx[u] = rec[(size_t)min(g + PT_SWEEP_AHEAD + u, ngroups - 1) * 64];
#pragma unroll
        for (int u = 0; u < PT_SWEEP_AHEAD; ++u) {
            if (g + u >= ngroups) break;                              // (wave-uniform)
            pt_float4v acc[4];
            uint32_t any = 0;
#pragma unroll
            for (int gI = 0; gI < 4; ++gI) {
                const pt_float4v z = {0.0f, 0.0f, 0.0f, 0.0f};
                acc[gI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], bf[gI], z, 0, 0, 0);
                any |= (__float_as_uint(acc[gI][0]) | __float_as_uint(acc[gI][1])) | (__float_as_uint(acc[gI][2]) | __float_as_uint(acc[gI][3]));
            }
            if (__builtin_expect(ballot64((int)any < 0) != 0, 0)) {   // rare: ~1 group in 50 on BASELINE C4's mesh
                // which pairs: the result's lane l holds column (ray) 16 gI + (l & 15), rows (triangles) 4 (l >> 4) + r
#pragma unroll
                for (int gI = 0; gI < 4; ++gI)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 16 * (g + u) + 4 * (lane >> 4) + r;
                        const uint64_t mm = ballot64((__float_as_uint(acc[gI][r]) >> 31) != 0 && k < count);
                        if (mm) {
                            if (lane_of(mm)) ring[(total + rank_below(mm)) & (TRQ_SLOTS - 1)] = (uint32_t)(16 * gI + (lane & 15)) | ((uint32_t)(first + k) << 6);
                            total += (uint32_t)__popcll((unsigned long long)mm);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            if (total - head >= 64) { tri_cand_pass(ry0, ring, keys, sc.tris, head, 64); head += 64; }
                        }
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < PT_SWEEP_AHEAD; ++u) af[u] = nx[u];
    }
    while (total != head) {
        const uint32_t cnt = min(64u, total - head);
        tri_cand_pass(ry0, ring, keys, sc.tris, head, cnt);
        head += cnt;
    }
    const unsigned long long key = keys[lane];
    if ((uint32_t)key != 0xffffffffu) { best = __uint_as_float((uint32_t)(key >> 32)); best_i = (int)(uint32_t)key; }
}

}  // namespace
