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
//     sqrt3 * 2^-23 (|o| + |P|) of the ray's line, and the test below misplaces that line by less than
//     2^-20 (R + |c|) (R = the |origin|_1 bound of the non-wild rays: pt_h_scene.hpp, make_tri_bounds, with the error budget)
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

}  // namespace
