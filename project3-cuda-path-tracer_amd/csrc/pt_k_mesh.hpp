// pt_k_mesh.hpp -- k_mesh, the lane-dense mesh pre-pass of PT_MESH_BVH
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

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

#ifdef PT_MESH_STATS
// diagnostic build only (profiles/tools/build_variant.sh NAME WORK -DPT_MESH_STATS): where k_mesh's wave-steps go.
//   [0..8] walks by length (records visited: 1, 2-3, 4-7, ... 256+)   [9] walks  [10] sum of their lengths
//   [11..18] wave-steps by the number of lanes that walk (1-8, 9-16, ... 57-64)   [19] wave-steps where nobody walks
//   [20] loop iterations of the flagged form  [21] ... that appended a batch  [22] candidates appended
//   [23] flag words examined  [24] ... non-zero   [25] waves   [26] triangle passes  [27] their lanes
__device__ unsigned long long g_mesh_stats[32];
#define MESH_STAT(k, v) atomicAdd(&g_mesh_stats[k], (unsigned long long)(v))
#define MESH_STAT_WAVE(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_mesh_stats[k], (unsigned long long)(v)); } while (0)
#endif

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
#ifdef PT_MESH_STATS
    MESH_STAT_WAVE(26, 1); MESH_STAT_WAVE(27, min(count, 64u));
#endif
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

// A walk that is over (and whose queued triangles have been tested) folds its mesh's winner and publishes its result
// or moves on to the next mesh.
__device__ __forceinline__ void mesh_finish(MeshWalker &w, unsigned long long *keys, const float *mtab, const MeshRings &rg, const BounceArgs &a) {
    const int lane = threadIdx.x & 63;
    if (w.have && w.node < 0 && (int32_t)(rg.t_head - w.ticket) >= 0) {    // this mesh is done and fully tested
#ifdef PT_MESH_STATS
        { int b = 0; for (int v = w.steps; v > 1 && b < 8; v >>= 1) ++b; MESH_STAT(b, 1); MESH_STAT(9, 1); MESH_STAT(10, w.steps); }
#endif
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
    auto finish = [&]() { mesh_finish(w, keys, mtab, rg, a); };
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
            {
                const int nw = (int)__popcll((unsigned long long)ballot64(w.have && w.node >= 0));     // lanes that will walk in the NEXT step
                (void)nw;
                const int nb = (int)__popcll((unsigned long long)bb);
                if (nb) MESH_STAT_WAVE(11 + (nb - 1) / 8, 1); else MESH_STAT_WAVE(19, 1);
            }
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
#ifndef PT_FLAG_ADJ
#define PT_FLAG_ADJ 1
#endif
constexpr uint32_t FLAG_ADJ = PT_FLAG_ADJ;    // ... of which runs of this many are NEIGHBOURS in the pool (1: every tile of a group W tiles from the next)
static_assert(FLAG_GROUP % FLAG_ADJ == 0, "runs of adjacent tiles tile a group");
// tile j of group g of wave `wid` of W: the groups' runs are dealt round-robin over the waves
__device__ __forceinline__ uint32_t flag_tile(uint32_t g, uint32_t j, uint32_t W, uint32_t wid) {
    return ((g * (FLAG_GROUP / FLAG_ADJ) + j / FLAG_ADJ) * W + wid) * FLAG_ADJ + j % FLAG_ADJ;
}

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
    const uint32_t span_in = packed_in ? *a.dir_in.span() : 0;    // slots per range, as the producer wrote it down
#ifdef PT_MESH_STATS
    MESH_STAT_WAVE(25, 1);
#endif
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
        const uint32_t rounds = (phys_tiles + W * FLAG_ADJ - 1) / (W * FLAG_ADJ);     // runs per wave
        const uint32_t groups = (rounds * FLAG_ADJ + FLAG_GROUP - 1) / FLAG_GROUP;
        auto load_group = [&](uint32_t g) -> unsigned long long {
            const uint32_t tile = flag_tile(g, (uint32_t)lane & (FLAG_GROUP - 1), W, wid);
            return ((uint32_t)lane < FLAG_GROUP && g < groups && tile < phys_tiles) ? a.mesh_flags_in[tile] : 0ull;
        };
        unsigned long long f_next = load_group(0), f_cur = 0;
        uint32_t g_next = 0, g_cur = 0, done = 0, total = 0;
        uint32_t p_take = 0, p_src = 0;                              // the batch in flight: candidates, slot, ray
        f3 p_ro = ptd::mk(0, 0, 0), p_rd = ptd::mk(0, 0, 1);
        for (;;) {
#ifdef PT_MESH_STATS
            MESH_STAT_WAVE(20, 1);
            if (p_take) { MESH_STAT_WAVE(21, 1); MESH_STAT_WAVE(22, p_take); }
#endif
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
#ifdef PT_MESH_STATS
                    if ((uint32_t)lane < FLAG_GROUP) { MESH_STAT(23, 1); if (f_cur) MESH_STAT(24, 1); }
#endif
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
                        p_src = (flag_tile(g_cur, tj, W, wid) << 6) + kth_set_bit(word, i - base);
                        const SlotPtr q = a.in.slot(p_src);
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
                    const SlotPtr q = a.in.slot(src);
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

}  // namespace
