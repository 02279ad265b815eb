// pt_k_intersect.hpp -- computeIntersections for one tile (cull_scene, tile_result) and the kernels that only intersect: k_intersect, k_cull0_mask, k_cache_first
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

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
    // The records come through wave-uniform scalar loads (s_load_dwordx8 + x4), walked by POINTER: the geom's number is
    // only needed on the rare paths (bounce-0 masks, meshes) and the candidate ring's entry -- type << 7 | geom << 9 -- is
    // word 11 of the record (round 5: k_bounce's scalar pipe is as full as its vector pipe; a loop counter and three
    // scalar instructions per queued primitive to build that entry were among the 43 scalar instructions per primitive).
    // Round 2 requested the next primitive's record before using this one's; by round 3 the eleven scalar registers that
    // keeps alive across the loop cost more than the latency (profiles/r03/variants_cull_prefetch.log).
    cfloat *cc = as_const(sc.cull);
    cfloat *const cc_end = cc + sc.ngeoms * CULL_WORDS;
    for (; cc != cc_end; cc += CULL_WORDS) {
        float cb[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) cb[k] = cc[k];
        uint32_t ent = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cb[11]));     // type << 7 | geom << 9 (wave-uniform)
        // (pinned here: left to itself the compiler sinks this one load into the branch that queues the candidates, where its
        // latency is exposed once per primitive; with the record's other words it costs nothing)
        asm volatile("" : "+s"(ent));
        const int g = (int)(ent >> 9);
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
                             sc.bvh_guard, ro, rd, best, best_i, cr.wild);
            } else {
                // every triangle of the mesh, for every ray (the completion spec's loop, 8.0): mesh_sweep
                const int first = __float_as_int(rec[2]);
                const int count = __float_as_int(rec[3]);
                const int boff = __float_as_int(rec[ptd::G_INV + 6]);
                mesh_sweep(sc, q, par, tri_lds, first, count, boff, rec[ptd::G_INV + 7], rec[ptd::G_INV + 8], rec[ptd::G_INV + 9], rec[ptd::G_INV + 10],
                           ro, rd, m_act, m_wild, best, best_i);
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
                q.ring()[s] = tag | ent;
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
    const uint32_t span = packed ? *dir_in.span() : 0;            // slots per range, as the producer wrote it down
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
            const SlotPtr p = in.slot(src);
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

}  // namespace
