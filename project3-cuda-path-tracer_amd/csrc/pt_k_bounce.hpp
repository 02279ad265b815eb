// pt_k_bounce.hpp -- the fused bounce kernel k_bounce (intersect + shade / scatter + stable compaction) and k_iteration (all bounces of a small batch in one launch)
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

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
    bool epi_direct = false;    // k_iteration doing its own finalGather path by path (BounceArgs::epi_direct)
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
            const SlotPtr p = (c.kargs ? karg_pool(offsetof(BounceArgs, in)) : in).slot(src);
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
            if (c.epi_direct) {
                // finalGather for this path, here (pathtrace.cu:380-392 at one sample per pixel): image[pixel] += colour, the
                // same single addition per channel; colour 0 leaves the sum as it is (x + 0 = x exactly, sums are never -0)
                // and as the host has it.  NaN compares false: added.
                if (!(ps.c.x == 0.0f && ps.c.y == 0.0f && ps.c.z == 0.0f)) {
                    float *img = karg_field<float *>(offsetof(BounceArgs, epi_image)) + 3 * (size_t)tr.pixel;
                    const float r = img[0] + ps.c.x, g = img[1] + ps.c.y, b = img[2] + ps.c.z;
                    img[0] = r; img[1] = g; img[2] = b;
                    float *host = karg_field<float *>(offsetof(BounceArgs, epi_host));
                    if (host) { host += 3 * (size_t)tr.pixel; host[0] = r; host[1] = g; host[2] = b; }
                }
            } else {
                put_final((c.kargs || c.kmisc) ? karg_field<float *>(offsetof(BounceArgs, fin)) : a.fin, tr.pid, ps.c, c.stamp);
            }
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
        const SlotPtr p = (c.kargs ? karg_pool(offsetof(BounceArgs, out)) : out).slot(dst);
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
                                          uint32_t key_stride = 0, bool aligned = false) {
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
        } else if (aligned) {
            // tile `tile` of the range-aligned sequence (RangeDir): whole rows of ONE physical tile of the source pool
            have = tile < tiles;
            active = false; i = 0; src = 0;
            if (have) {
                const uint32_t *tb = a.dir_in.tbase();
                while (tile >= tb[cur + 1]) ++cur;               // wave-uniform; empty ranges are stepped over
                const uint32_t t64 = (tile - tb[cur]) * TILE + (uint32_t)lane;
                active = t64 < a.dir_in.count()[cur];
                src = cur * span_in + t64;
                i = a.dir_in.base()[cur] + t64;
            }
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
                unsigned long long *fl = c.kargs ? karg_field<unsigned long long *>(offsetof(BounceArgs, mesh_flags_in)) : a.mesh_flags_in;
                const unsigned long long word = fl[src >> 6];
                if ((word >> (src & 63u)) & 1ull) pre_hit = (c.kargs ? karg_field<float4 *>(offsetof(BounceArgs, mesh_hit)) : a.mesh_hit) + src;
                // ... and this wave is the word's last reader (k_mesh read it a launch ago; a tile of this kernel is one
                // PHYSICAL tile of the pool -- dense, or aligned to the ranges -- so nobody else looks at this word): it puts
                // it back to zero, ready to be the output flags of the next bounce.  Round 5 cleared the whole array with a
                // hipMemsetAsync per bounce (eight extra launches per batch: profiles/r05/rocprof_r05_c4_bvh_summary.txt).
                if (word != 0ull && (uint32_t)lane == (uint32_t)__builtin_ctzll((unsigned long long)ballot64(tr.active))) fl[src >> 6] = 0ull;
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
    const bool packed_in = COMPACT && !GEN && a.dir_in.mem != nullptr;
    // an unsorted packed pool is read in tiles aligned to its ranges (pt_types.hpp: RangeDir): `tiles` counts those
    constexpr bool ALIGNED = PT_ALIGNED_TILES && MODE == MODE_FUSED && COMPACT && !GEN && !SORT;
    const bool aligned = ALIGNED && packed_in;
    const uint32_t tiles = aligned ? a.dir_in.tbase()[a.dir_in.nr] : (n + TILE - 1) / TILE;
    const uint32_t R = (tiles + W - 1) / W;                      // tiles per run (contiguous)
    const uint32_t span_in = packed_in ? *a.dir_in.span() : 0;   // slots per range, as the producer wrote it down
    uint32_t traced = 0;
    if (GEN && blockIdx.x == 0 && threadIdx.x == 0) a.ctl->nlive[0] = a.pool_n;   // k_raygen's job otherwise
    for (uint32_t j = 0; j < runs_per_wave; ++j) {
        const uint32_t wid = j * Wp + wid0;
        uint32_t packed = 0;                                     // survivors of this run written so far (wave-uniform; SORT: lane k counts key k)
        uint32_t cur = 0;                                        // source range of the run's current position
        if (packed_in && wid * R < tiles)
            cur = aligned ? find_range(a.dir_in.tbase(), a.dir_in.nr, wid * R) : find_range(a.dir_in.base(), a.dir_in.nr, wid * R * TILE);
        STAMP(2);
        // the run's R consecutive 64-path tiles; no workgroup barrier inside the loop
        run_tiles<MODE, COMPACT, MESH, GEN, SORT>(a, c, q, a.in, a.out, a.depth, wid * R, R, tiles, n, packed_in, span_in,
                                                  cur, wid * R * TILE, false, WgSpans{}, packed, traced, W * R * TILE, aligned);
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
            scan_range_counts(a.dir_out, &a.ctl->nlive[a.depth + 1], sctl + 16, R * TILE, PT_ALIGNED_TILES && !SORT);
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
    c.epi_direct = a.epi_direct != 0;
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
    if (epi_image && !c.epi_direct) {                               // (epi_direct: every ending path has done it for its pixel)
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

}  // namespace
