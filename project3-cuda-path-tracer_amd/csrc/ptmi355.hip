// ptmi355.hip -- libptmi355.so: kernels + C-ABI (include/ptmi355.h).
//
// MI355X-native replacement for the hot path of the reference's
// src/pathtrace.cu (pathtraceInit / pathtrace / pathtraceFree and its five
// kernels).  Design:
//   * path state lives in SoA planes (ox..oz, dx..dz, cr..cb, pid) of a pool
//     that holds `batch` iterations of the tile's pixels; two pools ping-pong;
//   * one fused kernel per bounce: intersect (scene records broadcast from
//     LDS) -> shade/scatter -> stable compaction -> write survivors;
//   * compaction is tile-local and wait-free: a 256-path tile packs its
//     survivors (wave64 ballot + popcount rank, 4 wave counts through LDS) to
//     the front of its own 256-slot span and publishes its count; the last
//     workgroup to finish the launch (one agent-scope atomic per workgroup)
//     scans the tile counts into tile bases; the next bounce reads logical
//     path i through those bases (64-entry window in LDS + binary search), so
//     the logical order is exactly the stable partition's while no workgroup
//     ever waits on another (round-1 look-back version: 81 % of wave time
//     parked, profiles/r01a);
//   * the live count stays on the device: the next bounce reads it from HBM,
//     no host round trip inside an iteration;
//   * terminated paths drop their final colour into final[sample][pixel];
//     one gather kernel adds the samples into the float3 accumulation buffer
//     in iteration order (bit-identical to sequential iterations).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA contraction:
// parity with the reference arithmetic is bit-exact, tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ptmi355.h"
#include "pt_device.hpp"

static_assert(sizeof(pt_vec3) == 12 && sizeof(pt_mat4) == 64 && sizeof(pt_ray) == 24, "ABI");
static_assert(sizeof(pt_geom) == 236 && offsetof(pt_geom, transform) == 44 &&
              offsetof(pt_geom, inverseTransform) == 108 && offsetof(pt_geom, invTranspose) == 172, "Geom ABI");
static_assert(sizeof(pt_material) == 44 && offsetof(pt_material, hasReflective) == 28 &&
              offsetof(pt_material, emittance) == 40, "Material ABI");
static_assert(sizeof(pt_camera) == 84 && offsetof(pt_camera, view) == 32 &&
              offsetof(pt_camera, pixelLength) == 76, "Camera ABI");
static_assert(sizeof(pt_path_segment) == 44 && offsetof(pt_path_segment, pixelIndex) == 36, "PathSegment ABI");
static_assert(sizeof(pt_shadeable_intersection) == 20 && offsetof(pt_shadeable_intersection, materialId) == 16,
              "ShadeableIntersection ABI");
static_assert(sizeof(pt_triangle) == 36, "triangle ABI");

using ptd::f3;

namespace {

#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4                     // waves per SIMD the bounce kernels are register-budgeted for
#endif
#ifndef PT_GEOM_LDS
#define PT_GEOM_LDS 0                      // 1: broadcast geom records from LDS, 0: scalar loads (SGPRs)
#endif
#ifndef PT_QUEUE
#define PT_QUEUE 1                         // evaluate the world-distance tails lane-dense from a per-wave LDS queue
#endif
constexpr int BLOCK = 256;                 // 4 waves of 64
constexpr int WAVES = BLOCK / 64;
constexpr int TILE = 64;                   // paths per tile = one wave64
constexpr int TRI_TILE = 512;              // triangles staged in LDS per pass (24 KiB)
constexpr int TRI_WORDS = 12;              // v0 e1 e2 + 3 pad: three 16-B words per triangle
constexpr int MAX_DEPTH = 64;
constexpr uint32_t DEAD_PID = 0xffffffffu;

// ---------------------------------------------------------------------------
// device-side parameter blocks (few pointers: every extra pointer pair costs
// 2 SGPRs per wave for the whole kernel)
// ---------------------------------------------------------------------------
// element `i` of a wave-uniform plane pointer through a 32-bit byte offset: the address is
// SGPR base + zero-extended VGPR offset (one global_load/store, no 64-bit VALU address math).
// pt_init guarantees cap * 4 < 2^32.
template <typename T>
__device__ __forceinline__ T &at(T *plane, uint32_t i) {
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(plane) + (i << 2));
}

// Path pool: SoA *per 64-path tile* -- tile T holds its ten planes (ox oy oz dx dy dz cr cg cb pid)
// as ten consecutive 256-B rows, 2560 B per tile.  A wave reads/writes whole rows (coalesced),
// and all ten fields of slot s sit at one per-lane address plus the immediates 0, 256, ... 2304:
// one address computation per path instead of ten, and no plane base pointers in SGPRs.
struct Pool {
    float *base;
    uint32_t cap;        // slots, a multiple of 64
    __device__ __forceinline__ char *slot(uint32_t s) const {
        return reinterpret_cast<char *>(base) + (size_t)(s >> 6) * 2560u + ((s & 63u) << 2);
    }
    __device__ __forceinline__ float &f(uint32_t s, int k) const { return *reinterpret_cast<float *>(slot(s) + k * 256); }
    __device__ __forceinline__ uint32_t &pid(uint32_t s) const { return *reinterpret_cast<uint32_t *>(slot(s) + 9 * 256); }
};
__device__ __forceinline__ float &pf(char *slot, int k) { return *reinterpret_cast<float *>(slot + k * 256); }
__device__ __forceinline__ uint32_t &ppid(char *slot) { return *reinterpret_cast<uint32_t *>(slot + 9 * 256); }

struct Isect {           // ShadeableIntersection planes t nx ny nz mat (unfused / sort / fake-shader modes)
    float *base;         // mat: bit 31 carries the winning test's !outside
    uint32_t cap;
    __device__ __forceinline__ float *plane(int k) const { return base + (size_t)k * cap; }
    __device__ __forceinline__ int *mat() const { return reinterpret_cast<int *>(base + (size_t)4 * cap); }
};

struct TileMap {         // local pixel index -> global pixelIndex (x + y*W)
    int W, H;
    int tile_index, tile_count, strip_rows;
    int tile_pixels;     // pixels owned by this tile
    uint32_t div_magic;  // pid / tile_pixels without an integer divide (see sample_of)
    uint32_t div_shift;
};

// pid / tile_pixels for every 32-bit pid: round-up magic number, branch-free form
// (q = mulhi(magic, n); ((n - q) >> 1) + q) >> shift), magic/shift chosen by make_div_magic().
__device__ __forceinline__ uint32_t sample_of(const TileMap &m, uint32_t pid) {
    if (m.tile_pixels == 1) return pid;                  // the branch-free form needs a divisor >= 2
    const uint32_t q = __umulhi(m.div_magic, pid);
    return (((pid - q) >> 1) + q) >> m.div_shift;
}

struct Control {         // zeroed by one hipMemsetAsync per batch (2 KiB)
    unsigned long long stamp[16];   // -DPT_STAMPS: s_memrealtime at the phases of wave 0 / the last workgroup
    uint32_t nlive[MAX_DEPTH + 1];  // nlive[d] = paths entering bounce d (compaction on)
    uint32_t alive[MAX_DEPTH + 1];  // paths actually traced at bounce d
    uint32_t done[MAX_DEPTH];       // election buckets that finished bounce d (last-one-out election, top level)
    uint32_t done_sort[MAX_DEPTH];  // same for the material-sort histogram of bounce d
    uint32_t error;
    uint32_t scan_ticks[MAX_DEPTH]; // 100 MHz ticks the last workgroup spent scanning (diagnostic)
    uint32_t pad[512 - 32 - 2 * (MAX_DEPTH + 1) - 3 * MAX_DEPTH - 1];
    // first-level election counters: 32 buckets per bounce, one 64-B line apart
    uint32_t bucket[MAX_DEPTH][2][32 * 16];       // [bounce][bounce kernel | sort histogram][bucket * 16]
};
constexpr int ELECT_BUCKETS = 32;
static_assert(sizeof(Control) == 2048 + 2 * MAX_DEPTH * 32 * 16 * 4, "Control is one memset block");
static_assert(sizeof(Control) % 16 == 0, "memset block is a multiple of 16 B");

// Last-workgroup-out election without hammering one address: a same-address atomic costs ~12 ns at
// the memory side, so 1-2 thousand workgroups finishing together would serialise for tens of
// microseconds.  Workgroup b adds to bucket b % 32 (own cache line); the last arriver of a bucket
// adds to the top counter; the last of those is the last workgroup of the launch.  Call from ONE
// thread, after the workgroup's stores have drained and its barrier.
__device__ __forceinline__ bool elect_last(uint32_t *buckets /* [32*16] */, uint32_t *top) {
    const uint32_t G = gridDim.x;
    const uint32_t k = blockIdx.x % ELECT_BUCKETS;
    const uint32_t members = (G - k + ELECT_BUCKETS - 1) / ELECT_BUCKETS;       // workgroups with b % 32 == k
    const uint32_t used = G < ELECT_BUCKETS ? G : ELECT_BUCKETS;                 // buckets that have members
    const uint32_t old = __hip_atomic_fetch_add(&buckets[k * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old != members - 1) return false;
    const uint32_t t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t == used - 1;
}

// Range directory of one bounce's OUTPUT pool.  Wave w of the persistent grid owns the
// contiguous run of `R` logical tiles [wR, (w+1)R) (R = ceil(tiles / W)) and packs every
// survivor of that run, in order, to the front of the run's own span of R*64 slots; count[w]
// is how many it packed, base[] the exclusive scan of count[] (W+1 entries).  Logical path i
// of the output therefore lives in slot r*R*64 + (i - base[r]) for the range r with
// base[r] <= i < base[r+1] -- the stable partition's order, with no cross-wave communication
// inside the launch and only W (<= 8192) words to scan at its end.
struct RangeDir {
    uint32_t *mem;       // count[Wp] | base[Wp+4]  (Wp = W rounded up to 4); nullptr = dense pool
    uint32_t W;          // waves in the persistent grid = ranges
    __device__ __forceinline__ uint32_t *count() const { return mem; }
    __device__ __forceinline__ uint32_t *base() const { return mem + ((W + 3u) & ~3u); }
};

// Which run of tiles a wave owns: wave j of workgroup b takes run j*G + b, so the first G runs go
// to G different workgroups.  When a bounce has fewer runs than waves (small pools, late
// bounces) the busy waves are then spread over every CU instead of filling the first workgroups
// the dispatcher happens to co-locate (measured at 800x800, 1 spp, bounce 7: 42 -> 2x shorter).
__device__ __forceinline__ uint32_t run_id() {
    return __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
}

// tiles per range for a pool of n paths split over W waves
__host__ __device__ __forceinline__ uint32_t range_tiles(uint32_t n, uint32_t W) {
    const uint32_t tiles = (n + 63u) / 64u;
    return (tiles + W - 1) / W;
}

struct Persist {         // survives the per-batch memset
    unsigned long long rays;        // sum over bounces of paths traced since pt_init
    unsigned long long iterations;
    unsigned long long first_rays;  // paths traced at bounce 0 (rays - first_rays = compaction survivors)
};

struct SceneDev {
    const float *geoms;  int ngeoms;       // GEOM_WORDS dwords each
    const float *mats;   int nmats;        // MAT_WORDS dwords each
    const float *tris;   int ntris;        // v0, e1, e2 + pad (12 dwords each)
};

struct BounceArgs {
    Pool in, out;
    Isect isect;
    SceneDev scene;
    TileMap map;
    Control *ctl;
    RangeDir dir_in;       // directory of the pool being read (mem == nullptr: dense)
    RangeDir dir_out;      // directory this launch produces
    float *fin;            // final colour planes r g b (stride in.cap), index = pid
    pt_camera cam;         // used when gen_rays != 0
    int depth, trace_depth, iter0;
    uint32_t pool_n;       // paths in the pool when compaction is off / at bounce 0
    int gen_rays;          // bounce 0 generates the camera ray instead of loading it
};

__device__ __forceinline__ int local_to_pixel(const TileMap &m, int j) {
    if (m.tile_count == 1) return j;
    int ly = j / m.W;
    int x = j - ly * m.W;
    int ls = ly / m.strip_rows;
    int y = (ls * m.tile_count + m.tile_index) * m.strip_rows + (ly - ls * m.strip_rows);
    return x + y * m.W;
}

// generateRayFromCamera (pathtrace.cu:122-143) for one pixel
__device__ __forceinline__ f3 camera_dir(const pt_camera &cam, int pix, int W) {
    const int y = pix / W;
    const int x = pix - y * W;
    f3 view = ptd::mk(cam.view.x, cam.view.y, cam.view.z);
    f3 right = ptd::mk(cam.right.x, cam.right.y, cam.right.z);
    f3 up = ptd::mk(cam.up.x, cam.up.y, cam.up.z);
    f3 a = ptd::scale(ptd::scale(right, cam.pixelLength[0]), ((float)x - (float)cam.resolution[0] * 0.5f));
    f3 b = ptd::scale(ptd::scale(up, cam.pixelLength[1]), ((float)y - (float)cam.resolution[1] * 0.5f));
    return ptd::normalize(ptd::sub(ptd::sub(view, a), b));
}

// ---------------------------------------------------------------------------
// generateRayFromCamera -> SoA pool, `count` samples (stepping interface; the
// batch path generates rays inside bounce 0)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_raygen(Pool p, pt_camera cam, TileMap map, int count,
                                                  Control *ctl) {
    uint32_t total = (uint32_t)map.tile_pixels * (uint32_t)count;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) { ctl->nlive[0] = total; }
    if (i >= total) return;
    uint32_t j = i % (uint32_t)map.tile_pixels;
    f3 d = camera_dir(cam, local_to_pixel(map, (int)j), map.W);
    char *q = p.slot(i);
    pf(q, 0) = cam.position.x; pf(q, 1) = cam.position.y; pf(q, 2) = cam.position.z;
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
constexpr int GF_WORDS = 16;         // staged per geom for gathers: transform[12], type, materialid, 2 pad
// per-wave candidate queue (PT_QUEUE): ring of 128 slots, SoA: qo.xyz qd.xyz t_obj (7 planes), meta, and
// the 64 per-lane best keys (u64)
constexpr int Q_SLOTS = 128;
constexpr int Q_WORDS = 7 * Q_SLOTS + Q_SLOTS + 2 * 64;
__host__ __device__ constexpr int scene_lds_words(int nmats, int ngeoms) {
    return ((nmats * ptd::MAT_WORDS + 3) & ~3) + PT_QUEUE * ngeoms * GF_WORDS + PT_GEOM_LDS * ngeoms * ptd::GEOM_WORDS;
}
__device__ __forceinline__ void stage_scene(float *lds_mats, const SceneDev &sc) {
    const int mw = sc.nmats * ptd::MAT_WORDS;
    for (int k = threadIdx.x; k < mw; k += BLOCK) lds_mats[k] = sc.mats[k];
#if PT_QUEUE
    {   // per-lane gathers of the tail: forward transform (12) + type + material per geom
        float *gf = lds_mats + ((mw + 3) & ~3);
        for (int k = threadIdx.x; k < sc.ngeoms * GF_WORDS; k += BLOCK) {
            const int g = k / GF_WORDS, w = k - g * GF_WORDS;
            gf[k] = w < 12 ? sc.geoms[g * ptd::GEOM_WORDS + ptd::G_FWD + w] : sc.geoms[g * ptd::GEOM_WORDS + (w - 12)];
        }
    }
#endif
#if PT_GEOM_LDS
    float *lds_geoms = lds_mats + ((mw + 3) & ~3) + PT_QUEUE * sc.ngeoms * GF_WORDS;
    for (int k = threadIdx.x; k < sc.ngeoms * ptd::GEOM_WORDS; k += BLOCK) lds_geoms[k] = sc.geoms[k];
#endif
    __syncthreads();
}

// `uniform_trips` != 0: every wave of the block executes the same geom sequence (needed when
// meshes stage triangle tiles through LDS with block barriers); `active` masks idle lanes.
// Geom records are read through the CONSTANT address space: the array is immutable for the
// lifetime of the launch and the address is wave-uniform, so the loads become s_load_dwordxN
// (scalar cache -> SGPRs) instead of per-lane vector loads.
typedef const __attribute__((address_space(4))) float cfloat;
__device__ __forceinline__ cfloat *as_const(const float *p) {
    return (cfloat *)(unsigned long long)p;
}

#if PT_QUEUE
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
#endif

template <bool HAS_MESH>
__device__ __forceinline__ void intersect_scene(const float *__restrict__ geoms, int ngeoms,
                                                const float *__restrict__ tris, float *tri_lds, bool active,
                                                f3 ro, f3 rd, ptd::Hit &h, float *wq = nullptr,
                                                const float *gf = nullptr) {
    h.t = FLT_MAX; h.geom = -1; h.outside = 1; h.aux = ptd::mk(0, 0, 0);
    int outside = 1;                                    // shared across tests, pathtrace.cu:169
    (void)outside;
#if PT_QUEUE
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
#endif
#ifdef PT_GEOM_UNROLL
#pragma unroll PT_GEOM_UNROLL
#endif
    for (int g = 0; g < ngeoms; ++g) {
#if PT_GEOM_LDS
        const float *rec = geoms + g * ptd::GEOM_WORDS;                // LDS broadcast
        const int type = __builtin_amdgcn_readfirstlane(__float_as_int(rec[0]));
#else
        cfloat *rec = as_const(geoms) + g * ptd::GEOM_WORDS;           // wave-uniform address -> s_load
        const int type = __float_as_int(rec[0]);
#endif
        if (HAS_MESH && type == PT_TRIANGLE_MESH) {
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
#if PT_QUEUE
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
#endif
        {   // monolithic reference-shaped tests (object-space test + world-distance tail in one body)
            float t = -1.0f;
            f3 aux = ptd::mk(0, 0, 0);
            if (type == PT_CUBE) { if (active) t = ptd::box_test(rec, ro, rd, aux, outside); }
            else if (type == PT_SPHERE) { if (active) t = ptd::sphere_test(rec, ro, rd, aux, outside); }
            if (t > 0.0f && h.t > t) {                  // pathtrace.cu:192 (first geom wins ties)
                h.t = t; h.geom = g; h.outside = outside; h.aux = aux;
            }
            continue;
        }
    }
#if PT_QUEUE
    if (q_total > q_head) {
        queue_pass(wq, gf, q_head, q_total - q_head, ro);
        latch();
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
#endif
}

// normal + materialId of the winning primitive (per-lane record: cached gather from the
// 1-KB record array)
__device__ __forceinline__ void resolve_hit(const float *__restrict__ geoms, const float *__restrict__ tris,
                                            const ptd::Hit &h, float &t, f3 &n, int &mat) {
    if (h.geom < 0) { t = -1.0f; n = ptd::mk(0, 0, 0); mat = 0; return; }
    const float *rec = geoms + h.geom * ptd::GEOM_WORDS;
    const int type = __float_as_int(rec[0]);
    t = h.t;
    mat = __float_as_int(rec[1]);
    if (type == PT_CUBE) n = ptd::cube_normal(rec, h.aux);
    else if (type == PT_SPHERE) n = ptd::sphere_normal(rec, h.aux, h.outside);
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
template <bool HAS_MESH>
__global__ __launch_bounds__(BLOCK, PT_MIN_WAVES) void k_intersect(Pool in, Isect out, SceneDev sc,
                                                                    const uint32_t *n_ptr, uint32_t n_fixed,
                                                                    RangeDir dir_in, const uint32_t *nprev_ptr,
                                                                    Control *ctl) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float *mats_lds = lds_raw + LDS_CTL_WORDS;
    const float *gf = mats_lds + ((sc.nmats * ptd::MAT_WORDS + 3) & ~3);
    float *wq = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + (threadIdx.x >> 6) * Q_WORDS;
    float *tri_lds = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + PT_QUEUE * WAVES * Q_WORDS;
#if PT_GEOM_LDS || PT_QUEUE
    stage_scene(mats_lds, sc);
#endif
#if PT_GEOM_LDS
    const float *gsrc = mats_lds + ((sc.nmats * ptd::MAT_WORDS + 3) & ~3) + PT_QUEUE * sc.ngeoms * GF_WORDS;
#else
    const float *gsrc = sc.geoms;
#endif
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
        if (!HAS_MESH && tile >= tiles) break;
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
        intersect_scene<HAS_MESH>(gsrc, sc.ngeoms, sc.tris, tri_lds, active, ro, rd, h, wq, gf);
        if (have && i < n) {
            float t; f3 nrm; int mat;
            resolve_hit(gsrc, sc.tris, h, t, nrm, mat);
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

template <int MODE, bool COMPACT, bool HAS_MESH>
__global__ __launch_bounds__(BLOCK, PT_MIN_WAVES) void k_bounce(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    float *mats = lds_raw + LDS_CTL_WORDS;
    const float *gf = mats + ((a.scene.nmats * ptd::MAT_WORDS + 3) & ~3);
    float *wq = mats + scene_lds_words(a.scene.nmats, a.scene.ngeoms) + (threadIdx.x >> 6) * Q_WORDS;
    float *tri_lds = mats + scene_lds_words(a.scene.nmats, a.scene.ngeoms) + PT_QUEUE * WAVES * Q_WORDS;
#ifdef PT_STAMPS
#define STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && a.depth == PT_STAMPS) a.ctl->stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
    STAMP(0);
    stage_scene(mats, a.scene);
    STAMP(1);
    const int lane = threadIdx.x & 63;
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t wid = run_id();
    const uint32_t n = (COMPACT && !a.gen_rays) ? a.ctl->nlive[a.depth] : a.pool_n;
    const uint32_t tiles = (n + TILE - 1) / TILE;
    const uint32_t R = range_tiles(n, W);                        // logical tiles per wave (one contiguous run)
    const bool packed_in = COMPACT && a.dir_in.mem != nullptr;
    const uint32_t span_in = packed_in ? range_tiles(a.ctl->nlive[a.depth - 1], W) * TILE : 0;
    const bool last_bounce = (a.depth == a.trace_depth - 1);
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
        if (!HAS_MESH && tile >= tiles) break;
        const bool have = tile < tiles;
        const uint32_t i = tile * TILE + lane;                    // logical path index
        bool active = have && i < n;
        uint32_t src = i;
        if (packed_in && have) src = resolve_src(a.dir_in, span_in, cur, i, active, a.ctl);
        uint32_t pid = DEAD_PID;
        f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1), col = ptd::mk(1.0f, 1.0f, 1.0f);
        if (active) {
            if (a.gen_rays) {
                pid = i;
            } else {
                // all ten fields of the slot in one burst of loads (one memory latency per tile)
                char *q = a.in.slot(src);
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
            if (a.gen_rays) {
                ro = ptd::mk(a.cam.position.x, a.cam.position.y, a.cam.position.z);
                rd = camera_dir(a.cam, pixel, a.map.W);
            }
        }
        if (r == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(3); }
        float t = -1.0f; f3 nrm = ptd::mk(0, 0, 0); int mat = 0; int outside = 1;
        if (MODE == MODE_FUSED) {
            ptd::Hit h;
#if PT_GEOM_LDS
            const float *gsrc = mats + ((a.scene.nmats * ptd::MAT_WORDS + 3) & ~3) + PT_QUEUE * a.scene.ngeoms * GF_WORDS;
#else
            const float *gsrc = a.scene.geoms;
#endif
            intersect_scene<HAS_MESH>(gsrc, a.scene.ngeoms, a.scene.tris, tri_lds, active, ro, rd, h, wq, gf);
            if (active) { resolve_hit(gsrc, a.scene.tris, h, t, nrm, mat); outside = h.outside; }
        } else if (active) {
            // MODE_ISECT: planes in logical order; MODE_CACHE0: one record per pixel of the tile
            const uint32_t q = (MODE == MODE_CACHE0) ? pid - smp * (uint32_t)a.map.tile_pixels : i;
            t = at(a.isect.plane(0), q);
            nrm = ptd::mk(at(a.isect.plane(1), q), at(a.isect.plane(2), q), at(a.isect.plane(3), q));
            const int m = at(a.isect.mat(), q);
            mat = m & 0x7fffffff; outside = (m < 0) ? 0 : 1;
        }
        if (r == 0) STAMP(4);
        bool alive = false;
        ptd::PathState ps;
        ps.o = ro; ps.d = rd; ps.c = col;
        if (active) {
            alive = ptd::shade_scatter(ps, t, nrm, mat, outside, mats, a.iter0 + (int)smp, pixel, a.depth,
                                       last_bounce);
            if (!alive) {
                at(a.fin, pid) = ps.c.x; at(a.fin + (size_t)a.in.cap, pid) = ps.c.y;
                at(a.fin + 2 * (size_t)a.in.cap, pid) = ps.c.z;
            }
        }
        if (r == 0) STAMP(5);
        // ---- survivors append to the wave's packed run (wave64 ballot + popcount rank) ----
        const uint64_t bal = __ballot(alive);
        const uint64_t act = __ballot(active);
        traced += (uint32_t)__popcll((unsigned long long)act);
        uint32_t dst = i;
        if (COMPACT) {
            dst = wid * R * TILE + packed + (uint32_t)__popcll((unsigned long long)(bal & ((1ull << lane) - 1)));
            packed += (uint32_t)__popcll((unsigned long long)bal);
        }
        if (alive) {
            char *q = a.out.slot(dst);
            pf(q, 0) = ps.o.x; pf(q, 1) = ps.o.y; pf(q, 2) = ps.o.z;
            pf(q, 3) = ps.d.x; pf(q, 4) = ps.d.y; pf(q, 5) = ps.d.z;
            pf(q, 6) = ps.c.x; pf(q, 7) = ps.c.y; pf(q, 8) = ps.c.z;
            ppid(q) = pid;
        } else if (!COMPACT && have && i < n) {
            a.out.pid(dst) = DEAD_PID;
        }
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

// First-bounce cache (INSTRUCTION.md:87-89): camera rays do not depend on the iteration (no
// jitter, pathtrace.cu:134), so computeIntersections of bounce 0 is evaluated once per pixel and
// camera and reused by every sample.
template <bool HAS_MESH>
__global__ __launch_bounds__(BLOCK, PT_MIN_WAVES) void k_cache_first(Isect cache, SceneDev sc, pt_camera cam,
                                                                      TileMap map) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    float *mats_lds = lds_raw + LDS_CTL_WORDS;
    const float *gf = mats_lds + ((sc.nmats * ptd::MAT_WORDS + 3) & ~3);
    float *wq = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + (threadIdx.x >> 6) * Q_WORDS;
    float *tri_lds = mats_lds + scene_lds_words(sc.nmats, sc.ngeoms) + PT_QUEUE * WAVES * Q_WORDS;
#if PT_GEOM_LDS || PT_QUEUE
    stage_scene(mats_lds, sc);
#endif
#if PT_GEOM_LDS
    const float *gsrc = mats_lds + ((sc.nmats * ptd::MAT_WORDS + 3) & ~3) + PT_QUEUE * sc.ngeoms * GF_WORDS;
#else
    const float *gsrc = sc.geoms;
#endif
    const uint32_t n = (uint32_t)map.tile_pixels;
    const uint32_t tiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint32_t j = tile * BLOCK + threadIdx.x;
        const bool active = j < n;
        f3 ro = ptd::mk(cam.position.x, cam.position.y, cam.position.z), rd = ptd::mk(0, 0, 1);
        if (active) rd = camera_dir(cam, local_to_pixel(map, (int)j), map.W);
        ptd::Hit h;
        intersect_scene<HAS_MESH>(gsrc, sc.ngeoms, sc.tris, tri_lds, active, ro, rd, h, wq, gf);
        if (active) {
            float t; f3 nrm; int mat;
            resolve_hit(gsrc, sc.tris, h, t, nrm, mat);
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
                                                  int count, const Control *ctl, Persist *per, int depths,
                                                  uint32_t fake_rays) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j == 0) {                    // fold this batch's ray count into the persistent counter
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

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(PT_ERR_DEVICE, "HIP error (%s:%d): %s: %s", "ptmi355.hip", __LINE__, #expr, \
                        hipGetErrorString(e_));                                                 \
    } while (0)

struct Renderer {
    bool live = false;
    pt_scene_desc desc{};
    pt_camera cam{};
    int trace_depth = 0;
    uint32_t flags = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    TileMap map{};
    int npix = 0;                 // full frame
    uint32_t cap = 0;             // pool capacity = max_batch * tile_pixels
    int max_batch = 1;
    float *pool_mem[2] = {nullptr, nullptr};
    Pool pool[2]{};
    int cur = 0;                  // pool holding the current live prefix
    float *isect_mem = nullptr;
    float *isect2_mem = nullptr;  // sorted intersections (PT_SORT_MATERIAL)
    uint32_t *sort_table = nullptr;
    float *cache_mem = nullptr;   // first-bounce cache: 5 planes of tile_pixels (PT_CACHE_FIRST)
    bool cache_valid = false;
    Isect isect{};
    float *final_mem = nullptr;   // 3 planes of cap floats
    float *image = nullptr;
    bool own_image = false;
    float *d_geoms = nullptr, *d_mats = nullptr, *d_tris = nullptr;
    SceneDev scene{};
    size_t lds_bytes = 0;
    Control *ctl = nullptr;
    Persist *persist = nullptr;
    uint32_t *dir_mem = nullptr;  // per bounce: count[Wp], base[Wp+4]
    size_t dir_stride = 0;        // words per bounce
    int cur_dir = -1;             // bounce whose directory describes pool[cur] (-1: dense)
    uint32_t max_tiles = 0;
    size_t ctl_bytes = 0;         // Control, zeroed per batch
    int grid = 0;                 // persistent grid size
    bool sorted_isects = false;   // the last bounce's intersections live in isect2 (sorted order)
    bool gen_fused = false;       // bounce 0 of the current batch generates its own rays
    bool has_mesh = false;
    void *scratch = nullptr;      // export / import staging
    size_t scratch_bytes = 0;
    // stepping state
    int step_iter0 = 0, step_count = 0, step_depth = 0;
    bool in_step = false;
    pt_stats stats{};
    // optional per-kernel HIP-event timing
    bool profiling = false;
    std::vector<hipEvent_t> ev;       // pairs (start, stop)
    std::vector<int> ev_stage;        // stage of each recorded pair
    size_t ev_used = 0;               // pairs recorded since the last drain
    pt_profile prof{};
} R;

constexpr size_t EV_PAIRS = 2048;

int drain_events(void) {
    if (R.ev_used == 0) return PT_OK;
    HIPCHK(hipStreamSynchronize(R.stream));
    for (size_t k = 0; k < R.ev_used; ++k) {
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, R.ev[2 * k], R.ev[2 * k + 1]));
        R.prof.ms[R.ev_stage[k]] += (double)ms;
        R.prof.launches[R.ev_stage[k]] += 1;
    }
    R.ev_used = 0;
    return PT_OK;
}

struct StageTimer {                  // brackets one launch when profiling is on
    bool on;
    size_t k;
    StageTimer(int stage) : on(false), k(0) {
        if (!R.profiling) return;
        if (R.ev_used >= EV_PAIRS && drain_events() != PT_OK) return;
        k = R.ev_used++;
        R.ev_stage[k] = stage;
        on = hipEventRecord(R.ev[2 * k], R.stream) == hipSuccess;
    }
    ~StageTimer() { if (on) (void)hipEventRecord(R.ev[2 * k + 1], R.stream); }
};

Pool carve_pool(float *mem, uint32_t cap) { return Pool{mem, cap}; }

// magic / shift for n / d, d >= 1, exact for all 32-bit n (checked on probes in pt_init)
void make_div_magic(uint32_t d, uint32_t *magic, uint32_t *shift) {
    if (d == 1) { *magic = 0; *shift = 0; return; }                     // handled separately in sample_of
    uint32_t L = 31;
    while (!((d >> L) & 1u)) --L;                                       // floor(log2 d)
    if ((d & (d - 1)) == 0) { *magic = 0; *shift = L - 1; return; }     // power of two: (n >> 1) >> (L - 1)
    const uint64_t num = 1ull << (32 + L);
    uint64_t m = num / d, rem = num % d;
    m += m;
    const uint64_t twice = rem + rem;
    if (twice >= d) m += 1;
    *magic = (uint32_t)(m + 1);
    *shift = L;
}

int tile_rows(int tile_index, int tile_count, int strip_rows, int H) {
    if (tile_count <= 1) return H;
    int rows = 0;
    for (int y = 0; y < H; ++y)
        if ((y / strip_rows) % tile_count == tile_index) rows++;
    return rows;
}

int ensure_scratch(size_t bytes) {
    if (bytes <= R.scratch_bytes) return PT_OK;
    if (R.scratch) (void)hipFree(R.scratch);
    R.scratch = nullptr; R.scratch_bytes = 0;
    HIPCHK(hipMalloc(&R.scratch, bytes));
    R.scratch_bytes = bytes;
    return PT_OK;
}

RangeDir tile_dir(int depth) {
    const uint32_t W = (uint32_t)R.grid * WAVES;
    if (depth < 0) return RangeDir{nullptr, W};
    return RangeDir{R.dir_mem + (size_t)depth * R.dir_stride, W};
}

BounceArgs bounce_args(int depth) {
    BounceArgs a{};
    a.in = R.pool[R.cur];
    a.out = (R.flags & PT_COMPACT) ? R.pool[R.cur ^ 1] : R.pool[R.cur];
    a.isect = R.isect;
    a.scene = R.scene;
    a.map = R.map;
    a.ctl = R.ctl;
    a.dir_in = tile_dir((R.flags & PT_COMPACT) ? R.cur_dir : -1);
    a.dir_out = tile_dir(depth);
    a.fin = R.final_mem;
    a.cam = R.cam;
    a.depth = depth; a.trace_depth = R.trace_depth; a.iter0 = R.step_iter0;
    a.pool_n = (uint32_t)R.map.tile_pixels * (uint32_t)R.step_count;
    a.gen_rays = (depth == 0 && R.gen_fused) ? 1 : 0;
    return a;
}

int enqueue_begin(int iter0, int count, bool stepping) {
    if (count < 1 || count > R.max_batch)
        return fail(PT_ERR_INVALID, "batch count %d outside [1, max_batch=%d]", count, R.max_batch);
    if (iter0 < 1 || (int64_t)iter0 + count - 1 >= (1 << 22))
        return fail(PT_ERR_INVALID, "iteration %d outside [1, 2^22): makeSeededRandomEngine packs iter in 22 bits", iter0);
    R.step_iter0 = iter0; R.step_count = count; R.step_depth = 0; R.cur = 0; R.cur_dir = -1;
    R.sorted_isects = false;
    HIPCHK(hipMemsetAsync(R.ctl, 0, R.ctl_bytes, R.stream));
    // batch path: bounce 0 generates the camera rays itself (no 40 B/path round trip through HBM)
    R.gen_fused = !stepping && !(R.flags & (PT_UNFUSED | PT_SORT_MATERIAL | PT_FAKE_SHADER));
    if (R.gen_fused) return PT_OK;
    const uint32_t total = (uint32_t)R.map.tile_pixels * (uint32_t)count;
    StageTimer tm(PT_STAGE_RAYGEN);
    hipLaunchKernelGGL(k_raygen, dim3((total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, R.pool[0], R.cam,
                       R.map, count, R.ctl);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

template <bool HAS_MESH>
void launch_intersect(const Pool &in, const uint32_t *n_ptr, uint32_t n_fixed, const RangeDir &dir,
                      const uint32_t *nprev) {
    hipLaunchKernelGGL((k_intersect<HAS_MESH>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, in, R.isect,
                       R.scene, n_ptr, n_fixed, dir, nprev, R.ctl);
}

template <int MODE, bool COMPACT>
void launch_bounce(const BounceArgs &a) {
    if (R.has_mesh) hipLaunchKernelGGL((k_bounce<MODE, COMPACT, true>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, a);
    else hipLaunchKernelGGL((k_bounce<MODE, COMPACT, false>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, a);
}

int enqueue_bounce(int depth) {
    BounceArgs a = bounce_args(depth);
    const bool compact = (R.flags & PT_COMPACT) != 0;
    const bool unfused = (R.flags & (PT_UNFUSED | PT_SORT_MATERIAL)) != 0;
    if (unfused) {
        StageTimer tm(PT_STAGE_INTERSECT);
        const uint32_t *n_ptr = compact ? &R.ctl->nlive[depth] : (const uint32_t *)nullptr;
        const uint32_t *nprev = (compact && depth > 0) ? &R.ctl->nlive[depth - 1] : (const uint32_t *)nullptr;
        if (R.has_mesh) launch_intersect<true>(a.in, n_ptr, a.pool_n, a.dir_in, nprev);
        else launch_intersect<false>(a.in, n_ptr, a.pool_n, a.dir_in, nprev);
        HIPCHK(hipGetLastError());
    }
    if (R.flags & PT_SORT_MATERIAL) {
        // intersections (logical order) -> histogram + scan -> scatter into the other pool, dense and
        // sorted; the shade/compact kernel then reads that pool and writes back into the first one
        StageTimer tm(PT_STAGE_SORT);
        SortArgs sa{};
        sa.in = a.in; sa.out = R.pool[R.cur ^ 1];
        sa.isect = R.isect; sa.isect_out = Isect{R.isect2_mem, R.cap};
        sa.dir_in = a.dir_in; sa.ctl = R.ctl; sa.table = R.sort_table;
        sa.depth = depth; sa.nbins = R.scene.nmats + 1; sa.pool_n = a.pool_n; sa.compact = compact ? 1 : 0;
        const size_t lds = ((size_t)LDS_CTL_WORDS + (size_t)WAVES * SORT_MAX_BINS) * 4;
        hipLaunchKernelGGL(k_sort_hist, dim3(R.grid), dim3(BLOCK), lds, R.stream, sa);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_sort_scatter, dim3(R.grid), dim3(BLOCK), lds, R.stream, sa);
        HIPCHK(hipGetLastError());
        a.in = sa.out; a.out = R.pool[R.cur];
        a.isect = sa.isect_out;
        a.dir_in = tile_dir(-1);                         // the sorted pool is dense
    }
    const bool cached0 = depth == 0 && !unfused && (R.flags & PT_CACHE_FIRST);
    if (cached0 && !R.cache_valid) {
        StageTimer tm(PT_STAGE_INTERSECT);
        const Isect cache{R.cache_mem, (uint32_t)R.map.tile_pixels};
        const int blocks = std::min(R.grid, (R.map.tile_pixels + BLOCK - 1) / BLOCK);
        if (R.has_mesh) hipLaunchKernelGGL((k_cache_first<true>), dim3(blocks), dim3(BLOCK), R.lds_bytes, R.stream, cache, R.scene, R.cam, R.map);
        else hipLaunchKernelGGL((k_cache_first<false>), dim3(blocks), dim3(BLOCK), R.lds_bytes, R.stream, cache, R.scene, R.cam, R.map);
        HIPCHK(hipGetLastError());
        R.cache_valid = true;
    }
    StageTimer tm(PT_STAGE_BOUNCE);
    if (cached0) {
        a.isect = Isect{R.cache_mem, (uint32_t)R.map.tile_pixels};
        if (compact) launch_bounce<MODE_CACHE0, true>(a); else launch_bounce<MODE_CACHE0, false>(a);
    } else if (unfused) {
        if (compact) launch_bounce<MODE_ISECT, true>(a); else launch_bounce<MODE_ISECT, false>(a);
    } else {
        if (compact) launch_bounce<MODE_FUSED, true>(a); else launch_bounce<MODE_FUSED, false>(a);
    }
    HIPCHK(hipGetLastError());
    if (R.flags & PT_SORT_MATERIAL) {
        // pool[cur] was rewritten in place of the pre-sort state; without compaction the survivors
        // stay where the sorted pass put them, i.e. in pool[cur] as well
        if (compact) R.cur_dir = depth;
        R.sorted_isects = true;
    } else if (compact) { R.cur ^= 1; R.cur_dir = depth; }
    R.step_depth = depth + 1;
    return PT_OK;
}

int enqueue_fake(void) {
    // the reference as shipped (pathtrace.cu:339-377): one bounce, fake shader
    const uint32_t total = (uint32_t)R.map.tile_pixels * (uint32_t)R.step_count;
    if (R.has_mesh) launch_intersect<true>(R.pool[R.cur], nullptr, total, tile_dir(-1), nullptr);
    else launch_intersect<false>(R.pool[R.cur], nullptr, total, tile_dir(-1), nullptr);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_shade_fake, dim3((total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, R.pool[R.cur],
                       R.isect, R.scene.mats, R.map, R.step_iter0, total, R.final_mem);
    HIPCHK(hipGetLastError());
    R.step_depth = 1;
    return PT_OK;
}

int enqueue_end(void) {
    StageTimer tm(PT_STAGE_GATHER);
    hipLaunchKernelGGL(k_gather, dim3((R.map.tile_pixels + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, R.image,
                       R.final_mem, R.cap, R.map,
                       R.step_count, R.ctl, R.persist, (R.flags & PT_FAKE_SHADER) ? 0 : R.trace_depth,
                       (R.flags & PT_FAKE_SHADER) ? (uint32_t)R.map.tile_pixels * (uint32_t)R.step_count : 0u);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

int enqueue_batch(int iter0, int count) {
    int rc = enqueue_begin(iter0, count, false);
    if (rc) return rc;
    if (R.flags & PT_FAKE_SHADER) {
        rc = enqueue_fake();
        if (rc) return rc;
    } else {
        for (int d = 0; d < R.trace_depth; ++d) {
            rc = enqueue_bounce(d);
            if (rc) return rc;
        }
    }
    return enqueue_end();
}

// reads the control block back (after a sync) and folds it into the stats
int collect_stats(void) {
    Control c;
    HIPCHK(hipMemcpyAsync(&c, R.ctl, offsetof(Control, bucket), hipMemcpyDeviceToHost, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));
    if (c.error) return fail(PT_ERR_INTERNAL, "kernel watchdog tripped: inconsistent tile directory (control.error=%u)", c.error);
    R.stats.bounces = 0; R.stats.rays = 0;
    memset(R.stats.live, 0, sizeof R.stats.live);
    if (R.flags & PT_FAKE_SHADER) {
        R.stats.live[0] = R.map.tile_pixels * R.step_count; R.stats.rays = R.stats.live[0]; R.stats.bounces = 1;
    } else {
        for (int d = 0; d < R.trace_depth && d < 64; ++d) {
            R.stats.live[d] = (int32_t)c.alive[d];
            R.stats.rays += c.alive[d];
            if (c.alive[d]) R.stats.bounces = d + 1;
        }
    }
#ifdef PT_STAMPS
    fprintf(stderr, "[ptmi355] stamps (us since block 0 start, bounce %d): stage %.1f range %.1f loaded %.1f isect %.1f shade %.1f loop-end %.1f elect %.1f | last block: scan-start %.1f scan-end %.1f\n",
            (int)PT_STAMPS, (c.stamp[1] - c.stamp[0]) / 100.0, (c.stamp[2] - c.stamp[0]) / 100.0, (c.stamp[3] - c.stamp[0]) / 100.0,
            (c.stamp[4] - c.stamp[0]) / 100.0, (c.stamp[5] - c.stamp[0]) / 100.0, (c.stamp[6] - c.stamp[0]) / 100.0,
            (c.stamp[7] - c.stamp[0]) / 100.0, ((double)c.stamp[8] - (double)c.stamp[0]) / 100.0, ((double)c.stamp[9] - (double)c.stamp[0]) / 100.0);
#endif
    if (getenv("PTMI355_DEBUG_SCAN")) {
        fprintf(stderr, "[ptmi355] scan us per bounce:");
        for (int d = 0; d < R.trace_depth; ++d) fprintf(stderr, " %.1f", c.scan_ticks[d] / 100.0);
        fprintf(stderr, "\n");
    }
    R.stats.total_rays += R.stats.rays;
    R.stats.total_iterations += R.step_count;
    return PT_OK;
}

}  // namespace

// ===========================================================================
// C-ABI
// ===========================================================================
extern "C" {

const char *pt_last_error(void) { return g_err; }
const char *pt_version(void) { return "ptmi355 0.1 (gfx950, fp32 no-contract, wave64)"; }

void pt_free(void) {
    if (!R.live && !R.scratch) return;
    if (R.stream) (void)hipStreamSynchronize(R.stream);
    for (int k = 0; k < 2; ++k) if (R.pool_mem[k]) (void)hipFree(R.pool_mem[k]);
    if (R.isect_mem) (void)hipFree(R.isect_mem);
    if (R.isect2_mem) (void)hipFree(R.isect2_mem);
    if (R.sort_table) (void)hipFree(R.sort_table);
    if (R.cache_mem) (void)hipFree(R.cache_mem);
    if (R.final_mem) (void)hipFree(R.final_mem);
    if (R.image && R.own_image) (void)hipFree(R.image);
    if (R.d_geoms) (void)hipFree(R.d_geoms);
    if (R.d_mats) (void)hipFree(R.d_mats);
    if (R.d_tris) (void)hipFree(R.d_tris);
    if (R.ctl) (void)hipFree(R.ctl);
    if (R.dir_mem) (void)hipFree(R.dir_mem);
    if (R.persist) (void)hipFree(R.persist);
    if (R.scratch) (void)hipFree(R.scratch);
    for (hipEvent_t e : R.ev) (void)hipEventDestroy(e);
    if (R.stream && R.own_stream) (void)hipStreamDestroy(R.stream);
    R = Renderer{};
}

static int init_impl(const pt_scene_desc *d);

int pt_init(const pt_scene_desc *d) {
    if (!d) return fail(PT_ERR_INVALID, "pt_init: null descriptor");
    if (R.live) pt_free();
    const int rc = init_impl(d);
    if (rc != PT_OK) {                 // release whatever was allocated; keep the message
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        R.live = true;
        pt_free();
        memcpy(g_err, keep, sizeof keep);
    }
    return rc;
}

static int init_impl(const pt_scene_desc *d) {
    const int W = d->camera.resolution[0], H = d->camera.resolution[1];
    if (W <= 0 || H <= 0 || (int64_t)W * H > (1 << 28)) return fail(PT_ERR_INVALID, "pt_init: bad resolution %dx%d", W, H);
    if (d->num_geoms < 0 || d->num_materials <= 0 || (d->num_geoms > 0 && !d->geoms) || !d->materials)
        return fail(PT_ERR_INVALID, "pt_init: geoms/materials missing");
    if (d->trace_depth < 1 || d->trace_depth > MAX_DEPTH) return fail(PT_ERR_INVALID, "pt_init: trace_depth %d outside [1,%d]", d->trace_depth, MAX_DEPTH);
    const int tile_count = d->tile_count <= 0 ? 1 : d->tile_count;
    if (d->tile_index < 0 || d->tile_index >= tile_count) return fail(PT_ERR_INVALID, "pt_init: tile_index %d / tile_count %d", d->tile_index, tile_count);
    if (tile_count > 1 && d->strip_rows <= 0) return fail(PT_ERR_INVALID, "pt_init: strip_rows must be > 0 when tiling");
    for (int i = 0; i < d->num_geoms; ++i) {
        const pt_geom &g = d->geoms[i];
        if (g.type < PT_SPHERE || g.type > PT_TRIANGLE_MESH) return fail(PT_ERR_INVALID, "pt_init: geom %d has type %d", i, g.type);
        if (g.materialid < 0 || g.materialid >= d->num_materials) return fail(PT_ERR_INVALID, "pt_init: geom %d materialid %d out of range", i, g.materialid);
    }
    for (int k = 0; k < d->num_meshes; ++k) {
        const pt_mesh &m = d->meshes[k];
        if (m.geom_index < 0 || m.geom_index >= d->num_geoms || d->geoms[m.geom_index].type != PT_TRIANGLE_MESH ||
            m.first_triangle < 0 || m.triangle_count < 0 || m.first_triangle + m.triangle_count > d->num_triangles)
            return fail(PT_ERR_INVALID, "pt_init: mesh %d is inconsistent", k);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PT_ERR_DEVICE, "pt_init: no HIP device (this library has no CPU fallback)");
    if (d->device < 0 || d->device >= ndev) return fail(PT_ERR_INVALID, "pt_init: device %d of %d", d->device, ndev);
    HIPCHK(hipSetDevice(d->device));

    R = Renderer{};
    R.desc = *d; R.cam = d->camera; R.trace_depth = d->trace_depth; R.flags = d->flags; R.device = d->device;
    R.npix = W * H;
    R.map.W = W; R.map.H = H; R.map.tile_index = d->tile_index; R.map.tile_count = tile_count;
    R.map.strip_rows = tile_count > 1 ? d->strip_rows : H;
    R.map.tile_pixels = tile_rows(d->tile_index, tile_count, R.map.strip_rows, H) * W;
    if (R.map.tile_pixels <= 0) return fail(PT_ERR_INVALID, "pt_init: tile owns no rows");
    make_div_magic((uint32_t)R.map.tile_pixels, &R.map.div_magic, &R.map.div_shift);
    {   // the magic must reproduce n / tile_pixels exactly; probe the edges of every sample and the extremes
        const uint32_t d = (uint32_t)R.map.tile_pixels;
        auto fast = [&](uint32_t n) {
            if (d == 1) return n;
            const uint32_t q = (uint32_t)(((uint64_t)R.map.div_magic * n) >> 32);
            return (((n - q) >> 1) + q) >> R.map.div_shift;
        };
        for (uint64_t k = 0; k <= 0xffffffffull / d && k < 4096; ++k)
            for (int e = -1; e <= 1; ++e) {
                const uint64_t n = k * d + (uint64_t)(int64_t)e;
                if (n <= 0xffffffffull && fast((uint32_t)n) != (uint32_t)n / d)
                    return fail(PT_ERR_INTERNAL, "pt_init: division magic failed for %u / %u", (uint32_t)n, d);
            }
        const uint32_t probes[] = {0u, 1u, d - 1, d, d + 1, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu};
        for (uint32_t n : probes)
            if (fast(n) != n / d) return fail(PT_ERR_INTERNAL, "pt_init: division magic failed for %u / %u", n, d);
    }
    R.max_batch = d->max_batch < 1 ? 1 : d->max_batch;
    if ((int64_t)R.max_batch * R.map.tile_pixels >= (int64_t)0x3ffffff0)
        return fail(PT_ERR_INVALID, "pt_init: max_batch * tile pixels must stay below 2^30 (32-bit byte offsets into the planes)");
    R.cap = (uint32_t)R.max_batch * (uint32_t)R.map.tile_pixels;
    if (d->stream) { R.stream = (hipStream_t)d->stream; R.own_stream = false; }
    else { HIPCHK(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking)); R.own_stream = true; }
    R.live = true;

    // scene -> device records
    std::vector<float> grec((size_t)std::max(1, d->num_geoms) * ptd::GEOM_WORDS, 0.0f);
    for (int i = 0; i < d->num_geoms; ++i) {
        const pt_geom &g = d->geoms[i];
        float *r = grec.data() + (size_t)i * ptd::GEOM_WORDS;
        int first = 0, count = 0;
        for (int k = 0; k < d->num_meshes; ++k)
            if (d->meshes[k].geom_index == i) { first = d->meshes[k].first_triangle; count = d->meshes[k].triangle_count; break; }
        memcpy(&r[0], &g.type, 4); memcpy(&r[1], &g.materialid, 4); memcpy(&r[2], &first, 4); memcpy(&r[3], &count, 4);
        const pt_mat4 *ms[3] = {&g.inverseTransform, &g.transform, &g.invTranspose};
        const int offs[3] = {ptd::G_INV, ptd::G_FWD, ptd::G_INVT};
        for (int m = 0; m < 3; ++m)
            for (int c = 0; c < 4; ++c)
                for (int rr = 0; rr < 3; ++rr) r[offs[m] + c * 3 + rr] = ms[m]->m[c][rr];
    }
    std::vector<float> mrec((size_t)d->num_materials * ptd::MAT_WORDS, 0.0f);
    for (int i = 0; i < d->num_materials; ++i) {
        const pt_material &m = d->materials[i];
        float *r = mrec.data() + (size_t)i * ptd::MAT_WORDS;
        r[0] = m.color.x; r[1] = m.color.y; r[2] = m.color.z;
        r[3] = m.specular.color.x; r[4] = m.specular.color.y; r[5] = m.specular.color.z;
        r[6] = m.hasReflective; r[7] = m.hasRefractive; r[8] = m.indexOfRefraction; r[9] = m.emittance;
    }
    std::vector<float> trec((size_t)std::max(1, d->num_triangles) * TRI_WORDS, 0.0f);
    for (int i = 0; i < d->num_triangles; ++i) {
        const pt_triangle &t = d->triangles[i];
        float *r = trec.data() + (size_t)i * TRI_WORDS;
        r[0] = t.v0.x; r[1] = t.v0.y; r[2] = t.v0.z;
        // e1 = v1 - v0, e2 = v2 - v0: the first two statements of glm::intersectRayTriangle, hoisted
        r[3] = t.v1.x - t.v0.x; r[4] = t.v1.y - t.v0.y; r[5] = t.v1.z - t.v0.z;
        r[6] = t.v2.x - t.v0.x; r[7] = t.v2.y - t.v0.y; r[8] = t.v2.z - t.v0.z;
    }
    HIPCHK(hipMalloc(&R.d_geoms, grec.size() * 4));
    HIPCHK(hipMalloc(&R.d_mats, mrec.size() * 4));
    HIPCHK(hipMalloc(&R.d_tris, trec.size() * 4));
    HIPCHK(hipMemcpy(R.d_geoms, grec.data(), grec.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(R.d_mats, mrec.data(), mrec.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(R.d_tris, trec.data(), trec.size() * 4, hipMemcpyHostToDevice));
    R.scene.geoms = R.d_geoms; R.scene.ngeoms = d->num_geoms;
    R.scene.mats = R.d_mats; R.scene.nmats = d->num_materials;
    R.scene.tris = R.d_tris; R.scene.ntris = d->num_triangles;
    R.has_mesh = false;
    for (int i = 0; i < d->num_geoms; ++i) R.has_mesh |= d->geoms[i].type == PT_TRIANGLE_MESH;
    R.lds_bytes = ((size_t)LDS_CTL_WORDS + (size_t)scene_lds_words(d->num_materials, d->num_geoms) +
                   (size_t)PT_QUEUE * WAVES * Q_WORDS) * 4;
    R.lds_bytes = (R.lds_bytes + 15) & ~(size_t)15;
    if (const char *pad = getenv("PTMI355_LDS_PAD")) R.lds_bytes += (size_t)atoi(pad);     // occupancy experiments
    if (R.has_mesh) R.lds_bytes += (size_t)TRI_TILE * TRI_WORDS * 4;
    if (R.lds_bytes > 60 * 1024) return fail(PT_ERR_INVALID, "pt_init: material records need %zu B of LDS (> 60 KiB)", R.lds_bytes);

    // pools, intersections, final colours, image, control
    const size_t capz = R.cap;
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipMalloc(&R.pool_mem[k], (((capz + 63) / 64) * 64) * 10 * 4));     // whole 64-path tiles
        R.pool[k] = carve_pool(R.pool_mem[k], R.cap);
    }
    HIPCHK(hipMalloc(&R.isect_mem, capz * 5 * 4));
    R.isect = Isect{R.isect_mem, R.cap};
    HIPCHK(hipMalloc(&R.final_mem, capz * 3 * 4));
    if (d->device_image) { R.image = d->device_image; R.own_image = false; }
    else {
        HIPCHK(hipMalloc(&R.image, (size_t)R.npix * 3 * 4));
        R.own_image = true;
        HIPCHK(hipMemsetAsync(R.image, 0, (size_t)R.npix * 3 * 4, R.stream));      // pathtrace.cu:85
    }
    R.max_tiles = (R.cap + TILE - 1) / TILE;
    // only the election buckets of the bounces this scene can run are cleared per batch
    R.ctl_bytes = offsetof(Control, bucket) + (size_t)R.trace_depth * sizeof(((Control *)nullptr)->bucket[0]);
    HIPCHK(hipMalloc((void **)&R.ctl, sizeof(Control)));

    HIPCHK(hipMalloc((void **)&R.persist, sizeof(Persist)));
    HIPCHK(hipMemsetAsync(R.persist, 0, sizeof(Persist), R.stream));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, d->device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // persistent grid: as many workgroups as are co-resident for the fused kernel (tiles are
    // dealt round-robin, so more workgroups than that only re-stage the scene)
    int per_cu = 0;
    if (R.has_mesh)
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_bounce<MODE_FUSED, true, true>,
                                                            BLOCK, R.lds_bytes));
    else
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_bounce<MODE_FUSED, true, false>,
                                                            BLOCK, R.lds_bytes));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    R.grid = (int)std::min<uint32_t>((R.max_tiles + WAVES - 1) / WAVES, (uint32_t)cus * (uint32_t)per_cu);
    if (R.grid < 1) R.grid = 1;
    if (R.flags & PT_CACHE_FIRST) HIPCHK(hipMalloc(&R.cache_mem, (size_t)R.map.tile_pixels * 5 * 4));
    if (R.flags & PT_SORT_MATERIAL) {
        if (d->num_materials + 1 > SORT_MAX_BINS)
            return fail(PT_ERR_INVALID, "pt_init: PT_SORT_MATERIAL supports at most %d materials", SORT_MAX_BINS - 1);
        HIPCHK(hipMalloc(&R.isect2_mem, capz * 5 * 4));
        HIPCHK(hipMalloc((void **)&R.sort_table, (size_t)(d->num_materials + 1) * R.grid * WAVES * sizeof(uint32_t)));
    }
    {   // range directory: one count + one base per wave of the persistent grid, per bounce
        const size_t Wp = ((size_t)R.grid * WAVES + 3) & ~(size_t)3;
        R.dir_stride = 2 * Wp + 8;
        HIPCHK(hipMalloc((void **)&R.dir_mem, (size_t)R.trace_depth * R.dir_stride * sizeof(uint32_t)));
    }
    HIPCHK(hipStreamSynchronize(R.stream));
    g_err[0] = 0;
    return PT_OK;
}

int pt_set_camera(const pt_camera *camera, int trace_depth) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_set_camera: not initialised");
    if (!camera) return fail(PT_ERR_INVALID, "pt_set_camera: null camera");
    if (camera->resolution[0] != R.map.W || camera->resolution[1] != R.map.H)
        return fail(PT_ERR_INVALID, "pt_set_camera: resolution changed (%dx%d -> %dx%d); re-init instead",
                    R.map.W, R.map.H, camera->resolution[0], camera->resolution[1]);
    if (trace_depth < 1 || trace_depth > R.desc.trace_depth)
        return fail(PT_ERR_INVALID, "pt_set_camera: trace_depth %d outside [1, %d]", trace_depth, R.desc.trace_depth);
    if (memcmp(&R.cam, camera, sizeof R.cam) != 0) R.cache_valid = false;     // new camera: refill the bounce-0 cache
    R.cam = *camera;
    R.trace_depth = trace_depth;
    return PT_OK;
}

int pt_synchronize(void) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_synchronize: not initialised");
    HIPCHK(hipStreamSynchronize(R.stream));
    return PT_OK;
}

int pt_trace_batch_async(int iter0, int count) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace_batch_async: not initialised");
    R.in_step = false;
    return enqueue_batch(iter0, count);
}

int pt_trace_batch(int iter0, int count, float *host_image_sum) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace_batch: not initialised");
    R.in_step = false;
    int rc = enqueue_batch(iter0, count);
    if (rc) return rc;
    rc = collect_stats();
    if (rc) return rc;
    if (host_image_sum) HIPCHK(hipMemcpy(host_image_sum, R.image, (size_t)R.npix * 12, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_trace(uint8_t *pbo_rgba, int frame, int iter, float *host_image_sum) {
    (void)frame;                                          // unused in the reference too (main.cpp:136)
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace: not initialised");
    R.in_step = false;
    int rc = enqueue_batch(iter, 1);
    if (rc) return rc;
    if (pbo_rgba) {
        hipLaunchKernelGGL(k_tonemap, dim3((R.npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, pbo_rgba,
                           R.image, R.npix, iter);
        HIPCHK(hipGetLastError());
    }
    rc = collect_stats();
    if (rc) return rc;
    if (host_image_sum) HIPCHK(hipMemcpy(host_image_sum, R.image, (size_t)R.npix * 12, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_trace_begin(int iter0, int count) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace_begin: not initialised");
    int rc = enqueue_begin(iter0, count, true);
    if (rc) return rc;
    R.in_step = true;
    HIPCHK(hipStreamSynchronize(R.stream));
    return PT_OK;
}

int pt_trace_bounce(int depth, int *n_live_after) {
    if (!R.live || !R.in_step) return fail(PT_ERR_INVALID, "pt_trace_bounce: call pt_trace_begin first");
    if (depth != R.step_depth || depth >= R.trace_depth)
        return fail(PT_ERR_INVALID, "pt_trace_bounce: depth %d, expected %d (< %d)", depth, R.step_depth, R.trace_depth);
    int rc = (R.flags & PT_FAKE_SHADER) ? enqueue_fake() : enqueue_bounce(depth);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(R.stream));
    if (n_live_after) {
        uint32_t n = 0;
        if ((R.flags & PT_COMPACT) && !(R.flags & PT_FAKE_SHADER))
            HIPCHK(hipMemcpy(&n, &R.ctl->nlive[depth + 1], 4, hipMemcpyDeviceToHost));
        else n = (uint32_t)R.map.tile_pixels * (uint32_t)R.step_count;
        *n_live_after = (int)n;
    }
    return PT_OK;
}

int pt_trace_end(void) {
    if (!R.live || !R.in_step) return fail(PT_ERR_INVALID, "pt_trace_end: call pt_trace_begin first");
    int rc = enqueue_end();
    if (rc) return rc;
    R.in_step = false;
    return collect_stats();
}

int pt_export_paths(pt_path_segment *host_paths, int capacity, int *n_live) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_export_paths: not initialised");
    uint32_t total = (uint32_t)R.map.tile_pixels * (uint32_t)std::max(1, R.step_count);
    uint32_t live = total;
    HIPCHK(hipStreamSynchronize(R.stream));
    if ((R.flags & PT_COMPACT) && !(R.flags & PT_FAKE_SHADER))
        HIPCHK(hipMemcpy(&live, &R.ctl->nlive[R.step_depth], 4, hipMemcpyDeviceToHost));
    const uint32_t n = (R.flags & PT_COMPACT) ? live : total;     // only the live prefix is meaningful after compaction
    if ((uint32_t)capacity < n) return fail(PT_ERR_INVALID, "pt_export_paths: capacity %d < %u", capacity, n);
    int rc = ensure_scratch((size_t)n * sizeof(pt_path_segment));
    if (rc) return rc;
    if (n) {
        uint32_t nprev = 0;
        const bool packed = (R.flags & PT_COMPACT) && R.cur_dir >= 0;
        if (packed) HIPCHK(hipMemcpy(&nprev, &R.ctl->nlive[R.cur_dir], 4, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(k_export_paths, dim3((n + 255) / 256), dim3(256), 0, R.stream, R.pool[R.cur], R.map, n,
                           live, R.trace_depth - R.step_depth, (pt_path_segment *)R.scratch,
                           tile_dir(packed ? R.cur_dir : -1), range_tiles(nprev, (uint32_t)R.grid * WAVES) * TILE);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(host_paths, R.scratch, (size_t)n * sizeof(pt_path_segment), hipMemcpyDeviceToHost, R.stream));
        HIPCHK(hipStreamSynchronize(R.stream));
    }
    if (n_live) *n_live = (int)live;
    return (int)n;
}

int pt_export_intersections(pt_shadeable_intersection *host_isects, uint8_t *host_outside, int capacity) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_export_intersections: not initialised");
    if (!(R.flags & (PT_UNFUSED | PT_SORT_MATERIAL | PT_FAKE_SHADER)))
        return fail(PT_ERR_INVALID, "pt_export_intersections: intersections are only materialised with PT_UNFUSED, PT_SORT_MATERIAL or PT_FAKE_SHADER");
    if (R.step_depth < 1) return fail(PT_ERR_INVALID, "pt_export_intersections: no bounce has run");
    uint32_t n = (uint32_t)R.map.tile_pixels * (uint32_t)std::max(1, R.step_count);
    HIPCHK(hipStreamSynchronize(R.stream));
    if ((R.flags & PT_COMPACT) && !(R.flags & PT_FAKE_SHADER))
        HIPCHK(hipMemcpy(&n, &R.ctl->nlive[R.step_depth - 1], 4, hipMemcpyDeviceToHost));
    if ((uint32_t)capacity < n) return fail(PT_ERR_INVALID, "pt_export_intersections: capacity %d < %u", capacity, n);
    int rc = ensure_scratch((size_t)n * (sizeof(pt_shadeable_intersection) + 1) + 64);
    if (rc) return rc;
    uint8_t *d_out = (uint8_t *)R.scratch + (size_t)n * sizeof(pt_shadeable_intersection);
    if (n) {
        hipLaunchKernelGGL(k_export_isects, dim3((n + 255) / 256), dim3(256), 0, R.stream,
                           R.sorted_isects ? Isect{R.isect2_mem, R.cap} : R.isect, n,
                           (pt_shadeable_intersection *)R.scratch, host_outside ? d_out : (uint8_t *)nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(host_isects, R.scratch, (size_t)n * sizeof(pt_shadeable_intersection), hipMemcpyDeviceToHost, R.stream));
        if (host_outside) HIPCHK(hipMemcpyAsync(host_outside, d_out, n, hipMemcpyDeviceToHost, R.stream));
        HIPCHK(hipStreamSynchronize(R.stream));
    }
    return (int)n;
}

int pt_intersect_once(const pt_path_segment *host_paths, int n, pt_shadeable_intersection *host_isects,
                      uint8_t *host_outside) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_intersect_once: not initialised");
    if (n < 0 || (uint32_t)n > R.cap) return fail(PT_ERR_INVALID, "pt_intersect_once: n=%d exceeds the pool capacity %u", n, R.cap);
    if (n == 0) return PT_OK;
    if (!host_paths || !host_isects) return fail(PT_ERR_INVALID, "pt_intersect_once: null buffer");
    int rc = ensure_scratch((size_t)n * (sizeof(pt_path_segment) + 1) + 64);
    if (rc) return rc;
    R.in_step = false;
    HIPCHK(hipMemcpyAsync(R.scratch, host_paths, (size_t)n * sizeof(pt_path_segment), hipMemcpyHostToDevice, R.stream));
    hipLaunchKernelGGL(k_import_paths, dim3((n + 255) / 256), dim3(256), 0, R.stream, R.pool[0],
                       (const pt_path_segment *)R.scratch, (uint32_t)n);
    HIPCHK(hipGetLastError());
    if (R.has_mesh) launch_intersect<true>(R.pool[0], nullptr, (uint32_t)n, tile_dir(-1), nullptr);
    else launch_intersect<false>(R.pool[0], nullptr, (uint32_t)n, tile_dir(-1), nullptr);
    HIPCHK(hipGetLastError());
    uint8_t *d_out = (uint8_t *)R.scratch + (size_t)n * sizeof(pt_shadeable_intersection);
    hipLaunchKernelGGL(k_export_isects, dim3((n + 255) / 256), dim3(256), 0, R.stream, R.isect, (uint32_t)n,
                       (pt_shadeable_intersection *)R.scratch, host_outside ? d_out : (uint8_t *)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_isects, R.scratch, (size_t)n * sizeof(pt_shadeable_intersection), hipMemcpyDeviceToHost, R.stream));
    if (host_outside) HIPCHK(hipMemcpyAsync(host_outside, d_out, n, hipMemcpyDeviceToHost, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));
    return PT_OK;
}

int pt_get_image(float *host_image_sum) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_get_image: not initialised");
    if (!host_image_sum) return fail(PT_ERR_INVALID, "pt_get_image: null buffer");
    HIPCHK(hipStreamSynchronize(R.stream));
    HIPCHK(hipMemcpy(host_image_sum, R.image, (size_t)R.npix * 12, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_tonemap(uint8_t *host_rgba, int iter) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_tonemap: not initialised");
    if (!host_rgba || iter < 1) return fail(PT_ERR_INVALID, "pt_tonemap: bad argument");
    int rc = ensure_scratch((size_t)R.npix * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tonemap, dim3((R.npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, (uint8_t *)R.scratch,
                       R.image, R.npix, iter);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_rgba, R.scratch, (size_t)R.npix * 4, hipMemcpyDeviceToHost, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));
    return PT_OK;
}

int pt_clear_image(void) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_clear_image: not initialised");
    HIPCHK(hipMemsetAsync(R.image, 0, (size_t)R.npix * 12, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));
    return PT_OK;
}

float *pt_device_image(void) { return R.live ? R.image : nullptr; }

long long pt_total_rays(void) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_total_rays: not initialised");
    Persist p;
    if (hipMemcpyAsync(&p, R.persist, sizeof p, hipMemcpyDeviceToHost, R.stream) != hipSuccess ||
        hipStreamSynchronize(R.stream) != hipSuccess)
        return fail(PT_ERR_DEVICE, "pt_total_rays: device read failed");
    return (long long)p.rays;
}

int pt_get_counters(int64_t *rays, int64_t *first_bounce_rays, int64_t *iterations) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_get_counters: not initialised");
    Persist p;
    HIPCHK(hipMemcpyAsync(&p, R.persist, sizeof p, hipMemcpyDeviceToHost, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));
    if (rays) *rays = (int64_t)p.rays;
    if (first_bounce_rays) *first_bounce_rays = (int64_t)p.first_rays;
    if (iterations) *iterations = (int64_t)p.iterations;
    return PT_OK;
}

int pt_set_profiling(int enable) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_set_profiling: not initialised");
    int rc = drain_events();
    if (rc) return rc;
    if (enable && R.ev.empty()) {
        R.ev.resize(2 * EV_PAIRS);
        R.ev_stage.assign(EV_PAIRS, 0);
        for (auto &e : R.ev) HIPCHK(hipEventCreate(&e));
    }
    R.profiling = enable != 0;
    R.prof = pt_profile{};
    return PT_OK;
}

int pt_get_profile(pt_profile *out) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_get_profile: not initialised");
    if (!out) return fail(PT_ERR_INVALID, "pt_get_profile: null");
    int rc = drain_events();
    if (rc) return rc;
    *out = R.prof;
    return PT_OK;
}

int pt_get_stats(pt_stats *stats) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_get_stats: not initialised");
    if (!stats) return fail(PT_ERR_INVALID, "pt_get_stats: null");
    *stats = R.stats;
    return PT_OK;
}

}  // extern "C"
