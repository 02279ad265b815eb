// ptmi355.hip -- libptmi355.so: kernels + C-ABI (include/ptmi355.h).
//
// MI355X-native replacement for the hot path of the reference's src/pathtrace.cu (pathtraceInit / pathtrace /
// pathtraceFree and its five kernels).  Design:
//   * path state lives in a pool that is SoA per 64-path tile (ten 256-B rows: ox..oz dx..dz cr..cb pid) and holds
//     `batch` iterations of the tile's pixels; two pools ping-pong;
//   * a persistent grid; every WAVE owns one contiguous run of 64-path tiles per bounce and walks it with two
//     tiles in flight: cull against per-primitive world boxes -> candidate ring in LDS -> lane-dense exact
//     object-space tests (64 candidates per pass) -> shade / scatter in registers (pt_kernels.hpp);
//   * stable compaction without cross-wave communication: survivors are ranked with a wave64 ballot and appended
//     to the front of the wave's own span; each wave publishes its count, the last workgroup out of the launch
//     scans the <= 8192 counts into bases (range directory) and the next bounce maps logical index -> slot;
//   * the live count stays on the device: the next bounce reads it from HBM, no host round trip inside a batch;
//   * terminated paths drop their final colour into final[sample][pixel] (one 16-B store); one gather kernel adds the
//     samples into the float3 accumulation buffer in iteration order (bit-identical to sequential iterations);
//   * small batches (the reference's one iteration per call) run all their bounces in ONE launch (k_iteration); with a
//     host image the kernel's waves write the new sums into the caller's device-mapped buffer themselves.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA contraction:
// parity with the reference arithmetic is bit-exact, tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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

// The SHIPPED library reads ten environment variables, each documented in include/ptmi355.h ("Environment") and each
// exercised by a -m gpu test.  Every other switch rounds 1-4 grew -- launch-plan, occupancy and transport experiments,
// test hooks -- exists only in a build with -DPT_EXPERIMENTS (profiles/tools/build_variant.sh; A/B tooling and the
// experiment tests load that build through PTMI355_LIB): a product whose point is bit-exactness does not change its
// launch plan because of a stray variable.
static inline const char *pt_experiment(const char *name) {
#ifdef PT_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#include "pt_types.hpp"
#include "pt_bvh.hpp"
#include "pt_cull.hpp"
#include "pt_kernels.hpp"

namespace {

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
constexpr size_t ERR_BYTES = 512;
char g_err[ERR_BYTES] = "";
// where fail() writes: the calling thread's buffer.  The host's thread uses g_err (pt_last_error); every worker thread
// of the multi-device layer (pt_multi.hpp) has its own, copied into g_err when its job fails.
thread_local char *t_err = g_err;

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, ERR_BYTES, fmt, ap);
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

constexpr int OV_MAX_LANES = 8;
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
    uint32_t *sort_table = nullptr;
    float *cache_mem = nullptr;   // first-bounce cache: 5 planes of tile_pixels (PT_CACHE_FIRST)
    bool cache_valid = false;
    Isect isect{};
    float *final_mem = nullptr;   // float4[cap]: {r, g, b, stamp} of the paths that ended with a non-zero colour, index = pid
    uint32_t fin_serial = 0;      // stamp of the current batch's entries (never 0; a wrap clears the buffer)
    float *image = nullptr;
    bool own_image = false;
    float *d_geoms = nullptr, *d_mats = nullptr, *d_tris = nullptr;
    float *d_cull = nullptr, *d_grec = nullptr;
    float *d_tri_bound = nullptr;  // every-triangle loop, stage 1: {centre, Rs^2} per triangle (upload_tri_bounds)
    size_t tri_bound_words = 0;
    uint32_t *d_ginfo = nullptr;
    double cull_eye_reach = 0.0;  // |camera position|_1 the cull boxes were made for
    std::vector<pt_geom> geoms_keep;   // host copies (pt_set_camera may have to remake the cull boxes)
    std::vector<pt_triangle> tris_keep;
    std::vector<pt_mesh> meshes_keep;
    std::vector<float> grec_keep;      // the geom records as uploaded (PT_MESH_BVH rewrites the meshes' words when the trees are rebuilt)
    bool scene_lds = true;        // gather records + materials staged in LDS (else read through the vector cache)
    SceneDev scene{};
    size_t lds_bytes = 0;
    Control *ctl = nullptr;
    Persist *persist = nullptr;
    uint32_t *iter_counts = nullptr;       // k_iteration: traced counts [bounce][workgroup] (BounceArgs::iter_counts); one per lane
    size_t iter_counts_bytes = 0;
    HostStats *h_stats = nullptr, *d_stats = nullptr;   // page-locked, device-mapped: the last workgroup of a synchronous call's k_iteration writes pt_stats' numbers here
    bool want_host_stats = false;          // this call ends in collect_stats (pt_trace / pt_trace_batch)
    uint32_t host_stats_serial = 0;        // != 0: the batch just enqueued leaves its counts in h_stats under this serial
    bool self_gathered = false;            // the batch just enqueued did finalGather inside k_iteration (no k_gather)
    uint32_t *dir_mem = nullptr;  // per bounce: count[Wp], base[Wp+4]
    size_t dir_stride = 0;        // words per bounce
    int cur_dir = -1;             // bounce whose directory describes pool[cur] (-1: dense)
    uint32_t max_tiles = 0;
    size_t flag_words = 0;                   // mesh pre-pass: 64-bit flag words per parity (one bit per physical pool slot)
    size_t ctl_bytes = 0;         // Control, zeroed per batch
    int grid = 0;                 // persistent grid size
    int grid_iter = 0;            // k_iteration's own (its register budget differs from the bounce kernels'): the co-resident maximum
    int grid_iter_cur = 0;        // ... and what the batch just enqueued was launched with (iter_grid_for)
    int iter_tpw = 4;             // under the lanes k_iteration's grid is sized for this many tiles per wave (0: always the whole grid; PTMI355_ITER_TPW) ...
    int iter_wgs_per_cu_all = 15; // ... but not below this many workgroups per CU over all lanes together (PTMI355_ITER_WGS_ALL)
    bool ov_lanes_set = false;    // PTMI355_OVERLAP named a lane count
    int ov_streams = 2;           // launch streams the lanes share (lane k uses stream k % ov_streams); PTMI355_LANE_STREAMS
    int cus = 0;
    int grid_sort = 0;            // workgroups of the material-sort kernels (k_sort_hist / k_shade_sorted)
    bool sort_wave = true;        // <= 64 keys: k_shade_sorted_w (PTMI355_SORT_WAVE=0 forces the workgroup-wide kernel)
    int sort_runs = 1;            // runs of tiles per wave of the fused sort (k_bounce); PTMI355_SORT_RUNS
    int sort_keys = 0;            // > 0: PT_SORT_MATERIAL in its fused form -- survivors placed by material, K = sort_keys ranges per wave (pt_types.hpp: RangeDir)
    bool sorted_isects = false;   // the last bounce was shaded in material order (the intersection planes keep the order the bounce received)
    bool gen_fused = false;       // bounce 0 of the current batch generates its own rays
    bool gen_sort = false;        // ... in the sorted pipeline (k_intersect + k_shade_sorted_w), no k_raygen either
    Lens lens{0, 0.0f, 0.0f};     // PT_AA_JITTER / thin lens (pt_scene_desc, pt_set_lens)
    // one captured graph per batch size: memset + every launch of a batch replayed with one hipGraphLaunch
    struct BatchGraph { hipGraphExec_t exec; int cur, cur_dir, step_depth; bool sorted_isects, gen_fused; };
    std::map<int, BatchGraph> graphs;
    uint64_t whole_max_paths = 6000000;  // batches up to this many paths run as ONE launch (k_iteration); PTMI355_WHOLE_MAX
    uint64_t whole_max_host_paths = 16000000;   // ... one iteration with a page-locked host image: up to this many (PTMI355_WHOLE_MAX_HOST)
    bool whole = false;           // the current batch did
    // Batches whose caller does not wait for them overlap on the device (enqueue_batch_direct): each runs on a LANE --
    // a launch stream of its own and its own set of the buffers a batch in flight owns
    struct Bufs {
        float *pool_mem[2]; Pool pool[2]; float *final_mem; Control *ctl; uint32_t *dir_mem;
        float4 *mesh_hit; unsigned long long *mesh_flags[2]; uint32_t *iter_counts;
    };
    struct Lane {
        hipStream_t stream = nullptr;
        Bufs b{};                                 // lane 0: the session's own
        hipEvent_t traced = nullptr, gathered = nullptr;
        bool gathered_valid = false;
    } lane[OV_MAX_LANES];
    Lane *lane_cur = nullptr;     // the lane whose buffers and stream currently stand in for the session's (while its batch is enqueued)
    hipStream_t lane_main = nullptr;   // ... and the session's launch stream meanwhile
    size_t pool_bytes = 0, final_bytes = 0, dir_bytes = 0, mesh_hit_bytes = 0;   // of one set (init_impl)
    int ov_lanes = 4;             // PTMI355_OVERLAP=n: n lanes (0: every batch on the launch stream); 4 measured best, 3 worst (profiles/r03/variants_overlap*.log)
    double ov_budget_gb = 64.0;   // PTMI355_OVERLAP_GB: HBM the extra lanes may take
    hipEvent_t ov_enter = nullptr;
    bool ov_ready = false;        // lanes allocated
    bool ov_enabled = true;
    bool ov_ok = false;           // this call does not wait for its own result (async entry points)
    bool ov_active = false;       // the last thing enqueued was an overlapped batch
    int ov_next = 0;
    Control *last_ctl = nullptr;  // the control block of the last batch (collect_stats)
    float *epi_host = nullptr;    // pt_trace: the caller's image, device-mapped, for k_iteration's own gather (this call only)
    bool epi_done = false;        // ... and k_iteration took it
    bool epi_direct_enabled = true;   // PTMI355_EPI_DIRECT=0: such launches keep the final-colour buffer and gather per wave at their end
    bool host_sparse_enabled = false; // PT_HOST_SPARSE (implied by PT_SHARED_IMAGE): only the pixels whose sum changed are written to a host image the launch wrote last
    uint64_t image_epoch = 0;     // bumped by everything that changes the accumulation buffer
    float *host_synced = nullptr; // the (device-mapped) host image that held exactly the buffer's content at epoch host_epoch
    uint64_t host_epoch = 0;
    bool epi_enabled = true;      // PTMI355_HOST_EPILOGUE=0: always copy after the iteration
    bool pin_enabled = true;      // PTMI355_PIN=0: never page-lock caller buffers (copies take the runtime's pageable path)
    bool use_graphs = false;      // PTMI355_GRAPH=1 turns replay on (measured slower than direct launches on ROCm 7.2: DESIGN.md 6.10)
    bool capturing = false;
    int mesh_mode = MESH_NONE;    // MESH_TILES: every triangle per ray; MESH_BVH: PT_MESH_BVH culling
    float *d_bvh_nodes = nullptr, *d_bvh_tris = nullptr, *d_bvh_top = nullptr;
    std::vector<float> mesh_grids;           // per mesh: lo xyz, hi xyz of its box grid (contains every box of its tree)
    unsigned long long *d_cam_mask = nullptr;   // bounce-0 tile mask (BounceArgs::cam_mask)
    bool cam_mask_valid = false;
    unsigned long long *d_cull0 = nullptr;      // bounce-0 candidate primitives per camera tile (BounceArgs::cull0)
    uint32_t cull0_tiles = 0;                   // 0: not applicable (> 64 primitives, tile_pixels not a multiple of 64, switched off)
    int4 *d_bvh_meshes = nullptr;
    float4 *mesh_hit = nullptr;              // mesh pre-pass results (k_mesh), one per pool slot
    unsigned long long *mesh_flags[2] = {nullptr, nullptr};   // one flag per pool slot: "mesh_hit[slot] is valid" (bounce parity)
    bool mesh_marked = false;                // the last bounce flagged the next bounce's mesh candidates
    int grid_mesh = 0;
    pt_bvh_info bvh_info{};
    // host buffers the caller hands to pt_trace (scene->state.image): page-locked once so that the per-call copy of
    // the running sum (pathtrace.cu:389-390) runs at PCIe speed instead of through the runtime's staging
    struct HostReg { void *ptr; size_t bytes; void *dev; };   // dev: the device's address of the mapping (looked up once)
    std::vector<HostReg> host_regs;
    // PT_ASYNC_IMAGE: snapshot of the running sum per call (device), copied out on a second stream while the next
    // call traces
    float *snap[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_snap[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    uint64_t async_calls = 0;
    // PT_ASYNC_IMAGE through the launch's own host writes (pt_trace, one launch per iteration): completion events of such
    // launches, the event the NEXT asynchronous call waits for before it returns (a copy's or a launch's), and the last
    // copy-engine transfer a launch that writes the host buffer itself has to come after
    hipEvent_t ev_direct[2] = {nullptr, nullptr};
    int direct_k = 0;
    hipEvent_t async_prev = nullptr, dma_last = nullptr;
    bool async_direct_enabled = true;     // PTMI355_ASYNC_DIRECT=0: always snapshot + copy engine
    unsigned int *dbg_counts = nullptr;   // PTMI355_DBG_COUNTS=<words>: buffer for an instrumented kernel build's block counts (BounceArgs::dbg_counts)
    size_t dbg_words = 0;
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
};

// One context = one device's renderer (the reference has one file-static set of buffers, pathtrace.cu:70-75).  A
// single-device session uses g_single on the caller's thread; a multi-device session (pt_multi.hpp) owns one context
// per device, each driven by its own host thread.  `R` is the context of the calling thread.
Renderer g_single;
thread_local Renderer *t_ctx = &g_single;
#define R (*t_ctx)

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

int ensure_isect(void) {
    if (R.isect_mem) return PT_OK;
    HIPCHK(hipMalloc(&R.isect_mem, (size_t)R.cap * 5 * 4));
    R.isect = Isect{R.isect_mem, R.cap};
    return PT_OK;
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
    const uint32_t W = (uint32_t)R.grid * WAVES * (uint32_t)(R.sort_keys > 0 ? R.sort_runs : 1);     // runs of tiles
    const uint32_t nr = W * (uint32_t)std::max(1, R.sort_keys);
    if (depth < 0) return RangeDir{nullptr, W, nr};
    return RangeDir{R.dir_mem + (size_t)depth * R.dir_stride, W, nr};
}

BounceArgs bounce_args(int depth) {
    BounceArgs a{};
    a.dbg_counts = R.dbg_counts;
    a.in = R.pool[R.cur];
    a.out = (R.flags & PT_COMPACT) ? R.pool[R.cur ^ 1] : R.pool[R.cur];
    a.isect = R.isect;
    a.scene = R.scene;
    a.map = R.map;
    a.ctl = R.ctl;
    a.dir_in = tile_dir((R.flags & PT_COMPACT) ? R.cur_dir : -1);
    a.dir_out = tile_dir(depth);
    a.fin = R.final_mem;
    a.fin_stamp = R.capturing ? 0u : R.fin_serial;
    a.cam = R.cam;
    a.lens = R.lens;
    a.depth = depth; a.trace_depth = R.trace_depth; a.iter0 = R.capturing ? -1 : R.step_iter0;
    a.pool_n = (uint32_t)R.map.tile_pixels * (uint32_t)R.step_count;
    a.gen_rays = (depth == 0 && R.gen_fused) ? 1 : 0;
    a.mesh_hit = R.mesh_hit;
    a.mesh_flags_in = R.mesh_flags[depth & 1]; a.mesh_flags_out = R.mesh_flags[(depth + 1) & 1];
    a.mesh_scan = R.mesh_marked ? 0 : 1;
    a.cam_mask = (R.cam_mask_valid && !(R.lens.radius > 0.0f)) ? R.d_cam_mask : nullptr;
    // the candidate masks describe the rays of a pinhole camera through pixel centres
    const bool same_rays = !R.lens.aa && !(R.lens.radius > 0.0f);
    a.cull0 = (R.cull0_tiles && same_rays) ? R.d_cull0 : nullptr;
    a.cull0_tiles = R.cull0_tiles;
    a.iter_counts = R.iter_counts;
    a.persist = R.persist;
    return a;
}

// every batch stamps the final colours it writes with a fresh serial number (put_final / k_gather); under graph replay
// the kernels read it from Control::keep[0]
int next_fin_stamp(void) {
    if (++R.fin_serial == 0) {                                    // 2^32 batches later: forget every old stamp
        HIPCHK(hipMemsetAsync(R.final_mem, 0, (size_t)R.cap * 16, R.stream));
        if (R.ov_ready)
            for (int j = 1; j < R.ov_lanes; ++j)
                HIPCHK(hipMemsetAsync(R.lane[j].b.final_mem, 0, R.final_bytes, R.stream));
        R.fin_serial = 1;
    }
    return PT_OK;
}

// `clear`: the per-batch clear of the control block (live counts, election counters).  A batch that runs as ONE launch
// (k_iteration) needs none: its counts are plain per-workgroup stores and its election puts its counters back itself.
int enqueue_begin(int iter0, int count, bool stepping, bool clear = true) {
    if (count < 1 || count > R.max_batch)
        return fail(PT_ERR_INVALID, "batch count %d outside [1, max_batch=%d]", count, R.max_batch);
    // makeSeededRandomEngine ORs the iteration into a word that holds the depth from bit 22 up (pathtrace.cu:41-45);
    // past 2^22 iterations the streams of different depths collide in the reference too -- reproduced, not refused
    if (iter0 < 0 || (int64_t)iter0 + count - 1 > 0x7fffffff)
        return fail(PT_ERR_INVALID, "iteration %d (+%d) outside [0, 2^31)", iter0, count);
    R.step_iter0 = iter0; R.step_count = count; R.step_depth = 0; R.cur = 0; R.cur_dir = -1;
    R.ov_active = false;          // (every overlapped batch's gather is on the launch stream: what follows is ordered after them)
    R.last_ctl = R.ctl;
    if (!R.capturing) { const int rc = next_fin_stamp(); if (rc) return rc; }
    R.sorted_isects = false;
    R.mesh_marked = false;
    R.self_gathered = false; R.host_stats_serial = 0;
    if (clear) HIPCHK(hipMemsetAsync(&R.ctl->stamp, 0, R.ctl_bytes, R.stream));      // everything but Control::iter0
    if (R.mesh_mode == MESH_BVH)
        for (int k = 0; k < 2; ++k)
            HIPCHK(hipMemsetAsync(R.mesh_flags[k], 0, R.flag_words * sizeof(unsigned long long), R.stream));
    // batch path: bounce 0 generates the camera rays itself (no 40 B/path round trip through HBM)
    R.gen_fused = !stepping && (!(R.flags & (PT_UNFUSED | PT_SORT_MATERIAL | PT_FAKE_SHADER)) || R.sort_keys > 0);
    // sorted batches of up to 64 keys: k_intersect and k_shade_sorted_w generate bounce 0's rays themselves
    R.gen_sort = !stepping && (R.flags & PT_SORT_MATERIAL) && !R.sort_keys && !(R.flags & PT_FAKE_SHADER) && R.sort_wave &&
                 R.scene.nmats + 1 <= SORTW_MAX_BINS;
    if (R.gen_fused || R.gen_sort) return PT_OK;
    const uint32_t total = (uint32_t)R.map.tile_pixels * (uint32_t)count;
    StageTimer tm(PT_STAGE_RAYGEN);
    hipLaunchKernelGGL(k_raygen, dim3((total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, R.pool[0], R.cam,
                       R.lens, R.map, count, R.capturing ? -1 : iter0, R.trace_depth, R.ctl);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

// the mesh mode and where the per-lane scene gathers come from (LDS / vector cache) are template switches of
// every kernel that intersects: pick the instantiation
#define PT_MESH_DISPATCH(CALL)                                          \
    do {                                                                \
        if (R.scene_lds) {                                              \
            constexpr bool SLDS = true;                                 \
            if (R.mesh_mode == MESH_BVH) { constexpr int MESH = MESH_BVH; CALL; }            \
            else if (R.mesh_mode == MESH_TILES) { constexpr int MESH = MESH_TILES; CALL; }   \
            else { constexpr int MESH = MESH_NONE; CALL; }              \
        } else {                                                        \
            constexpr bool SLDS = false;                                \
            if (R.mesh_mode == MESH_BVH) { constexpr int MESH = MESH_BVH; CALL; }            \
            else if (R.mesh_mode == MESH_TILES) { constexpr int MESH = MESH_TILES; CALL; }   \
            else { constexpr int MESH = MESH_NONE; CALL; }              \
        }                                                               \
    } while (0)

// `raygen_pool`: `in` is what k_raygen wrote for the current camera (bounce 0 of a batch or of the stepping interface)
// `generate`: bounce 0 of a sorted batch -- the kernel generates the camera rays itself (R.gen_sort), `in` is not read
void launch_intersect(const Pool &in, const uint32_t *n_ptr, uint32_t n_fixed, const RangeDir &dir,
                      const uint32_t *nprev, bool raygen_pool = false, bool generate = false) {
    const bool same_rays = !R.lens.aa && !(R.lens.radius > 0.0f);
    const unsigned long long *cull0 = (raygen_pool && R.cull0_tiles && same_rays) ? R.d_cull0 : nullptr;
    RayGen gen{};
    if (generate) {
        gen.cam = R.cam; gen.lens = R.lens; gen.map = R.map; gen.trace_depth = R.trace_depth;
        gen.iter0 = R.capturing ? -1 : R.step_iter0;
        PT_MESH_DISPATCH(hipLaunchKernelGGL((k_intersect<MESH, SLDS, true>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, in,
                                            R.isect, R.scene, n_ptr, n_fixed, dir, nprev, R.ctl, cull0, R.cull0_tiles, gen));
        return;
    }
    PT_MESH_DISPATCH(hipLaunchKernelGGL((k_intersect<MESH, SLDS>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, in,
                                        R.isect, R.scene, n_ptr, n_fixed, dir, nprev, R.ctl, cull0, R.cull0_tiles, gen));
}

// The instantiations of k_bounce that are ever launched: MODE_ISECT / MODE_CACHE0 intersect nothing (one mesh mode
// serves them all, no ray generation); the fused kernel reads the results of the mesh pre-pass under PT_MESH_BVH
// (the hierarchy is never walked inline by k_bounce) and generates bounce 0's rays itself in batches (GEN).
// the fused compacting kernel that launch_bounce_at picks for (scene in LDS, ray generation, material keys)
template <int MESH>
const void *bounce_fn(bool slds, bool gen, bool sorted) {
    if constexpr (MESH != MESH_PRE) {
        if (sorted) {
            if (slds) return gen ? (const void *)k_bounce<MODE_FUSED, true, MESH, true, true, true> : (const void *)k_bounce<MODE_FUSED, true, MESH, true, false, true>;
            return gen ? (const void *)k_bounce<MODE_FUSED, true, MESH, false, true, true> : (const void *)k_bounce<MODE_FUSED, true, MESH, false, false, true>;
        }
    }
    if (slds) return gen ? (const void *)k_bounce<MODE_FUSED, true, MESH, true, true> : (const void *)k_bounce<MODE_FUSED, true, MESH, true, false>;
    return gen ? (const void *)k_bounce<MODE_FUSED, true, MESH, false, true> : (const void *)k_bounce<MODE_FUSED, true, MESH, false, false>;
}
template <int MODE, bool COMPACT, int MESH, bool GEN>
void launch_bounce_at(const BounceArgs &a) {
    if constexpr (MODE == MODE_FUSED && COMPACT && MESH != MESH_PRE) {
        if (R.sort_keys > 0) {                                // PT_SORT_MATERIAL, fused: survivors placed by material
            if (R.scene_lds) hipLaunchKernelGGL((k_bounce<MODE, COMPACT, MESH, true, GEN, true>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, a);
            else hipLaunchKernelGGL((k_bounce<MODE, COMPACT, MESH, false, GEN, true>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, a);
            return;
        }
    }
    if (R.scene_lds) hipLaunchKernelGGL((k_bounce<MODE, COMPACT, MESH, true, GEN>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, a);
    else hipLaunchKernelGGL((k_bounce<MODE, COMPACT, MESH, false, GEN>), dim3(R.grid), dim3(BLOCK), R.lds_bytes, R.stream, a);
}
template <int MODE, bool COMPACT>
void launch_bounce(const BounceArgs &a) {
    if constexpr (MODE == MODE_ISECT) {
        launch_bounce_at<MODE, COMPACT, MESH_NONE, false>(a);
    } else if constexpr (MODE == MODE_CACHE0) {                  // bounce 0 by definition: batches generate their rays here too
        if (a.gen_rays) launch_bounce_at<MODE, COMPACT, MESH_NONE, true>(a); else launch_bounce_at<MODE, COMPACT, MESH_NONE, false>(a);
    } else {
        const bool gen = a.gen_rays != 0;
        if (R.mesh_mode == MESH_BVH) { if (gen) launch_bounce_at<MODE, COMPACT, MESH_PRE, true>(a); else launch_bounce_at<MODE, COMPACT, MESH_PRE, false>(a); }
        else if (R.mesh_mode == MESH_TILES) { if (gen) launch_bounce_at<MODE, COMPACT, MESH_TILES, true>(a); else launch_bounce_at<MODE, COMPACT, MESH_TILES, false>(a); }
        else { if (gen) launch_bounce_at<MODE, COMPACT, MESH_NONE, true>(a); else launch_bounce_at<MODE, COMPACT, MESH_NONE, false>(a); }
    }
}

int enqueue_bounce(int depth) {
    BounceArgs a = bounce_args(depth);
    const bool compact = (R.flags & PT_COMPACT) != 0;
    const bool sort2 = (R.flags & PT_SORT_MATERIAL) && R.sort_keys == 0;      // the two-kernel form of the sort
    const bool unfused = (R.flags & PT_UNFUSED) != 0 || sort2;
    if (unfused) {
        StageTimer tm(PT_STAGE_INTERSECT);
        const bool generate = depth == 0 && R.gen_sort;          // nobody has written nlive[0] yet: the pool size is a.pool_n
        const uint32_t *n_ptr = (compact && !generate) ? &R.ctl->nlive[depth] : (const uint32_t *)nullptr;
        const uint32_t *nprev = (compact && depth > 0) ? &R.ctl->nlive[depth - 1] : (const uint32_t *)nullptr;
        launch_intersect(a.in, n_ptr, a.pool_n, a.dir_in, nprev, depth == 0, generate);
        HIPCHK(hipGetLastError());
    }
    if (sort2) {
        // intersections of the (dense) pool -> per-workgroup key histogram + scan -> chunk-local counting sort fused
        // with shading: survivors land in the other pool in globally sorted, compacted order (pt_kernels.hpp)
        a.in = R.pool[R.cur]; a.out = R.pool[R.cur ^ 1];
        a.sort_table = R.sort_table; a.nbins = R.scene.nmats + 1;
        a.gen_rays = (depth == 0 && R.gen_sort) ? 1 : 0;         // k_shade_sorted_w generates bounce 0's rays as k_intersect did
        {
            StageTimer tm(PT_STAGE_SORT);
            const size_t lds = ((size_t)LDS_CTL_WORDS + (size_t)((a.nbins + 3) & ~3)) * 4;
            if (compact) hipLaunchKernelGGL(k_sort_hist<true>, dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
            else hipLaunchKernelGGL(k_sort_hist<false>, dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
            HIPCHK(hipGetLastError());
        }
        StageTimer tm(PT_STAGE_BOUNCE);
        if (a.nbins <= SORTW_MAX_BINS && R.sort_wave) {
            // up to 64 keys: wave-private sorting, one barrier per 512-path chunk (pt_kernels.hpp: k_shade_sorted_w)
            const size_t lds = shade_sorted_w_lds_words(R.scene.nmats) * 4;
            if (a.gen_rays) {
                if (compact) hipLaunchKernelGGL((k_shade_sorted_w<true, true>), dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
                else hipLaunchKernelGGL((k_shade_sorted_w<false, true>), dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
            } else {
                if (compact) hipLaunchKernelGGL((k_shade_sorted_w<true, false>), dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
                else hipLaunchKernelGGL((k_shade_sorted_w<false, false>), dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
            }
        } else {
            const size_t nb = (size_t)((a.nbins + 3) & ~3);
            const size_t lds = ((size_t)LDS_CTL_WORDS + (3 + WAVES) * nb + 2 * SORT_CHUNK +
                                (a.nbins <= 64 ? (size_t)R.scene.nmats * ptd::MAT_WORDS : 0)) * 4;
            if (compact) hipLaunchKernelGGL(k_shade_sorted<true>, dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
            else hipLaunchKernelGGL(k_shade_sorted<false>, dim3(R.grid_sort), dim3(BLOCK), lds, R.stream, a);
        }
        HIPCHK(hipGetLastError());
        R.cur ^= 1; R.cur_dir = -1;                      // the sorted pool is dense
        R.sorted_isects = true;
        R.step_depth = depth + 1;
        return PT_OK;
    }
    const bool cached0 = depth == 0 && !unfused && (R.flags & PT_CACHE_FIRST);
    if (cached0 && !R.cache_valid) {
        StageTimer tm(PT_STAGE_INTERSECT);
        const Isect cache{R.cache_mem, (uint32_t)R.map.tile_pixels};
        const int blocks = std::min(R.grid, (R.map.tile_pixels + BLOCK - 1) / BLOCK);
        PT_MESH_DISPATCH(hipLaunchKernelGGL((k_cache_first<MESH, SLDS>), dim3(blocks), dim3(BLOCK), R.lds_bytes, R.stream,
                                            cache, R.scene, R.cam, R.map));
        HIPCHK(hipGetLastError());
        R.cache_valid = true;
    }
    if (!cached0 && !unfused && R.mesh_mode == MESH_BVH) {
        StageTimer tm(PT_STAGE_MESH);
        if (compact) hipLaunchKernelGGL((k_mesh<true>), dim3(R.grid_mesh), dim3(MESH_BLOCK), MESH_LDS_BYTES, R.stream, a);
        else hipLaunchKernelGGL((k_mesh<false>), dim3(R.grid_mesh), dim3(MESH_BLOCK), MESH_LDS_BYTES, R.stream, a);
        HIPCHK(hipGetLastError());
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
    if (R.mesh_mode == MESH_BVH) {
        // this bounce's flags are spent; the array is the NEXT bounce's output flags.  Only a fused bounce marks
        // the candidates of the next one (the cached / unfused pipelines leave the finding to k_mesh's scan)
        HIPCHK(hipMemsetAsync(R.mesh_flags[depth & 1], 0, R.flag_words * sizeof(unsigned long long), R.stream));
        R.mesh_marked = !cached0 && !unfused;
    }
    if (compact) { R.cur ^= 1; R.cur_dir = depth; }
    R.step_depth = depth + 1;
    return PT_OK;
}

int enqueue_fake(void) {
    // the reference as shipped (pathtrace.cu:339-377): one bounce, fake shader
    const uint32_t total = (uint32_t)R.map.tile_pixels * (uint32_t)R.step_count;
    launch_intersect(R.pool[R.cur], nullptr, total, tile_dir(-1), nullptr);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_shade_fake, dim3((total + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, R.pool[R.cur],
                       R.isect, R.scene.mats, R.map, R.step_iter0, total, R.final_mem, R.fin_serial, R.ctl);
    HIPCHK(hipGetLastError());
    R.step_depth = 1;
    return PT_OK;
}

int enqueue_end(void) {
    if (R.self_gathered) { R.whole = false; return PT_OK; }   // finalGather and the counters were done inside k_iteration
    StageTimer tm(PT_STAGE_GATHER);
    hipStream_t gs = R.stream;
    if (R.lane_cur) {                                         // overlapped batch: gathers stay in call order on the launch stream
        HIPCHK(hipEventRecord(R.lane_cur->traced, R.stream));
        HIPCHK(hipStreamWaitEvent(R.lane_main, R.lane_cur->traced, 0));
        gs = R.lane_main;
    }
    hipLaunchKernelGGL(k_gather, dim3((R.map.tile_pixels + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, gs, R.image,
                       R.final_mem, R.cap, R.map,
                       R.step_count, R.ctl, R.persist, (R.flags & PT_FAKE_SHADER) ? 0 : R.trace_depth,
                       (R.flags & PT_FAKE_SHADER) ? (uint32_t)R.map.tile_pixels * (uint32_t)R.step_count : 0u,
                       R.whole ? 1 : 0, R.epi_done ? 1 : 0, R.capturing ? 0u : R.fin_serial, R.iter_counts, (uint32_t)R.grid_iter_cur,
                       (R.whole && R.host_stats_serial) ? R.d_stats : (HostStats *)nullptr);
    R.whole = false;
    R.image_epoch++;
    HIPCHK(hipGetLastError());
    if (R.lane_cur) {
        HIPCHK(hipEventRecord(R.lane_cur->gathered, gs));
        R.lane_cur->gathered_valid = true;
    }
    return PT_OK;
}

// Batches whose caller does not wait for them (pt_trace_batch_async) OVERLAP on the device.  A launch stream runs its kernels one after the other, and every kernel of this library ends
// with a tail: the persistent grid's waves do not finish together (mean residency 0.84-0.94 of a launch),
// and one iteration per launch (k_iteration) is a chain of `depth` dependent bounces per wave, ~10 us each at 800x800
// whatever the number of paths left.  Consecutive batches therefore go to different LANES -- each a set of pools,
// final-colour buffer, control block, directory and mesh pre-pass buffers, on one of TWO launch streams (lanes 0 and 2 on
// one, 1 and 3 on the other) -- and as the workgroups of one batch's kernel retire, those of another batch's take their
// slots.  Two streams, not one per lane: how many launches really run side by side is then this library's decision and
// not the runtime's -- it maps streams onto four hardware queues in creation order, kernels of streams that share a
// queue run one after the other, and with a stream per lane 1 spp per call measured anything between 15 and 34
// Grays/s depending on how many streams the process had created before (profiles/r04/ab_hw_queues*.log; four lanes
// on two streams: 29.4-30.2 in every combination tried).  A stream's second lane has its launch queued behind the first's
// while that one's gather is still to come.  What must stay ordered does: every k_gather
// runs on the session's launch stream, in call order, after its own batch's last kernel (event), so the image is
// summed in iteration order bit for bit and whatever is enqueued on the launch stream afterwards (tonemap, image
// copies, serial batches, pt_synchronize) comes after every batch before it; a lane's next batch waits for the gather
// of its previous one (it reuses the buffers that gather reads).  While a batch is enqueued its lane's buffers and
// stream stand in for the session's (put_bufs / R.stream), so the enqueue code is the serial one.
// Measured (profiles/r03/variants_overlap*.log), C2: 1 spp per call 20.0 -> 25.0 Grays/s, 8 spp 32.4 -> 38.8.
Renderer::Bufs take_bufs(void) {
    Renderer::Bufs b{};
    for (int k = 0; k < 2; ++k) { b.pool_mem[k] = R.pool_mem[k]; b.pool[k] = R.pool[k]; b.mesh_flags[k] = R.mesh_flags[k]; }
    b.final_mem = R.final_mem; b.ctl = R.ctl; b.dir_mem = R.dir_mem; b.mesh_hit = R.mesh_hit; b.iter_counts = R.iter_counts;
    return b;
}
void put_bufs(const Renderer::Bufs &b) {
    for (int k = 0; k < 2; ++k) { R.pool_mem[k] = b.pool_mem[k]; R.pool[k] = b.pool[k]; R.mesh_flags[k] = b.mesh_flags[k]; }
    R.final_mem = b.final_mem; R.ctl = b.ctl; R.dir_mem = b.dir_mem; R.mesh_hit = b.mesh_hit; R.iter_counts = b.iter_counts;
}

void free_lanes(void) {
    for (int k = 0; k < OV_MAX_LANES; ++k) {
        Renderer::Lane &l = R.lane[k];
        if (l.stream && k < R.ov_streams) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }   // lanes k, k + ov_streams, ... share one
        if (l.traced) (void)hipEventDestroy(l.traced);
        if (l.gathered) (void)hipEventDestroy(l.gathered);
        if (k > 0) {                                          // lane 0 borrows the session's own buffers
            for (int j = 0; j < 2; ++j) { if (l.b.pool_mem[j]) (void)hipFree(l.b.pool_mem[j]); if (l.b.mesh_flags[j]) (void)hipFree(l.b.mesh_flags[j]); }
            if (l.b.final_mem) (void)hipFree(l.b.final_mem);
            if (l.b.ctl) (void)hipFree(l.b.ctl);
            if (l.b.dir_mem) (void)hipFree(l.b.dir_mem);
            if (l.b.iter_counts) (void)hipFree(l.b.iter_counts);
            if (l.b.mesh_hit) (void)hipFree(l.b.mesh_hit);
        }
        l = Renderer::Lane{};
    }
    if (R.ov_enter) (void)hipEventDestroy(R.ov_enter);
    R.ov_enter = nullptr;
    R.ov_ready = false;
}

static int alloc_lanes(void) {
    R.lane[0].b = take_bufs();
    for (int j = 1; j < R.ov_lanes; ++j) {
        Renderer::Bufs &b = R.lane[j].b;
        for (int k = 0; k < 2; ++k) {
            HIPCHK(hipMalloc((void **)&b.pool_mem[k], R.pool_bytes));
            b.pool[k] = carve_pool(b.pool_mem[k], R.cap);
        }
        HIPCHK(hipMalloc((void **)&b.final_mem, R.final_bytes));
        HIPCHK(hipMemsetAsync(b.final_mem, 0, R.final_bytes, R.stream));
        HIPCHK(hipMalloc((void **)&b.ctl, sizeof(Control)));
        HIPCHK(hipMemsetAsync(b.ctl, 0, sizeof(Control), R.stream));
        HIPCHK(hipMalloc((void **)&b.dir_mem, R.dir_bytes));
        HIPCHK(hipMalloc((void **)&b.iter_counts, R.iter_counts_bytes));
        if (R.mesh_mode == MESH_BVH) {
            HIPCHK(hipMalloc((void **)&b.mesh_hit, R.mesh_hit_bytes));
            for (int k = 0; k < 2; ++k) {
                HIPCHK(hipMalloc((void **)&b.mesh_flags[k], R.flag_words * sizeof(unsigned long long)));
                HIPCHK(hipMemsetAsync(b.mesh_flags[k], 0, R.flag_words * sizeof(unsigned long long), R.stream));
            }
        }
    }
    R.ov_streams = std::max(1, std::min(R.ov_streams, R.ov_lanes));
    for (int k = 0; k < R.ov_lanes; ++k) {
        if (k < R.ov_streams) HIPCHK(hipStreamCreateWithFlags(&R.lane[k].stream, hipStreamNonBlocking));
        else R.lane[k].stream = R.lane[k % R.ov_streams].stream;
        HIPCHK(hipEventCreateWithFlags(&R.lane[k].traced, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&R.lane[k].gathered, hipEventDisableTiming));
    }
    HIPCHK(hipEventCreateWithFlags(&R.ov_enter, hipEventDisableTiming));
    return PT_OK;
}

// The lanes are an optimisation: when their buffers do not fit (the budget, or the device's free memory) or cannot be
// allocated, the session simply keeps tracing on its launch stream.
int ensure_lanes(void) {
    if (R.ov_ready || !R.ov_enabled) return PT_OK;
    const double per_lane = 2.0 * (double)R.pool_bytes + (double)R.final_bytes + (double)R.dir_bytes + (double)R.mesh_hit_bytes +
                            2.0 * (double)R.flag_words * 8.0 + (double)sizeof(Control);
    double budget = R.ov_budget_gb * 1e9;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, 0.5 * (double)free_b);   // leave room for the caller
    const int fit = 1 + (int)std::min(16.0, std::floor(budget / std::max(1.0, per_lane)));
    R.ov_lanes = std::min(R.ov_lanes, fit);
    if (R.ov_lanes == 3) R.ov_lanes = 2;                       // three lanes measured no better than one
    if (R.ov_lanes < 2) { R.ov_enabled = false; return PT_OK; }
    if (alloc_lanes() != PT_OK) {
        (void)hipGetLastError();
        free_lanes();
        R.ov_enabled = false;
        return PT_OK;
    }
    R.ov_ready = true;
    return PT_OK;
}

// the fused pipelines only: the unfused / two-kernel-sort / fake-shader ones keep intersection planes and sort tables
// (one set), the first-bounce cache is filled by the first batch that needs it
bool overlap_eligible(int count) {
    return R.ov_ok && R.ov_enabled && !R.capturing && !R.use_graphs && !R.profiling && !R.epi_host && !R.dbg_counts &&
           !(R.flags & (PT_UNFUSED | PT_FAKE_SHADER | PT_CACHE_FIRST)) && (!(R.flags & PT_SORT_MATERIAL) || R.sort_keys > 0) &&
           count >= 1 && count <= R.max_batch;
}

int enqueue_batch_serial(int iter0, int count);

int enqueue_batch_direct(int iter0, int count) {
    if (!overlap_eligible(count)) return enqueue_batch_serial(iter0, count);
    int rc = ensure_lanes();
    if (rc) return rc;
    if (!R.ov_enabled) return enqueue_batch_serial(iter0, count);       // the lanes do not fit the budget
    if (R.fin_serial == 0xffffffffu) {      // the stamp is about to wrap: nothing may be in flight while every lane's colours are forgotten
        HIPCHK(hipStreamSynchronize(R.stream));
        for (int j = 0; j < R.ov_lanes; ++j) HIPCHK(hipMemsetAsync(R.lane[j].b.final_mem, 0, R.final_bytes, R.stream));
        HIPCHK(hipStreamSynchronize(R.stream));
        R.fin_serial = 0;
        R.ov_active = false;
    }
    if (!R.ov_active) {
        // whatever the launch stream holds (uploads, masks, serial batches on the session's buffers) comes first
        HIPCHK(hipEventRecord(R.ov_enter, R.stream));
        for (int k = 0; k < R.ov_lanes; ++k) {
            HIPCHK(hipStreamWaitEvent(R.lane[k].stream, R.ov_enter, 0));
            R.lane[k].gathered_valid = false;
        }
    }
    Renderer::Lane &l = R.lane[R.ov_next];
    R.ov_next = (R.ov_next + 1) % R.ov_lanes;
    if (l.gathered_valid) HIPCHK(hipStreamWaitEvent(l.stream, l.gathered, 0));
    const Renderer::Bufs home = take_bufs();
    R.lane_main = R.stream; R.lane_cur = &l;
    put_bufs(l.b); R.stream = l.stream;
    rc = enqueue_batch_serial(iter0, count);                            // its gather goes to the launch stream (enqueue_end)
    R.stream = R.lane_main; put_bufs(home);
    R.lane_cur = nullptr; R.lane_main = nullptr;
    R.ov_active = rc == PT_OK;
    return rc;
}

// k_iteration's grid.  A launch of its own wants every co-resident workgroup (latency: 121 us at 800x800).  Under the
// lanes several launches share the device, and a workgroup of a full grid holds its slot for all eight bounces with two
// tiles per wave at bounce 0 and less than one from bounce 3 on.  Measured at 800x800 (profiles/r04/ab_iter_grid*.log,
// ab_lane_streams2.log): what counts is the workgroups the lanes ask for together -- best at ~15 per CU, three times
// what is co-resident, so that the slots turn over between the launches -- as long as a wave still has a few tiles
// (from 2 spp per call on the full grid is best again): 1 spp per call 28.6 -> 30.1 Grays/s.
int iter_grid_for(uint64_t paths, bool shared) {
    if (!shared || R.iter_tpw <= 0) return R.grid_iter;
    const uint64_t tiles = (paths + TILE - 1) / TILE;
    const uint64_t by_tiles = (tiles + (uint64_t)(WAVES * R.iter_tpw) - 1) / (uint64_t)(WAVES * R.iter_tpw);
    const uint64_t floor_g = ((uint64_t)R.iter_wgs_per_cu_all * (uint64_t)R.cus + (uint64_t)R.ov_lanes - 1) / (uint64_t)std::max(1, R.ov_lanes);
    return (int)std::min<uint64_t>(std::max(by_tiles, floor_g), (uint64_t)R.grid_iter);
}

int enqueue_batch_serial(int iter0, int count) {
    // small batch: every bounce in one launch (k_iteration)
    // (one iteration straight into a page-locked host image: the launch hides the PCIe transfer under its tracing, which
    // a kernel per bounce + a copy cannot: worth it for larger frames too -- 3840x2160: 2.49 -> see profiles/r04/ab_percall_4k.log)
    const uint64_t whole_limit = (count == 1 && R.epi_host) ? std::max(R.whole_max_paths, R.whole_max_host_paths) : R.whole_max_paths;
    const bool whole = !(R.flags & (PT_UNFUSED | PT_SORT_MATERIAL | PT_FAKE_SHADER | PT_CACHE_FIRST)) && (R.flags & PT_COMPACT) &&
                       R.mesh_mode == MESH_NONE && R.sort_keys == 0 && count >= 1 &&
                       (uint64_t)R.map.tile_pixels * (uint64_t)count <= whole_limit;
    int rc = enqueue_begin(iter0, count, false, !whole);
    if (rc) return rc;
    if (R.flags & PT_FAKE_SHADER) {
        rc = enqueue_fake();
        if (rc) return rc;
    } else if (whole) {
        StageTimer tm(PT_STAGE_BOUNCE);
        BounceArgs a = bounce_args(0);
        // One iteration, and no other launch of this session running beside it (the lanes of pt_trace_batch_async): the
        // workgroup that traces a pixel's path also does finalGather for it -- image[pixel] += colour inside the launch
        // (a second launch's waves adding to the same pixels at the same time would lose updates, and the order of the
        // float additions is part of the result) -- and, with a page-locked host image, writes the new sums there.
        if (count == 1 && !R.lane_cur && !R.capturing && !R.use_graphs && R.epi_enabled) {
            a.epi_image = R.image; a.epi_host = R.epi_host;
            // path by path (BounceArgs::epi_direct) when nothing but this library has written the accumulation buffer since
            // the host's copy was complete -- otherwise every pixel is written once more by the launch's epilogue.  A
            // caller-owned device buffer (pt_scene_desc.device_image) can change behind the library's back.
            const bool host_current = R.host_sparse_enabled && R.own_image && R.host_synced == R.epi_host && R.host_epoch == R.image_epoch;
            a.epi_direct = (R.epi_direct_enabled && (!R.epi_host || host_current)) ? 1 : 0;
            R.image_epoch++;
            if (R.epi_host) { R.epi_done = true; R.host_synced = R.epi_host; R.host_epoch = R.image_epoch; }
            R.self_gathered = true;
        }
        // a synchronous call's statistics go straight to page-locked host memory: written by whoever folds the counts, this
        // launch's last workgroup (own finalGather) or k_gather's first
        if (R.want_host_stats && !R.capturing && !R.use_graphs && R.d_stats) { a.host_stats = R.d_stats; R.host_stats_serial = R.fin_serial; }
        R.grid_iter_cur = iter_grid_for((uint64_t)R.map.tile_pixels * (uint64_t)count, R.lane_cur != nullptr);
        if (R.scene_lds) hipLaunchKernelGGL(k_iteration<true>, dim3(R.grid_iter_cur), dim3(BLOCK), R.lds_bytes, R.stream, a);
        else hipLaunchKernelGGL(k_iteration<false>, dim3(R.grid_iter_cur), dim3(BLOCK), R.lds_bytes, R.stream, a);
        HIPCHK(hipGetLastError());
        R.step_depth = R.trace_depth;
        R.whole = true;
    } else {
        for (int d = 0; d < R.trace_depth; ++d) {
            rc = enqueue_bounce(d);
            if (rc) return rc;
        }
    }
    return enqueue_end();
}

void drop_graphs(void) {
    for (auto &g : R.graphs) (void)hipGraphExecDestroy(g.second.exec);
    R.graphs.clear();
}

// A batch is the same sequence of launches every time (per-batch clear, ray generation, one kernel
// per bounce, gather) and differs only in its first iteration number, so it can be captured once per
// batch size and replayed with a single hipGraphLaunch; the iteration number travels through
// Control::iter0, written on the stream ahead of the graph.  Anything that changes a frozen launch
// argument (camera, lens, trace depth) drops the captured graphs.  Opt-in (PTMI355_GRAPH=1): on
// ROCm 7.2 / MI355X replay measured 4.5 % SLOWER than the ten direct launches at 1 spp per call
// (0.240 vs 0.230 ms) and 0.5 % slower at 16 spp, so direct launches stay the default.
int enqueue_batch(int iter0, int count) {
    const bool graphable = R.use_graphs && !R.profiling && !(R.flags & PT_FAKE_SHADER) &&
                           !((R.flags & PT_CACHE_FIRST) && !R.cache_valid);
    if (!graphable) return enqueue_batch_direct(iter0, count);
    if (count < 1 || count > R.max_batch)
        return fail(PT_ERR_INVALID, "batch count %d outside [1, max_batch=%d]", count, R.max_batch);
    // makeSeededRandomEngine ORs the iteration into a word that holds the depth from bit 22 up (pathtrace.cu:41-45);
    // past 2^22 iterations the streams of different depths collide in the reference too -- reproduced, not refused
    if (iter0 < 0 || (int64_t)iter0 + count - 1 > 0x7fffffff)
        return fail(PT_ERR_INVALID, "iteration %d (+%d) outside [0, 2^31)", iter0, count);
    auto it = R.graphs.find(count);
    if (it == R.graphs.end()) {
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(R.stream, hipStreamCaptureModeRelaxed));
        R.capturing = true;
        const int rc = enqueue_batch_direct(iter0, count);
        R.capturing = false;
        const hipError_t ce = hipStreamEndCapture(R.stream, &graph);
        if (rc != PT_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ce != hipSuccess || !graph) return fail(PT_ERR_DEVICE, "hipStreamEndCapture: %s", hipGetErrorString(ce));
        Renderer::BatchGraph g{};
        const hipError_t ie = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) return fail(PT_ERR_DEVICE, "hipGraphInstantiate: %s", hipGetErrorString(ie));
        g.cur = R.cur; g.cur_dir = R.cur_dir; g.step_depth = R.step_depth;
        g.sorted_isects = R.sorted_isects; g.gen_fused = R.gen_fused;
        it = R.graphs.emplace(count, g).first;
    }
    const Renderer::BatchGraph &g = it->second;
    {
        const int rc = next_fin_stamp();
        if (rc) return rc;
    }
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&R.ctl->iter0, iter0, 1, R.stream));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&R.ctl->keep[0], (int)R.fin_serial, 1, R.stream));
    HIPCHK(hipGraphLaunch(g.exec, R.stream));
    R.step_iter0 = iter0; R.step_count = count;
    R.cur = g.cur; R.cur_dir = g.cur_dir; R.step_depth = g.step_depth;
    R.sorted_isects = g.sorted_isects; R.gen_fused = g.gen_fused;
    return PT_OK;
}

int upload_tri_bounds(const pt_scene_desc *d, double Rorigin);

// per-primitive cull boxes (pt_cull.hpp) for the scene of R.desc as seen from camera `cam`: the |origin|_1 bound
// they are derived for covers the scene and the camera; a camera that later moves beyond it gets new boxes
int upload_cull(const pt_scene_desc *d, const pt_camera &cam) {
    const int n = d->num_geoms;
    std::vector<const float *> inv((size_t)std::max(1, n));
    std::vector<char> sph((size_t)std::max(1, n)), skip((size_t)std::max(1, n));
    for (int i = 0; i < n; ++i) {
        inv[(size_t)i] = &d->geoms[i].inverseTransform.m[0][0];
        sph[(size_t)i] = d->geoms[i].type == PT_SPHERE;
        skip[(size_t)i] = d->geoms[i].type == PT_TRIANGLE_MESH;
    }
    const double eye[3] = {(double)cam.position.x, (double)cam.position.y, (double)cam.position.z};
    std::vector<ptcull::Box> boxes;
    std::vector<double> pts(eye, eye + 3);
    // triangle meshes are world-space soups: their vertices bound where rays can start as well
    for (int t = 0; t < d->num_triangles; ++t) {
        const pt_vec3 *v = &d->triangles[t].v0;
        double m = 0.0;
        for (int k = 0; k < 3; ++k) m = std::max(m, (double)std::fabs(v[k].x) + std::fabs(v[k].y) + std::fabs(v[k].z));
        if (t == 0 || m > pts[3]) { if (pts.size() < 6) pts.resize(6, 0.0); pts[3] = m; pts[4] = 0.0; pts[5] = 0.0; }
    }
    R.scene.rmax = ptcull::make_boxes(inv.data(), reinterpret_cast<const bool *>(sph.data()),
                                      reinterpret_cast<const bool *>(skip.data()), n, pts.data(), (int)(pts.size() / 3), boxes);
    std::vector<float> rec((size_t)std::max(1, n) * CULL_WORDS, 0.0f);
    for (int i = 0; i < n; ++i) {
        float *r = rec.data() + (size_t)i * CULL_WORDS;
        for (int k = 0; k < 3; ++k) ptcull::centre_half(boxes[(size_t)i].lo[k], boxes[(size_t)i].hi[k], r[2 * k], r[2 * k + 1]);
        int ax = 3;
        if (d->geoms[i].type == PT_CUBE && !pt_experiment("PTMI355_NO_AXIS_REJECT"))
            ax = ptcull::reject_row(&d->geoms[i].inverseTransform.m[0][0], &r[7]);       // words 7..10: the row
        if (ax == 4 && pt_experiment("PTMI355_NO_ROW_REJECT")) ax = 3;
        const int tw = d->geoms[i].type | (ax << 8);
        memcpy(&r[6], &tw, 4);
    }
    if (!R.d_cull) HIPCHK(hipMalloc(&R.d_cull, rec.size() * 4));
    HIPCHK(hipMemcpyAsync(R.d_cull, rec.data(), rec.size() * 4, hipMemcpyHostToDevice, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));            // `rec` is pageable host memory about to go out of scope
    R.scene.cull = R.d_cull;
    R.cull_eye_reach = std::fabs(eye[0]) + std::fabs(eye[1]) + std::fabs(eye[2]);
    return upload_tri_bounds(d, (double)R.scene.rmax);       // the bounding spheres hold for origins within the same bound
}

// Every-triangle loop (MESH_TILES), stage 1 (pt_kernels.hpp: mesh_sweep): per triangle a sphere {c, Rs} such that a
// ray whose line passes c at more than Rs cannot be ACCEPTED for the triangle.  Derivation (u = 2^-24):
//   * the spec accepts a hit only if the point it reports, P_k = fl(o_k + fl(d_k tz)), lies in the triangle's box
//     [fl(lo_k - pad), fl(hi_k + pad)] (tri_point_ok, evaluated on these very floats): |P - c| <= R0 = half diagonal
//     of that box, c its centre;
//   * P_k differs from the line's point o_k + d_k tz by at most u |d_k tz| + u |P_k| <= 2u (|o_k| + |P_k|): the line
//     passes P within delta = 2 sqrt3 u (R + |c|_inf + R0)  (non-wild rays: |o|_1 <= R);
//   * the kernel's q = c x d' - fl(o x d') (fused multiply-adds, d' = d / |d| up to 2^-20) is off by at most
//     2^-21 (|o|_inf + |c|_inf) |d'|_inf per component, and its |q|^2 and the compare lose another 2^-20 relative;
//     rounding c to float moves it by u |c|_inf.
//   Rs = (R0 + 4 (delta + 2^-19 (R + |c|_inf + R0))) (1 + 2^-10), squared and rounded up.  Non-finite triangles get
//   Rs^2 = +inf (always a candidate: the exact test decides, and it never accepts them).  Each mesh's entries are
//   padded to a multiple of four with Rs^2 = -1 (no ray is a candidate: |q|^2 > -1).
// one mesh: `count` triangles -> ((count + 3) & ~3) x {cx, cy, cz, Rs^2}
void make_tri_bounds(const pt_triangle *tris, int count, double Rorigin, float *out) {
    const double u = 0x1p-24;
    const float pad = ptbvh::spec_pad(reinterpret_cast<const float *>(tris), count);
    const int n4 = (count + 3) & ~3;
    for (int i = 0; i < n4; ++i) {
        float *o = out + (size_t)i * 4;
        if (i >= count) { o[0] = o[1] = o[2] = 0.0f; o[3] = -1.0f; continue; }
        const pt_triangle &t = tris[i];
        const float v0[3] = {t.v0.x, t.v0.y, t.v0.z};
        const float e1[3] = {t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z};      // the device record's e1, e2
        const float e2[3] = {t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z};
        double c[3], h2 = 0.0, cinf = 0.0;
        bool fin = true;
        for (int a = 0; a < 3; ++a) {
            const float x1 = v0[a] + e1[a], x2 = v0[a] + e2[a];                       // tri_point_ok's own floats
            const float lo = std::fmin(v0[a], std::fmin(x1, x2)) - pad, hi = std::fmax(v0[a], std::fmax(x1, x2)) + pad;
            if (!std::isfinite(lo) || !std::isfinite(hi)) fin = false;
            c[a] = 0.5 * ((double)lo + (double)hi);
            const double h = 0.5 * ((double)hi - (double)lo);
            h2 += h * h;
            cinf = std::fmax(cinf, std::fabs(c[a]));
        }
        if (!fin || !std::isfinite(Rorigin)) { o[0] = o[1] = o[2] = 0.0f; o[3] = INFINITY; continue; }
        const double R0 = std::sqrt(h2);
        const double reach = Rorigin + cinf + R0;
        const double delta = 2.0 * 1.7320508075688772 * u * reach;
        const double Rs = (R0 + 4.0 * (delta + 0x1p-19 * reach)) * (1.0 + 0x1p-10);
        for (int a = 0; a < 3; ++a) o[a] = (float)c[a];
        o[3] = ptcull::round_up(Rs * Rs);
        if (!std::isfinite(o[3])) o[3] = INFINITY;
    }
}

int upload_tri_bounds(const pt_scene_desc *d, double Rorigin) {
    if (R.mesh_mode != MESH_TILES || d->num_meshes <= 0) return PT_OK;
    size_t words = 0;
    for (int k = 0; k < d->num_meshes; ++k) words += (size_t)((d->meshes[k].triangle_count + 63) & ~63) * 4;
    std::vector<float> tb(std::max<size_t>(words, 256), 0.0f);
    size_t off = 0;
    for (int k = 0; k < d->num_meshes; ++k) {
        const pt_mesh &m = d->meshes[k];
        make_tri_bounds(d->triangles + m.first_triangle, m.triangle_count, Rorigin, tb.data() + off);
        const size_t n4 = (size_t)((m.triangle_count + 3) & ~3), n64 = (size_t)((m.triangle_count + 63) & ~63);
        for (size_t i = n4; i < n64; ++i) { float *o = tb.data() + off + i * 4; o[0] = o[1] = o[2] = 0.0f; o[3] = -1.0f; }
        off += n64 * 4;
    }
    if (!R.d_tri_bound || R.tri_bound_words < tb.size()) {
        if (R.d_tri_bound) { HIPCHK(hipStreamSynchronize(R.stream)); (void)hipFree(R.d_tri_bound); R.d_tri_bound = nullptr; }
        HIPCHK(hipMalloc(&R.d_tri_bound, tb.size() * 4));
        R.tri_bound_words = tb.size();
    }
    HIPCHK(hipMemcpyAsync(R.d_tri_bound, tb.data(), tb.size() * 4, hipMemcpyHostToDevice, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));            // `tb` is pageable host memory about to go out of scope
    R.scene.tri_bound = R.d_tri_bound;
    return PT_OK;
}

// page-lock a caller-owned host buffer (idempotent per pointer; failures are not errors: the copy then takes the
// runtime's pageable path).  Returns true when [ptr, ptr + bytes) is registered BY THIS LIBRARY right now.  A recorded
// registration that overlaps the new range without being it belongs to a buffer the caller has since freed (the
// allocator handed part of its pages to this one): it is dropped first -- a stale registration would make
// hipHostRegister fail for the new buffer while hipHostGetDevicePointer / the runtime's copy path still resolve the
// new address through the old mapping, which ends where the OLD buffer ended (a GPU page fault past it).
// Only buffers of 1 MiB and more are page-locked: the allocator gives those their own mapping (whole pages that belong
// to nothing else).  Smaller ones share their pages with the caller's other heap objects; registering and later
// unregistering such pages left the runtime's copy path with stale ideas about them -- device-to-host copies into
// OTHER small buffers on the same pages ended in GPU page faults ("Memory access fault", found by the full GPU test
// suite) -- and at that size the pageable path costs nothing that matters.
bool pin_host(void *ptr, size_t bytes) {
    // only on the caller's word that the buffer outlives the session (PT_PIN_IMAGE / PT_ASYNC_IMAGE): a registration
    // cannot be re-validated -- a buffer freed and reallocated at the same address looks exactly like the old one to
    // the runtime while the device mapping still points at the old (pinned) pages
    if (!R.pin_enabled || !(R.flags & (PT_PIN_IMAGE | PT_ASYNC_IMAGE | PT_SHARED_IMAGE)) || bytes < ((size_t)1 << 20)) return false;
    const char *lo = (const char *)ptr, *hi = lo + bytes;
    for (size_t k = 0; k < R.host_regs.size();) {
        auto &h = R.host_regs[k];
        const char *hlo = (const char *)h.ptr, *hhi = hlo + h.bytes;
        if (h.ptr == ptr && h.bytes >= bytes) return true;
        if (hlo < hi && lo < hhi) {                       // overlaps (or the same start, too short): stale
            (void)hipHostUnregister(h.ptr);
            R.host_regs.erase(R.host_regs.begin() + (long)k);
            continue;
        }
        ++k;
    }
    if (R.host_regs.size() >= 4) {                    // a host that keeps handing over new buffers: forget the oldest
        (void)hipHostUnregister(R.host_regs.front().ptr);
        R.host_regs.erase(R.host_regs.begin());
    }
    if (hipHostRegister(ptr, bytes, hipHostRegisterMapped) == hipSuccess) { R.host_regs.push_back({ptr, bytes, nullptr}); return true; }
    (void)hipGetLastError();
    return false;
}

// the device's address of a page-locked host buffer (nullptr: not registered by us or not mappable -- the caller falls
// back to a copy)
float *map_host(float *host, size_t bytes) {
    if (!pin_host(host, bytes)) return nullptr;
    for (auto &h : R.host_regs)
        if (h.ptr == host) {
            if (!h.dev && hipHostGetDevicePointer(&h.dev, host, 0) != hipSuccess) { (void)hipGetLastError(); h.dev = nullptr; }
            return (float *)h.dev;
        }
    return nullptr;
}

// PT_ASYNC_IMAGE: the running sum after this call is snapshotted on the launch stream (device to device, microseconds)
// and copied to the host on a second stream while the NEXT call traces; `host` is complete when the next
// pt_trace / pt_trace_batch returns, or after pt_synchronize / pt_get_image / pt_free.
int enqueue_async_image(float *host) {
    const size_t bytes = (size_t)R.npix * 12;
    const int k = (int)(R.async_calls & 1);
    if (!R.copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&R.copy_stream, hipStreamNonBlocking));
        for (int j = 0; j < 2; ++j) {
            HIPCHK(hipMalloc(&R.snap[j], bytes));
            HIPCHK(hipEventCreateWithFlags(&R.ev_snap[j], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&R.ev_copied[j], hipEventDisableTiming));
        }
    }
    pin_host(host, bytes);
    if (R.async_calls >= 2) HIPCHK(hipStreamWaitEvent(R.stream, R.ev_copied[k], 0));     // the copy that last read snap[k]
    HIPCHK(hipMemcpyAsync(R.snap[k], R.image, bytes, hipMemcpyDeviceToDevice, R.stream));
    HIPCHK(hipEventRecord(R.ev_snap[k], R.stream));
    HIPCHK(hipStreamWaitEvent(R.copy_stream, R.ev_snap[k], 0));
    // The copy engine moves the 7.68 MB of an 800x800 frame in ~0.15 ms beside the next call's tracing (15.9 Grays/s
    // PCIe-inclusive).  PTMI355_ASYNC_COPY_WGS=n hands the snapshot over through n workgroups that store into the buffer's
    // device mapping instead (as k_iteration's epilogue does for synchronous calls): measured slower -- 64 workgroups
    // 0.25 ms (profiles/r04/ab_async_copy.log) -- and kept as an experiment switch only.
    const int copy_wgs = pt_experiment("PTMI355_ASYNC_COPY_WGS") ? atoi(pt_experiment("PTMI355_ASYNC_COPY_WGS")) : 0;
    float *mapped = copy_wgs > 0 ? map_host(host, bytes) : nullptr;
    if (mapped && ((uintptr_t)mapped & 15u) == 0) {
        hipLaunchKernelGGL(k_copy_out, dim3((unsigned)copy_wgs), dim3(BLOCK), 0, R.copy_stream, reinterpret_cast<float4 *>(mapped),
                           reinterpret_cast<const float4 *>(R.snap[k]), (uint32_t)(bytes / 16), mapped + (bytes / 16) * 4, R.snap[k] + (bytes / 16) * 4,
                           (uint32_t)((bytes % 16) / 4));
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(host, R.snap[k], bytes, hipMemcpyDeviceToHost, R.copy_stream));
    }
    HIPCHK(hipEventRecord(R.ev_copied[k], R.copy_stream));
    // the buffer handed over by the PREVIOUS call is complete when this call returns (its copy has been running
    // beside this call's tracing, which is already enqueued)
    if (R.async_prev) HIPCHK(hipEventSynchronize(R.async_prev));
    R.async_prev = R.ev_copied[k]; R.dma_last = R.ev_copied[k];
    R.async_calls++;
    // once this copy has landed the buffer holds the sum as of now: a later launch that writes the host itself (after
    // dma_last) only has to write what changes
    R.host_synced = R.own_image ? map_host(host, bytes) : nullptr; R.host_epoch = R.image_epoch;
    return PT_OK;
}

// the synchronous copy of the running sum (the reference's semantics): on the launch stream, into a pinned buffer
int enqueue_image_copy(float *host) {
    const size_t bytes = (size_t)R.npix * 12;
    pin_host(host, bytes);
    HIPCHK(hipMemcpyAsync(host, R.image, bytes, hipMemcpyDeviceToHost, R.stream));
    return PT_OK;
}

// reads the control block back (after a sync) and folds it into the stats
int collect_stats(void) {
    Control c;
    if (R.host_stats_serial) {
        // the launch's last workgroup wrote the counts into page-locked host memory: nothing to copy
        HIPCHK(hipStreamSynchronize(R.stream));
        if (R.h_stats->serial != R.host_stats_serial)
            return fail(PT_ERR_INTERNAL, "k_iteration left no statistics (serial %u, expected %u)", R.h_stats->serial, R.host_stats_serial);
        memset(&c, 0, offsetof(Control, bucket));
        memcpy(c.alive, R.h_stats->alive, sizeof c.alive);
        c.error = R.h_stats->error;
        R.host_stats_serial = 0;
    } else {
        HIPCHK(hipMemcpyAsync(&c, R.last_ctl ? R.last_ctl : R.ctl, offsetof(Control, bucket), hipMemcpyDeviceToHost, R.stream));
        HIPCHK(hipStreamSynchronize(R.stream));
    }
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
#ifdef PT_MESH_STATS
    fprintf(stderr, "[ptmi355] mesh pre-pass since init: %u candidates, %u lane-steps, %u wave-steps (density %.1f lanes)\n",
            c.keep[0], c.keep[1], c.keep[2], c.keep[2] ? (double)c.keep[1] / c.keep[2] : 0.0);
    {
        float f[7]; memcpy(f, &c.keep[4], sizeof f);
        fprintf(stderr, "[ptmi355] per walk-loop step (%u in all, incl. steps where nobody walks): %.1f lanes waiting for queued triangles, %.1f lanes without a walk\n",
                c.keep[10], c.keep[10] ? (double)c.keep[8] / c.keep[10] : 0.0, c.keep[10] ? (double)c.keep[9] / c.keep[10] : 0.0);
        fprintf(stderr, "[ptmi355] longest walk %u records; records where nothing was hit: %u, with a leaf hit: %u\n", c.keep[14], c.keep[12], c.keep[13]);
        fprintf(stderr, "[ptmi355] walks past 5000 steps: %u; last: o=(%.9g %.9g %.9g) d=(%.9g %.9g %.9g) tz=%g depth %u\n", c.keep[3],
                f[0], f[1], f[2], f[3], f[4], f[5], f[6], c.keep[11]);
        unsigned long long ms[32] = {0};
        (void)hipMemcpyFromSymbol(ms, HIP_SYMBOL(g_mesh_stats), sizeof ms);
        fprintf(stderr, "[ptmi355] mesh since load: %llu walks, %.2f records each; by length 1 | 2-3 | 4-7 | 8-15 | 16-31 | 32-63 | 64-127 | 128-255 | 256+:", ms[9], ms[9] ? (double)ms[10] / ms[9] : 0.0);
        for (int k = 0; k < 9; ++k) fprintf(stderr, " %llu", ms[k]);
        fprintf(stderr, "\n[ptmi355] wave-steps by walking lanes 1-8 | 9-16 | ... | 57-64:");
        for (int k = 11; k < 19; ++k) fprintf(stderr, " %llu", ms[k]);
        fprintf(stderr, "; nobody: %llu\n", ms[19]);
        fprintf(stderr, "[ptmi355] flagged form: %llu loop iterations, %llu appended a batch (%llu candidates: %.1f each); flag words %llu, non-zero %llu; "
                        "waves %llu (%.1f walks, %.1f loop iterations each); triangle passes %llu at %.1f lanes\n",
                ms[20], ms[21], ms[22], ms[21] ? (double)ms[22] / ms[21] : 0.0, ms[23], ms[24], ms[25], ms[25] ? (double)ms[9] / ms[25] : 0.0,
                ms[25] ? (double)ms[20] / ms[25] : 0.0, ms[26], ms[26] ? (double)ms[27] / ms[26] : 0.0);
    }
#endif
#ifdef PT_CULL_STATS
    {
        unsigned long long st[8] = {0};
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_cull_stats), sizeof st);
        fprintf(stderr, "[ptmi355] cull since load: %llu tiles, %llu active paths (%.1f per tile), %llu wild, %llu candidates (%.3f per path), "
                        "%llu passes (%.3f per tile, %.1f lanes each), %llu hits (%.3f per path)\n",
                st[0], st[6], st[0] ? (double)st[6] / st[0] : 0.0, st[5], st[1], st[6] ? (double)st[1] / st[6] : 0.0, st[2],
                st[0] ? (double)st[2] / st[0] : 0.0, st[2] ? (double)st[3] / st[2] : 0.0, st[4], st[6] ? (double)st[4] / st[6] : 0.0);
    }
#endif
    if (pt_experiment("PTMI355_DEBUG_SCAN")) {
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
// the entry points of ONE context (the C-ABI of include/ptmi355.h, applied to the calling thread's context `R`);
// the exported symbols are defined in pt_multi.hpp, which forwards to these directly (one device) or through the
// per-device worker threads (several)
// ===========================================================================
namespace one {

const char *pt_last_error(void) { return g_err; }
const char *pt_version(void) {
#ifdef PT_EXPERIMENTS
    return "ptmi355 0.1 (gfx950, fp32 no-contract, wave64) +experiments";
#else
    return "ptmi355 0.1 (gfx950, fp32 no-contract, wave64)";
#endif
}

void pt_free(void) {
    if (!R.live && !R.scratch) return;
    if (R.stream) (void)hipStreamSynchronize(R.stream);
    for (int k = 0; k < 2; ++k) if (R.pool_mem[k]) (void)hipFree(R.pool_mem[k]);
    if (R.isect_mem) (void)hipFree(R.isect_mem);
    if (R.sort_table) (void)hipFree(R.sort_table);
    if (R.cache_mem) (void)hipFree(R.cache_mem);
    if (R.final_mem) (void)hipFree(R.final_mem);
    if (R.image && R.own_image) (void)hipFree(R.image);
    if (R.d_geoms) (void)hipFree(R.d_geoms);
    if (R.d_mats) (void)hipFree(R.d_mats);
    if (R.d_tris) (void)hipFree(R.d_tris);
    if (R.d_cull) (void)hipFree(R.d_cull);
    if (R.d_grec) (void)hipFree(R.d_grec);
    if (R.d_tri_bound) (void)hipFree(R.d_tri_bound);
    if (R.d_ginfo) (void)hipFree(R.d_ginfo);
    drop_graphs();
    if (R.mesh_hit) (void)hipFree(R.mesh_hit);
    for (int k = 0; k < 2; ++k) if (R.mesh_flags[k]) (void)hipFree(R.mesh_flags[k]);
    if (R.d_bvh_nodes) (void)hipFree(R.d_bvh_nodes);
    if (R.d_bvh_meshes) (void)hipFree(R.d_bvh_meshes);
    if (R.d_bvh_tris) (void)hipFree(R.d_bvh_tris);
    if (R.d_bvh_top) (void)hipFree(R.d_bvh_top);
    if (R.d_cam_mask) (void)hipFree(R.d_cam_mask);
    R.d_cam_mask = nullptr; R.cam_mask_valid = false;
    if (R.d_cull0) (void)hipFree(R.d_cull0);
    R.d_cull0 = nullptr; R.cull0_tiles = 0;
    free_lanes();
    if (R.ctl) (void)hipFree(R.ctl);
    if (R.dir_mem) (void)hipFree(R.dir_mem);
    if (R.persist) (void)hipFree(R.persist);
    if (R.iter_counts) (void)hipFree(R.iter_counts);
    if (R.h_stats) (void)hipHostFree(R.h_stats);
    if (R.scratch) (void)hipFree(R.scratch);
    if (R.dbg_counts) (void)hipFree(R.dbg_counts);
    if (R.copy_stream) (void)hipStreamSynchronize(R.copy_stream);
    for (auto &h : R.host_regs) (void)hipHostUnregister(h.ptr);
    for (int j = 0; j < 2; ++j) {
        if (R.snap[j]) (void)hipFree(R.snap[j]);
        if (R.ev_snap[j]) (void)hipEventDestroy(R.ev_snap[j]);
        if (R.ev_copied[j]) (void)hipEventDestroy(R.ev_copied[j]);
    }
    if (R.copy_stream) (void)hipStreamDestroy(R.copy_stream);
    for (hipEvent_t e : R.ev) (void)hipEventDestroy(e);
    if (R.stream && R.own_stream) (void)hipStreamDestroy(R.stream);
    R = Renderer{};
}

static int init_impl(const pt_scene_desc *d);

// PT_MESH_BVH: one tree per mesh (pt_bvh.hpp), all trees in one node buffer; the leaf-ordered copies
// of the triangle records carry the original index in word 9.  Geom record words 2/3 of a mesh
// become (root node, triangle count).
static int upload_bvh(const pt_scene_desc *d, std::vector<float> &grec) {
    std::vector<float> nodes, btris, tops;
    std::vector<int32_t> mesh_list;                          // {geom, root record, triangles, 0} in geom order
    float prune = 0.0f;
    int guard = 1;
    R.bvh_info = pt_bvh_info{};
    R.mesh_grids.clear();
    std::vector<int> by_geom((size_t)d->num_meshes);
    for (int k = 0; k < d->num_meshes; ++k) by_geom[(size_t)k] = k;
    std::sort(by_geom.begin(), by_geom.end(), [&](int x, int y) { return d->meshes[x].geom_index < d->meshes[y].geom_index; });
    for (int kk = 0; kk < d->num_meshes; ++kk) {
        const int k = by_geom[(size_t)kk];
        const pt_mesh &m = d->meshes[k];
        if (kk > 0 && d->meshes[by_geom[(size_t)kk - 1]].geom_index == m.geom_index)
            return fail(PT_ERR_INVALID, "pt_init: geom %d owns more than one mesh", m.geom_index);
        ptbvh::Tree tree;
        ptbvh::build(reinterpret_cast<const float *>(d->triangles + m.first_triangle), m.triangle_count, tree, (double)R.scene.rmax);
        const int root = (int)(nodes.size() / BVH_NODE_WORDS);
        const int slot0 = (int)(btris.size() / TRI_WORDS);
        if ((int64_t)slot0 + m.triangle_count >= (1 << ptbvh::LINK_BITS) || tree.num_nodes() >= (1 << ptbvh::LINK_BITS))
            return fail(PT_ERR_INVALID, "pt_init: PT_MESH_BVH holds at most 2^24 triangles (record links are 24 bits)");
        for (int n = 0; n < tree.num_nodes(); ++n) {           // leaf children: slot in the tree -> slot in the shared buffer
            float *w = &tree.nodes[(size_t)n * BVH_NODE_WORDS];
            for (int c = 0; c < 2; ++c) {
                uint32_t link;
                memcpy(&link, &w[6 + c], 4);
                if ((link >> ptbvh::LINK_BITS) & ptbvh::INFO_LEAF) { link += (uint32_t)slot0; memcpy(&w[6 + c], &link, 4); }
            }
        }
        nodes.insert(nodes.end(), tree.nodes.begin(), tree.nodes.end());
        for (int s = 0; s < m.triangle_count; ++s) {
            const int32_t orig = m.first_triangle + tree.order[(size_t)s];
            const pt_triangle &t = d->triangles[orig];
            float r[TRI_WORDS] = {t.v0.x, t.v0.y, t.v0.z,
                                  t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z,
                                  t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z, 0.0f, 0.0f, 0.0f};
            memcpy(&r[9], &orig, 4);
            r[10] = tree.spec_pad;
            btris.insert(btris.end(), r, r + TRI_WORDS);
        }
        float *g = grec.data() + (size_t)m.geom_index * ptd::GEOM_WORDS;
        memcpy(&g[2], &root, 4); memcpy(&g[3], &m.triangle_count, 4);
        for (int a = 0; a < 3; ++a) { g[ptd::G_INV + a] = tree.origin[a]; g[ptd::G_INV + 3 + a] = tree.step[a]; }   // the mesh's grid
        for (int a = 0; a < 3; ++a) R.mesh_grids.push_back(tree.origin[a] - tree.step[a]);
        for (int a = 0; a < 3; ++a) R.mesh_grids.push_back(tree.origin[a] + (float)(ptbvh::GRID_MAX + 1) * tree.step[a]);
        // the first records of this tree (its most visited ones, pt_bvh.hpp: number) go into the LDS copy k_mesh keeps
        const int share = d->num_meshes <= BVH_TOP ? BVH_TOP / d->num_meshes : 0;
        const int top_cnt = std::min(share, tree.num_nodes()), top_off = (int)(tops.size() / BVH_NODE_WORDS);
        tops.insert(tops.end(), tree.nodes.begin(), tree.nodes.begin() + (size_t)top_cnt * BVH_NODE_WORDS);
        const int32_t entry[4] = {m.geom_index, root, m.triangle_count, top_off | (top_cnt << 16)};
        mesh_list.insert(mesh_list.end(), entry, entry + 4);
        prune = std::max(prune, tree.prune);
        guard = std::max(guard, tree.num_nodes() + 1);
        R.bvh_info.nodes += tree.num_nodes();
        R.bvh_info.triangles += m.triangle_count;
        R.bvh_info.depth = std::max(R.bvh_info.depth, tree.depth);
        R.bvh_info.pad = std::max(R.bvh_info.pad, tree.pad);
    }
    R.bvh_info.prune = prune;
    if (nodes.size() * 4 >= ((size_t)1 << 32))
        return fail(PT_ERR_INVALID, "pt_init: PT_MESH_BVH holds at most 4 GiB of hierarchy records (k_mesh addresses them with 32-bit offsets)");
    if (nodes.empty()) nodes.assign(BVH_NODE_WORDS, 0.0f);
    if (btris.empty()) btris.assign(TRI_WORDS, 0.0f);
    HIPCHK(hipMalloc(&R.d_bvh_nodes, nodes.size() * 4));
    HIPCHK(hipMalloc(&R.d_bvh_tris, btris.size() * 4));
    HIPCHK(hipMemcpy(R.d_bvh_nodes, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(R.d_bvh_tris, btris.data(), btris.size() * 4, hipMemcpyHostToDevice));
    if (mesh_list.empty()) mesh_list.assign(4, 0);
    HIPCHK(hipMalloc((void **)&R.d_bvh_meshes, mesh_list.size() * 4));
    HIPCHK(hipMemcpy(R.d_bvh_meshes, mesh_list.data(), mesh_list.size() * 4, hipMemcpyHostToDevice));
    R.scene.bvh_meshes = R.d_bvh_meshes; R.scene.bvh_nmesh = d->num_meshes;
    R.scene.bvh_nodes = R.d_bvh_nodes; R.scene.bvh_tris = R.d_bvh_tris;
    R.scene.bvh_top_n = (int)(tops.size() / BVH_NODE_WORDS);
    if (tops.empty()) tops.assign(BVH_NODE_WORDS, 0.0f);
    HIPCHK(hipMalloc(&R.d_bvh_top, tops.size() * 4));
    HIPCHK(hipMemcpy(R.d_bvh_top, tops.data(), tops.size() * 4, hipMemcpyHostToDevice));
    R.scene.bvh_top = R.d_bvh_top;
    R.scene.bvh_prune = prune; R.scene.bvh_guard = guard;
    return PT_OK;
}

// Bounce 0, pinhole camera: which 64-pixel tiles of the local frame can see a mesh at all.  A camera ray is
// d = view - right * alpha - up * beta with alpha = pixelLength.x * (fx - W/2), beta likewise (pathtrace.cu:136-139),
// fx within half a pixel of the pixel's x.  A ray whose triangle hit the spec accepts reports a point inside that
// mesh's box grid (the hit-point test, pt_bvh.hpp), so the pixel lies inside the perspective image of the grid's
// eight corners -- computed here in double, widened by two pixels -- and every other tile can skip ray generation,
// root tests and walks in k_mesh.  No mask (nullptr) when a corner is not in front of the camera, the frame does
// not tile by 64 pixels, or a thin lens is on (then rays do not start at the eye).
// (re)build the bounce-0 candidate masks for the current camera and cull boxes: one launch on the stream, ordered
// behind whatever still reads the old masks and ahead of everything enqueued later (the buffer never moves, so
// captured graphs stay valid)
static int update_cull0() {
    if (!R.cull0_tiles) return PT_OK;
    hipLaunchKernelGGL(k_cull0_mask, dim3((R.cull0_tiles + WAVES - 1) / WAVES), dim3(BLOCK), 0, R.stream, R.scene, R.cam,
                       R.map, R.trace_depth, R.d_cull0, R.cull0_tiles);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

static int update_cam_mask() {
    R.cam_mask_valid = false;
    if (R.mesh_mode != MESH_BVH || R.map.tile_pixels % TILE != 0 || R.mesh_grids.empty()) return PT_OK;
    const pt_camera &c = R.cam;
    const double M[3][3] = {{c.view.x, -c.right.x, -c.up.x}, {c.view.y, -c.right.y, -c.up.y}, {c.view.z, -c.right.z, -c.up.z}};
    const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                       M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    if (!(std::fabs(det) > 1e-9) || !(c.pixelLength[0] != 0.0f) || !(c.pixelLength[1] != 0.0f)) return PT_OK;
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (size_t m = 0; m + 6 <= R.mesh_grids.size(); m += 6) {
        for (int corner = 0; corner < 8; ++corner) {
            const double v[3] = {(double)R.mesh_grids[m + ((corner & 1) ? 3 : 0)] - c.position.x,
                                 (double)R.mesh_grids[m + 1 + ((corner & 2) ? 3 : 0)] - c.position.y,
                                 (double)R.mesh_grids[m + 2 + ((corner & 4) ? 3 : 0)] - c.position.z};
            // Cramer: (s, s*alpha, s*beta) = M^-1 v
            auto det3 = [](const double A[3][3]) {
                return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
            };
            double sol[3];
            for (int k = 0; k < 3; ++k) {
                double A[3][3];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = (j == k) ? v[i] : M[i][j];
                sol[k] = det3(A) / det;
            }
            const double reach = std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]);
            if (!(sol[0] > 1e-6 * (reach + 1.0))) return PT_OK;                  // at or behind the eye: no mask
            const double fx = 0.5 * c.resolution[0] + sol[1] / sol[0] / (double)c.pixelLength[0];
            const double fy = 0.5 * c.resolution[1] + sol[2] / sol[0] / (double)c.pixelLength[1];
            if (!std::isfinite(fx) || !std::isfinite(fy)) return PT_OK;
            x0 = std::min(x0, fx); x1 = std::max(x1, fx); y0 = std::min(y0, fy); y1 = std::max(y1, fy);
        }
    }
    x0 -= 2.0; x1 += 2.0; y0 -= 2.0; y1 += 2.0;
    const uint32_t tps = (uint32_t)R.map.tile_pixels / TILE;
    std::vector<unsigned long long> mask((tps + 63) / 64, 0ull);
    for (uint32_t t = 0; t < tps; ++t) {
        bool any = false;
        for (int k = 0; k < TILE && !any; ++k) {
            const int j = (int)t * TILE + k;
            int pix = j;
            if (R.map.tile_count != 1) {                                          // pt_types.hpp: local_to_pixel
                const int ly = j / R.map.W, x = j - ly * R.map.W, ls = ly / R.map.strip_rows;
                pix = x + ((ls * R.map.tile_count + R.map.tile_index) * R.map.strip_rows + (ly - ls * R.map.strip_rows)) * R.map.W;
            }
            const int y = pix / R.map.W, x = pix - y * R.map.W;
            any = x >= x0 && x <= x1 && y >= y0 && y <= y1;
        }
        if (any) mask[t >> 6] |= 1ull << (t & 63u);
    }
    if (!R.d_cam_mask) HIPCHK(hipMalloc((void **)&R.d_cam_mask, mask.size() * sizeof(unsigned long long)));
    HIPCHK(hipMemcpy(R.d_cam_mask, mask.data(), mask.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    R.cam_mask_valid = true;
    return PT_OK;
}

int pt_init(const pt_scene_desc *d) {
    if (!d) return fail(PT_ERR_INVALID, "pt_init: null descriptor");
    if (R.live) pt_free();
    const int rc = init_impl(d);
    if (rc != PT_OK) {                 // release whatever was allocated; keep the message
        char keep[ERR_BYTES];
        memcpy(keep, t_err, sizeof keep);
        R.live = true;
        pt_free();
        memcpy(t_err, keep, sizeof keep);
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
    if (d->num_meshes < 0 || d->num_triangles < 0 || (d->num_meshes > 0 && !d->meshes) || (d->num_triangles > 0 && !d->triangles))
        return fail(PT_ERR_INVALID, "pt_init: meshes / triangles missing");
    for (int k = 0; k < d->num_meshes; ++k) {
        const pt_mesh &m = d->meshes[k];
        if (m.geom_index < 0 || m.geom_index >= d->num_geoms || d->geoms[m.geom_index].type != PT_TRIANGLE_MESH ||
            m.first_triangle < 0 || m.triangle_count < 0 ||
            (int64_t)m.first_triangle + (int64_t)m.triangle_count > (int64_t)d->num_triangles)
            return fail(PT_ERR_INVALID, "pt_init: mesh %d is inconsistent", k);
        for (int j = 0; j < k; ++j)          // one mesh per geom, whatever the mesh mode (the loop would silently use the first)
            if (d->meshes[j].geom_index == m.geom_index)
                return fail(PT_ERR_INVALID, "pt_init: geom %d owns more than one mesh", m.geom_index);
    }
    if ((d->flags & PT_CACHE_FIRST) && ((d->flags & PT_AA_JITTER) || d->lens_radius > 0.0f))
        return fail(PT_ERR_INVALID, "pt_init: PT_CACHE_FIRST needs identical camera rays every iteration; it cannot be "
                                    "combined with PT_AA_JITTER or a lens (INSTRUCTION.md:113)");
    if (d->lens_radius > 0.0f && !(d->focal_distance > 0.0f))
        return fail(PT_ERR_INVALID, "pt_init: a lens needs focal_distance > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PT_ERR_DEVICE, "pt_init: no HIP device (this library has no CPU fallback)");
    if (d->device < 0 || d->device >= ndev) return fail(PT_ERR_INVALID, "pt_init: device %d of %d", d->device, ndev);
    HIPCHK(hipSetDevice(d->device));

    R = Renderer{};
    R.desc = *d; R.cam = d->camera; R.trace_depth = d->trace_depth; R.flags = d->flags; R.device = d->device;
    R.lens = Lens{(d->flags & PT_AA_JITTER) ? 1 : 0, d->lens_radius, d->focal_distance};
    if (const char *ug = getenv("PTMI355_GRAPH")) R.use_graphs = atoi(ug) != 0;
    R.whole_max_paths = 6000000;     // measured at 800x800 (r02): 1 spp +38 %, 4 spp +20 %, 8 spp +8 %, 16 spp -4 %
    if (const char *wm = getenv("PTMI355_WHOLE_MAX")) R.whole_max_paths = strtoull(wm, nullptr, 10);
    R.whole_max_host_paths = 16000000;
    if (const char *wm = getenv("PTMI355_WHOLE_MAX_HOST")) R.whole_max_host_paths = strtoull(wm, nullptr, 10);
    if (getenv("PTMI355_WHOLE_MAX") && !getenv("PTMI355_WHOLE_MAX_HOST")) R.whole_max_host_paths = R.whole_max_paths;   // (tests pin the launch plan with it)
    if (const char *e = getenv("PTMI355_OVERLAP")) {           // 0: off; 1: on (default lanes); n >= 2: n lanes
        const int nl = atoi(e);
        R.ov_enabled = nl != 0;
        if (nl >= 2) { R.ov_lanes = std::min(nl, OV_MAX_LANES); R.ov_lanes_set = true; }
    }
    if (const char *e = getenv("PTMI355_OVERLAP_GB")) R.ov_budget_gb = atof(e);
    if (const char *e = pt_experiment("PTMI355_LANE_STREAMS")) R.ov_streams = std::max(1, atoi(e));
    R.epi_enabled = true;
    if (const char *e = pt_experiment("PTMI355_HOST_EPILOGUE")) R.epi_enabled = atoi(e) != 0;
    if (const char *e = pt_experiment("PTMI355_EPI_DIRECT")) R.epi_direct_enabled = atoi(e) != 0;
    R.host_sparse_enabled = (d->flags & (PT_HOST_SPARSE | PT_SHARED_IMAGE)) != 0;
    if (const char *e = pt_experiment("PTMI355_ASYNC_DIRECT")) R.async_direct_enabled = atoi(e) != 0;
    R.pin_enabled = true;
    if (const char *e = pt_experiment("PTMI355_PIN")) R.pin_enabled = atoi(e) != 0;
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
    else {
        // The library's own launch stream ranks above the lanes' streams.  What runs on it between overlapped batches are
        // their gathers, a few microseconds each, and a lane's next batch waits for one.  Priority classes have hardware
        // queues of their own: at the default priority the launch stream shares one of the runtime's four queues with
        // whichever lanes were created fourth, eighth, ... after it, and a gather then waits behind a whole k_iteration
        // launch of such a lane (or not, depending on how many streams the process had made before: 1 spp per call
        // measured anything between 15 and 31 Grays/s with 2-8 lanes and 4 / 8 queues, profiles/r04/ab_hw_queues.log).
        int lo = 0, hi = 0;
        const char *pe = pt_experiment("PTMI355_MAIN_PRIO");
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        if ((pe && atoi(pe) == 0) || hipStreamCreateWithPriority(&R.stream, hipStreamNonBlocking, hi) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(hipStreamCreateWithFlags(&R.stream, hipStreamNonBlocking));
        }
        R.own_stream = true;
    }
    R.live = true;

    // scene -> device records
    std::vector<float> grec((size_t)std::max(1, d->num_geoms) * ptd::GEOM_WORDS, 0.0f);
    for (int i = 0; i < d->num_geoms; ++i) {
        const pt_geom &g = d->geoms[i];
        float *r = grec.data() + (size_t)i * ptd::GEOM_WORDS;
        int first = 0, count = 0, boff = 0;
        for (int k = 0, off = 0; k < d->num_meshes; ++k) {       // boff: where upload_tri_bounds puts the mesh's spheres
            if (d->meshes[k].geom_index == i) { first = d->meshes[k].first_triangle; count = d->meshes[k].triangle_count; boff = off; break; }
            off += (d->meshes[k].triangle_count + 63) & ~63;
        }
        memcpy(&r[0], &g.type, 4); memcpy(&r[1], &g.materialid, 4); memcpy(&r[2], &first, 4); memcpy(&r[3], &count, 4);
        const pt_mat4 *ms[3] = {&g.inverseTransform, &g.transform, &g.invTranspose};
        const int offs[3] = {ptd::G_INV, ptd::G_FWD, ptd::G_INVT};
        for (int m = 0; m < 3; ++m)
            for (int c = 0; c < 4; ++c)
                for (int rr = 0; rr < 3; ++rr) r[offs[m] + c * 3 + rr] = ms[m]->m[c][rr];
        if (g.type == PT_TRIANGLE_MESH) memcpy(&r[ptd::G_INV + 6], &boff, 4);     // a mesh's matrices are never read
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
    for (int k = 0; k < d->num_meshes; ++k) {              // word 10: the pad of the spec's hit-point test (per mesh)
        const pt_mesh &m = d->meshes[k];
        const float pad = ptbvh::spec_pad(reinterpret_cast<const float *>(d->triangles + m.first_triangle), m.triangle_count);
        for (int i = 0; i < m.triangle_count; ++i) trec[(size_t)(m.first_triangle + i) * TRI_WORDS + 10] = pad;
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
    {   // per-lane gather records (the three matrices, 4 columns x 3 rows each) and geom info words
        std::vector<float> gath((size_t)std::max(1, d->num_geoms) * GREC_WORDS, 0.0f);
        std::vector<uint32_t> ginfo((size_t)std::max(1, d->num_geoms), 0u);
        for (int i = 0; i < d->num_geoms; ++i) {
            const pt_geom &g = d->geoms[i];
            float *r = gath.data() + (size_t)i * GREC_WORDS;
            const pt_mat4 *ms[3] = {&g.inverseTransform, &g.transform, &g.invTranspose};
            for (int m = 0; m < 3; ++m)
                for (int c = 0; c < 4; ++c)
                    for (int rr = 0; rr < 3; ++rr) r[m * 12 + c * 3 + rr] = ms[m]->m[c][rr];
            ginfo[(size_t)i] = (uint32_t)g.materialid | ((uint32_t)g.type << 28);
        }
        if (d->num_materials >= (1 << 28)) return fail(PT_ERR_INVALID, "pt_init: at most 2^28 materials");
        HIPCHK(hipMalloc(&R.d_grec, gath.size() * 4));
        HIPCHK(hipMalloc((void **)&R.d_ginfo, ginfo.size() * 4));
        HIPCHK(hipMemcpy(R.d_grec, gath.data(), gath.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(R.d_ginfo, ginfo.data(), ginfo.size() * 4, hipMemcpyHostToDevice));
        R.scene.grec = R.d_grec; R.scene.ginfo = R.d_ginfo;
    }
    R.mesh_mode = MESH_NONE;
    for (int i = 0; i < d->num_geoms; ++i)
        if (d->geoms[i].type == PT_TRIANGLE_MESH) R.mesh_mode = (d->flags & PT_MESH_BVH) ? MESH_BVH : MESH_TILES;
    R.geoms_keep.assign(d->geoms, d->geoms + d->num_geoms);
    if (d->num_triangles > 0) R.tris_keep.assign(d->triangles, d->triangles + d->num_triangles);
    R.desc.geoms = R.geoms_keep.data();
    R.desc.triangles = R.tris_keep.empty() ? nullptr : R.tris_keep.data();
    if (d->num_meshes > 0) R.meshes_keep.assign(d->meshes, d->meshes + d->num_meshes);
    R.desc.meshes = R.meshes_keep.empty() ? nullptr : R.meshes_keep.data();
    {
        const int rc = upload_cull(&R.desc, R.cam);
        if (rc != PT_OK) return rc;
    }
    if (R.mesh_mode == MESH_BVH) {
        const int rc = upload_bvh(&R.desc, grec);
        if (rc != PT_OK) return rc;
        HIPCHK(hipMemcpy(R.d_geoms, grec.data(), grec.size() * 4, hipMemcpyHostToDevice));   // records now name tree roots
    }
    R.grec_keep = grec;
    // LDS per workgroup: control words + (scene block, when it is small enough to leave room for five workgroups
    // per CU) + the four per-wave blocks (+ the triangle tile).  A scene that does not fit is gathered from global
    // memory through the vector cache instead: any number of primitives / materials runs.
    {
        const size_t base = ((size_t)LDS_CTL_WORDS + (size_t)WAVES * PW_WORDS) * 4 +
                            (R.mesh_mode == MESH_TILES ? (size_t)WAVES * TRQ_WORDS * 4 : 0);
        const size_t scene = (size_t)scene_lds_words(d->num_materials, d->num_geoms) * 4;
        R.scene_lds = base + scene <= 32 * 1024;
        if (const char *e = getenv("PTMI355_SCENE_LDS")) R.scene_lds = atoi(e) != 0 && base + scene <= 64 * 1024;   // tests force the global path
        R.lds_bytes = base + (R.scene_lds ? scene : 0);
        R.lds_bytes = (R.lds_bytes + 15) & ~(size_t)15;
        if (const char *pad = pt_experiment("PTMI355_LDS_PAD")) R.lds_bytes += (size_t)atoi(pad);     // occupancy experiments
    }

    // PT_SORT_MATERIAL in its fused form (pt_types.hpp: RangeDir): survivors are placed by the material they hit, one span
    // per (material, wave) -- the pools are K times as large, nothing else is read or written for the sort.  Taken when
    // the scene has up to 64 materials (one counter per lane), compaction is on, no other pipeline flag asks for
    // materialised intersections, meshes are not walked by the pre-pass (its flags are per physical slot) and the pools
    // fit the budget (PTMI355_SORT_FUSED_GB, default 96 of the 288 GB); otherwise the two-kernel form (k_intersect ->
    // k_sort_hist -> k_shade_sorted_w) runs.  PT_UNFUSED | PT_SORT_MATERIAL always selects the latter.
    R.sort_keys = 0;
    if ((R.flags & PT_SORT_MATERIAL) && (R.flags & PT_COMPACT) && !(R.flags & (PT_UNFUSED | PT_FAKE_SHADER | PT_CACHE_FIRST)) &&
        R.mesh_mode != MESH_BVH && d->num_materials <= 64) {
        bool on = true;
        if (const char *e = pt_experiment("PTMI355_SORT_FUSED")) on = atoi(e) != 0;
        double budget_gb = 96.0;
        if (const char *e = pt_experiment("PTMI355_SORT_FUSED_GB")) budget_gb = atof(e);
        R.sort_runs = 1;              // more runs per wave (each wave a share of every part of the key space): measured slower (profiles/r03/variants_sort.log)
        if (const char *e = pt_experiment("PTMI355_SORT_RUNS")) R.sort_runs = std::max(1, std::min(8, atoi(e)));
        const double tiles_k = (double)d->num_materials * ((double)((R.cap + 63) / 64) + 8192.0 * R.sort_runs);
        if (on && tiles_k * 2560.0 * 2.0 <= budget_gb * 1e9 && tiles_k * 64.0 < 2147483648.0) R.sort_keys = d->num_materials;
    }
    // pools, intersections, final colours, image, control
    const size_t capz = R.cap;
    const size_t pool_mult = (size_t)std::max(1, R.sort_keys);
    const size_t run_mult = R.sort_keys > 0 ? (size_t)R.sort_runs : 1;        // every run's span is rounded up to whole tiles
    for (int k = 0; k < 2; ++k) {
        // whole 64-path tiles, plus one tile per wave of the largest grid (W <= 8192): wave w's span starts at slot
        // w * R * 64 with R = ceil(tiles / W), so the spans of the last waves reach up to W tiles past the pool's paths --
        // never written while a wave only packs its own survivors, but k_iteration deals a workgroup's survivors to
        // all four of its waves, whichever of them had paths at bounce 0
        // ... and, with tiles aligned to the ranges (pt_types.hpp: RangeDir), one more: a reader's run is R' = ceil((tiles + up
        // to one partly filled tile per range) / W) tiles long, and its survivors' span is as long as its run
        R.pool_bytes = pool_mult * (((capz + 63) / 64) + 2 * 8192 * run_mult) * 64 * 10 * 4;
        HIPCHK(hipMalloc(&R.pool_mem[k], R.pool_bytes));
        R.pool[k] = carve_pool(R.pool_mem[k], R.cap);
    }
    // the ShadeableIntersection planes exist only where a pipeline materialises them (the fused path keeps them
    // in registers): unfused / sorted / fake-shader pipelines now, pt_intersect_once on first use
    if ((R.flags & (PT_UNFUSED | PT_FAKE_SHADER)) || ((R.flags & PT_SORT_MATERIAL) && !R.sort_keys)) {
        const int rc = ensure_isect();
        if (rc != PT_OK) return rc;
    }
    R.final_bytes = capz * 4 * 4;
    HIPCHK(hipMalloc(&R.final_mem, R.final_bytes));
    HIPCHK(hipMemsetAsync(R.final_mem, 0, capz * 4 * 4, R.stream));          // no entry carries a stamp yet (stamps start at 1)
    R.fin_serial = 0;
    if (const char *e = pt_experiment("PTMI355_FIN_SERIAL")) R.fin_serial = (uint32_t)strtoul(e, nullptr, 0);   // tests: start near the wrap
    if (d->device_image) { R.image = d->device_image; R.own_image = false; }
    else {
        HIPCHK(hipMalloc(&R.image, (size_t)R.npix * 3 * 4));
        R.own_image = true;
        HIPCHK(hipMemsetAsync(R.image, 0, (size_t)R.npix * 3 * 4, R.stream));      // pathtrace.cu:85
    }
    R.max_tiles = (R.cap + TILE - 1) / TILE;
    // only the election buckets of the bounces this scene can run are cleared per batch
    R.ctl_bytes = offsetof(Control, bucket) - offsetof(Control, stamp) +
                  (size_t)R.trace_depth * sizeof(((Control *)nullptr)->bucket[0]);
    HIPCHK(hipMalloc((void **)&R.ctl, sizeof(Control)));
    HIPCHK(hipMemsetAsync(R.ctl, 0, sizeof(Control), R.stream));      // incl. Control::ticket, which no batch clears
    HIPCHK(hipMalloc((void **)&R.persist, sizeof(Persist)));
    HIPCHK(hipMemsetAsync(R.persist, 0, sizeof(Persist), R.stream));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, d->device));
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // persistent grid: as many workgroups as are co-resident for the fused kernel (tiles are
    // dealt round-robin, so more workgroups than that only re-stage the scene)
    // ... counted on the variants this session launches (scene in LDS or not, with and without ray generation, with or
    // without the material keys): they differ in registers, and a grid one workgroup per CU too large for the variant
    // that runs serialises a whole extra round of workgroups (C3 sorted at 6 per CU instead of its 5: -23 %)
    int per_cu = 8;
    {
        const bool sorted = R.sort_keys > 0;
        const void *fns[2];
        if (R.mesh_mode == MESH_BVH) { fns[0] = bounce_fn<MESH_PRE>(R.scene_lds, false, false); fns[1] = bounce_fn<MESH_PRE>(R.scene_lds, true, false); }
        else if (R.mesh_mode == MESH_TILES) { fns[0] = bounce_fn<MESH_TILES>(R.scene_lds, false, sorted); fns[1] = bounce_fn<MESH_TILES>(R.scene_lds, true, sorted); }
        else { fns[0] = bounce_fn<MESH_NONE>(R.scene_lds, false, sorted); fns[1] = bounce_fn<MESH_NONE>(R.scene_lds, true, sorted); }
        for (const void *f : fns) {
            int n = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, BLOCK, R.lds_bytes));
            per_cu = std::min(per_cu, n);
        }
    }
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    if (const char *e = pt_experiment("PTMI355_WGS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e)));   // occupancy experiments
    R.grid = (int)std::min<uint32_t>((R.max_tiles + WAVES - 1) / WAVES, (uint32_t)cus * (uint32_t)per_cu);
    if (R.grid < 1) R.grid = 1;
    if (R.grid * WAVES > 8192) R.grid = 8192 / WAVES;           // the pools' slack and the directory scan are sized for W <= 8192
    {   // k_iteration has no directory and no cross-workgroup step: its grid is its own co-resident count
        int n = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &n, R.scene_lds ? (const void *)k_iteration<true> : (const void *)k_iteration<false>, BLOCK, R.lds_bytes));
        n = std::max(1, std::min(n, 8));
        if (const char *e = pt_experiment("PTMI355_WGS_PER_CU")) n = std::max(1, std::min(n, atoi(e)));
        R.grid_iter = (int)std::min<uint32_t>((R.max_tiles + WAVES - 1) / WAVES, (uint32_t)cus * (uint32_t)n);
        R.grid_iter = std::max(1, std::min(R.grid_iter, 8192 / WAVES));
        R.grid_iter_cur = R.grid_iter; R.cus = cus;
        if (const char *e = pt_experiment("PTMI355_ITER_TPW")) R.iter_tpw = std::max(0, atoi(e));
        if (const char *e = pt_experiment("PTMI355_ITER_WGS_ALL")) R.iter_wgs_per_cu_all = std::max(1, atoi(e));
        // its traced counts, [bounce][workgroup], and the page-locked block its last workgroup writes a synchronous call's
        // statistics to (if the host allocation cannot be mapped the control block is copied back as before)
        R.iter_counts_bytes = (size_t)MAX_DEPTH * (size_t)R.grid_iter * 4;
        HIPCHK(hipMalloc((void **)&R.iter_counts, R.iter_counts_bytes));
        void *hs = nullptr, *ds = nullptr;
        if (hipHostMalloc(&hs, sizeof(HostStats), hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&ds, hs, 0) == hipSuccess) {
            memset(hs, 0, sizeof(HostStats));
            R.h_stats = (HostStats *)hs; R.d_stats = (HostStats *)ds;
        } else {
            (void)hipGetLastError();
            if (hs) (void)hipHostFree(hs);
        }
    }
    if (R.mesh_mode == MESH_BVH) {
        R.mesh_hit_bytes = (size_t)(((capz + 63) / 64) * 64) * sizeof(float4);
        HIPCHK(hipMalloc((void **)&R.mesh_hit, R.mesh_hit_bytes));
        // k_mesh reads the flags of whole ranges (waves x tiles per range can overshoot the pool by up to one tile per
        // wave) and in chunks of 8 tiles: the words past the pool exist and stay zero
        R.flag_words = (size_t)R.max_tiles + 2 * (size_t)R.grid * WAVES + 8;
        for (int k = 0; k < 2; ++k) {
            HIPCHK(hipMalloc((void **)&R.mesh_flags[k], R.flag_words * sizeof(unsigned long long)));
            HIPCHK(hipMemsetAsync(R.mesh_flags[k], 0, R.flag_words * sizeof(unsigned long long), R.stream));
        }
        // one 16-wave workgroup per CU: 4 waves per SIMD (the kernel's register budget), one LDS copy of the tree tops
        HIPCHK(hipFuncSetAttribute((const void *)k_mesh<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MESH_LDS_BYTES));
        HIPCHK(hipFuncSetAttribute((const void *)k_mesh<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MESH_LDS_BYTES));
        R.grid_mesh = (int)std::min<uint32_t>((R.max_tiles + MESH_WG_WAVES - 1) / MESH_WG_WAVES, (uint32_t)cus);
        if (R.grid_mesh < 1) R.grid_mesh = 1;
    }
    if (R.flags & PT_CACHE_FIRST) HIPCHK(hipMalloc(&R.cache_mem, (size_t)R.map.tile_pixels * 5 * 4));
    if ((R.flags & PT_SORT_MATERIAL) && !R.sort_keys) {
        if (d->num_materials + 1 > SORT_MAX_BINS)
            return fail(PT_ERR_INVALID, "pt_init: PT_SORT_MATERIAL keeps one bin per material in LDS: at most %d materials", SORT_MAX_BINS - 1);
        {
            int per_cu_sort = 8;                              // nothing in these kernels needs co-residency; 8 per CU measured best (5: -4 %)
            if (const char *e = pt_experiment("PTMI355_SORT_WGS")) per_cu_sort = std::max(1, atoi(e));
            R.sort_wave = true;
            if (const char *e = pt_experiment("PTMI355_SORT_WAVE")) R.sort_wave = atoi(e) != 0;
            const uint32_t chunks = (R.cap + SORT_CHUNK - 1) / SORT_CHUNK;
            R.grid_sort = (int)std::max<uint32_t>(1u, std::min<uint32_t>(chunks, (uint32_t)cus * (uint32_t)per_cu_sort));
        }
        HIPCHK(hipMalloc((void **)&R.sort_table, ((size_t)(d->num_materials + 1) * R.grid_sort + 4) * sizeof(uint32_t)));   // + the scan's last 16-B load
    }
    {   // range directory: one count + one base per wave of the persistent grid, per bounce
        const size_t Wp = ((size_t)R.grid * WAVES * pool_mult * run_mult + 3) & ~(size_t)3;
        R.dir_stride = range_dir_words(Wp);
        // one directory per bounce up to MAX_DEPTH: traceDepth is re-read on every call and may GROW (pathtrace.cu:286)
        R.dir_bytes = (size_t)MAX_DEPTH * R.dir_stride * sizeof(uint32_t);
        HIPCHK(hipMalloc((void **)&R.dir_mem, R.dir_bytes));
    }
    {
        const int rc = update_cam_mask();
        if (rc != PT_OK) return rc;
    }
    {
        bool on = true;
        if (const char *e = getenv("PTMI355_CULL0")) on = atoi(e) != 0;
        if (on && R.scene.ngeoms >= 1 && R.scene.ngeoms <= 64 && R.map.tile_pixels % TILE == 0) {
            R.cull0_tiles = (uint32_t)(R.map.tile_pixels / TILE);
            HIPCHK(hipMalloc((void **)&R.d_cull0, (size_t)R.cull0_tiles * sizeof(unsigned long long)));
            const int rc = update_cull0();
            if (rc != PT_OK) return rc;
        }
    }
    if (const char *e = pt_experiment("PTMI355_DBG_COUNTS")) {
        R.dbg_words = (size_t)std::max(64, atoi(e));
        HIPCHK(hipMalloc((void **)&R.dbg_counts, R.dbg_words * 4));
        HIPCHK(hipMemsetAsync(R.dbg_counts, 0, R.dbg_words * 4, R.stream));
    }
    HIPCHK(hipStreamSynchronize(R.stream));
    t_err[0] = 0;
    return PT_OK;
}

int pt_set_camera(const pt_camera *camera, int trace_depth) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_set_camera: not initialised");
    if (!camera) return fail(PT_ERR_INVALID, "pt_set_camera: null camera");
    if (camera->resolution[0] != R.map.W || camera->resolution[1] != R.map.H)
        return fail(PT_ERR_INVALID, "pt_set_camera: resolution changed (%dx%d -> %dx%d); re-init instead",
                    R.map.W, R.map.H, camera->resolution[0], camera->resolution[1]);
    if (trace_depth < 1 || trace_depth > MAX_DEPTH)
        return fail(PT_ERR_INVALID, "pt_set_camera: trace_depth %d outside [1, %d]", trace_depth, MAX_DEPTH);
    if (memcmp(&R.cam, camera, sizeof R.cam) != 0) { R.cache_valid = false; drop_graphs(); }   // refill the bounce-0 cache
    bool recull = false;
    {   // the cull boxes hold for ray origins within R.scene.rmax (1-norm); a camera outside that range would only
        // make its rays candidates of every primitive (correct, slow): remake the boxes around the new position
        const double reach = (double)std::fabs(camera->position.x) + std::fabs(camera->position.y) + std::fabs(camera->position.z);
        if (std::isfinite(reach) && reach > (double)R.scene.rmax && reach != R.cull_eye_reach) {
            const int rc = upload_cull(&R.desc, *camera);
            if (rc != PT_OK) return rc;
            drop_graphs();
            recull = true;
        }
    }
    if (trace_depth != R.trace_depth) {
        drop_graphs();
        // the per-batch clear covers the election buckets of the bounces that can run
        R.ctl_bytes = offsetof(Control, bucket) - offsetof(Control, stamp) +
                      (size_t)trace_depth * sizeof(((Control *)nullptr)->bucket[0]);
    }
    const bool moved = memcmp(&R.cam, camera, sizeof R.cam) != 0;
    R.cam = *camera;
    R.trace_depth = trace_depth;
    if (moved || recull) {
        R.ov_active = false;      // overlapped batches to come wait for what is enqueued here (the launch stream orders it after the ones in flight)
        const int rc = update_cull0();
        if (rc != PT_OK) return rc;
    }
    if (recull && R.mesh_mode == MESH_BVH) {
        // the trees' box padding covers ray origins within the bound that has just grown: rebuild them for the new one
        HIPCHK(hipStreamSynchronize(R.stream));
        float **old[] = {&R.d_bvh_nodes, &R.d_bvh_tris, &R.d_bvh_top};
        for (float **p : old) { if (*p) (void)hipFree(*p); *p = nullptr; }
        if (R.d_bvh_meshes) { (void)hipFree(R.d_bvh_meshes); R.d_bvh_meshes = nullptr; }
        const int rc = upload_bvh(&R.desc, R.grec_keep);
        if (rc != PT_OK) return rc;
        HIPCHK(hipMemcpy(R.d_geoms, R.grec_keep.data(), R.grec_keep.size() * 4, hipMemcpyHostToDevice));
        drop_graphs();
    }
    if ((moved || recull) && R.mesh_mode == MESH_BVH) {
        const bool had = R.cam_mask_valid;
        HIPCHK(hipStreamSynchronize(R.stream));                  // launches in flight still read the old mask
        const int rc = update_cam_mask();
        if (rc != PT_OK) return rc;
        if (had != R.cam_mask_valid) drop_graphs();              // the mask pointer is a (frozen) kernel argument
    }
    return PT_OK;
}

int pt_set_lens(float lens_radius, float focal_distance) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_set_lens: not initialised");
    if (lens_radius > 0.0f && !(focal_distance > 0.0f)) return fail(PT_ERR_INVALID, "pt_set_lens: a lens needs focal_distance > 0");
    if (lens_radius > 0.0f && (R.flags & PT_CACHE_FIRST))
        return fail(PT_ERR_INVALID, "pt_set_lens: PT_CACHE_FIRST cannot be combined with a lens");
    if (R.lens.radius != lens_radius || R.lens.focal != focal_distance) drop_graphs();
    R.lens.radius = lens_radius; R.lens.focal = focal_distance;
    return PT_OK;
}

int pt_synchronize(void) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_synchronize: not initialised");
    HIPCHK(hipStreamSynchronize(R.stream));
    if (R.copy_stream) HIPCHK(hipStreamSynchronize(R.copy_stream));
    return PT_OK;
}

int pt_trace_batch_async(int iter0, int count) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace_batch_async: not initialised");
    R.in_step = false;
    R.ov_ok = true;
    const int rc = enqueue_batch(iter0, count);
    R.ov_ok = false;
    return rc;
}

int pt_trace_batch(int iter0, int count, float *host_image_sum) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace_batch: not initialised");
    R.in_step = false;
    R.ov_ok = false;       // (PT_ASYNC_IMAGE calls are bound by their 7.68 MB copy: lanes measured 14.7 against 15.8 Grays/s there)
    R.want_host_stats = !(host_image_sum && (R.flags & PT_ASYNC_IMAGE));
    int rc = enqueue_batch(iter0, count);
    R.ov_ok = false; R.want_host_stats = false;
    if (rc) return rc;
    if (host_image_sum && (R.flags & PT_ASYNC_IMAGE)) return enqueue_async_image(host_image_sum);
    if (host_image_sum) {
        rc = enqueue_image_copy(host_image_sum);
        if (rc) return rc;
    }
    return collect_stats();                               // one stream synchronisation covers the copy as well
}

// can ONE iteration of this session with a page-locked host image run as one launch that does its own finalGather?
// (what pt_trace decides per call; the multi-GPU form asks once at pt_init: pt_multi.hpp)
bool whole_host_possible(void) {
    return R.live && !(R.flags & (PT_UNFUSED | PT_SORT_MATERIAL | PT_FAKE_SHADER | PT_CACHE_FIRST)) && (R.flags & PT_COMPACT) &&
           R.mesh_mode == MESH_NONE && R.sort_keys == 0 && R.epi_enabled && !R.use_graphs &&
           (uint64_t)R.map.tile_pixels <= std::max(R.whole_max_paths, R.whole_max_host_paths);
}

// One iteration of this context's tile, synchronously, its launch writing the tile's pixels into a host frame that is
// ALREADY page-locked and mapped (`mapped` = this device's address of it): the in-library multi-GPU form of
// pathtrace() with a host image -- every context calls this on its own thread, nothing is exchanged.
int pt_trace_mapped(int iter, float *mapped) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace: not initialised");
    if (!whole_host_possible() || !mapped) return fail(PT_ERR_INTERNAL, "pt_trace_mapped: this context cannot trace an iteration as one launch");
    R.in_step = false;
    R.epi_host = mapped; R.epi_done = false;
    R.ov_ok = false;
    R.want_host_stats = true;
    const int rc = enqueue_batch(iter, 1);
    R.want_host_stats = false;
    const bool gathered = R.epi_done;
    R.epi_host = nullptr; R.epi_done = false;
    if (rc) return rc;
    if (!gathered) return fail(PT_ERR_INTERNAL, "pt_trace_mapped: the iteration did not run as one launch");
    return collect_stats();
}

int pt_trace(uint8_t *pbo_rgba, int frame, int iter, float *host_image_sum) {
    (void)frame;                                          // unused in the reference too (main.cpp:136)
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace: not initialised");
    R.in_step = false;
    // synchronous host image: when this iteration runs as one launch, its waves write the new sums into the caller's
    // (page-locked, device-mapped) buffer as they finish, under the tracing of the others (k_iteration's epilogue)
    R.epi_host = nullptr; R.epi_done = false;
    const bool async_image = host_image_sum && (R.flags & PT_ASYNC_IMAGE);
    // (a tile of a larger frame writes only its own pixels: into a frame its ranks share, PT_SHARED_IMAGE)
    const bool shared_frame = host_image_sum && (R.flags & PT_SHARED_IMAGE) && R.map.tile_count > 1;
    if (host_image_sum && (!async_image || R.async_direct_enabled) && R.epi_enabled && !R.use_graphs && (R.map.tile_count == 1 || shared_frame))
        R.epi_host = map_host(host_image_sum, (size_t)R.npix * 12);
    if (shared_frame && !R.epi_host)
        return fail(PT_ERR_INVALID, "pt_trace: PT_SHARED_IMAGE needs a host frame of 1 MiB or more that can be page-locked and mapped");
    if (R.epi_host && R.dma_last) {                        // a copy-engine transfer into a host buffer may still be running
        HIPCHK(hipStreamWaitEvent(R.stream, R.dma_last, 0));
        R.dma_last = nullptr;
    }
    R.ov_ok = false;       // (PT_ASYNC_IMAGE calls are bound by their 7.68 MB copy: lanes measured 14.7 against 15.8 Grays/s there)
    R.want_host_stats = !(host_image_sum && (R.flags & PT_ASYNC_IMAGE));
    int rc = enqueue_batch(iter, 1);
    R.ov_ok = false; R.want_host_stats = false;
    const bool gathered = R.epi_done;
    R.epi_host = nullptr; R.epi_done = false;
    if (rc) return rc;
    if (pbo_rgba) {
        hipLaunchKernelGGL(k_tonemap, dim3((R.npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, R.stream, pbo_rgba,
                           R.image, R.npix, iter);
        HIPCHK(hipGetLastError());
    }
    if (shared_frame && !gathered)
        return fail(PT_ERR_INVALID, "pt_trace: PT_SHARED_IMAGE needs iterations that run as one launch (PT_COMPACT, no material sort, no mesh, "
                                    "at most %llu paths per tile)", (unsigned long long)std::max(R.whole_max_paths, R.whole_max_host_paths));
    if (async_image && gathered) {
        // PT_ASYNC_IMAGE and the launch wrote the host image itself: nothing to copy.  The buffer is complete when the launch
        // is; this call returns without waiting for it, but not before the PREVIOUS call's buffer is complete.
        for (int j = 0; j < 2; ++j)
            if (!R.ev_direct[j]) HIPCHK(hipEventCreateWithFlags(&R.ev_direct[j], hipEventDisableTiming));
        hipEvent_t mine = R.ev_direct[R.direct_k];
        R.direct_k ^= 1;
        HIPCHK(hipEventRecord(mine, R.stream));
        if (R.async_prev && R.async_prev != mine) HIPCHK(hipEventSynchronize(R.async_prev));
        R.async_prev = mine;
        return PT_OK;
    }
    if (async_image) return enqueue_async_image(host_image_sum);
    if (host_image_sum && !gathered) {
        rc = enqueue_image_copy(host_image_sum);
        if (rc) return rc;
    }
    return collect_stats();                               // one stream synchronisation covers the copy as well
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
        uint32_t span = 0;                                  // slots per range, as the bounce that packed the pool wrote it down
        const bool packed = (R.flags & PT_COMPACT) && R.cur_dir >= 0;
        if (packed) {
            const size_t nrp = ((size_t)tile_dir(R.cur_dir).nr + 3) & ~(size_t)3;
            HIPCHK(hipMemcpy(&span, tile_dir(R.cur_dir).mem + 3 * nrp + 8, 4, hipMemcpyDeviceToHost));
        }
        hipLaunchKernelGGL(k_export_paths, dim3((n + 255) / 256), dim3(256), 0, R.stream, R.pool[R.cur], R.map, n,
                           live, R.trace_depth - R.step_depth, (pt_path_segment *)R.scratch,
                           tile_dir(packed ? R.cur_dir : -1), span);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(host_paths, R.scratch, (size_t)n * sizeof(pt_path_segment), hipMemcpyDeviceToHost, R.stream));
        HIPCHK(hipStreamSynchronize(R.stream));
    }
    if (n_live) *n_live = (int)live;
    return (int)n;
}

int pt_export_intersections(pt_shadeable_intersection *host_isects, uint8_t *host_outside, int capacity) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_export_intersections: not initialised");
    if (!(R.flags & (PT_UNFUSED | PT_SORT_MATERIAL | PT_FAKE_SHADER)) || R.sort_keys)
        return fail(PT_ERR_INVALID, "pt_export_intersections: intersections are only materialised with PT_UNFUSED (also beside "
                                    "PT_SORT_MATERIAL: its two-kernel form) or PT_FAKE_SHADER");
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
                           R.isect, n, (pt_shadeable_intersection *)R.scratch, host_outside ? d_out : (uint8_t *)nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(host_isects, R.scratch, (size_t)n * sizeof(pt_shadeable_intersection), hipMemcpyDeviceToHost, R.stream));
        if (host_outside) HIPCHK(hipMemcpyAsync(host_outside, d_out, n, hipMemcpyDeviceToHost, R.stream));
        HIPCHK(hipStreamSynchronize(R.stream));
    }
    return (int)n;
}

int pt_intersect_once(const pt_path_segment *host_paths, int n, pt_shadeable_intersection *host_isects,
                      uint8_t *host_outside) {
    R.ov_active = false;
    if (!R.live) return fail(PT_ERR_INVALID, "pt_intersect_once: not initialised");
    if (n < 0 || (uint32_t)n > R.cap) return fail(PT_ERR_INVALID, "pt_intersect_once: n=%d exceeds the pool capacity %u", n, R.cap);
    if (n == 0) return PT_OK;
    if (!host_paths || !host_isects) return fail(PT_ERR_INVALID, "pt_intersect_once: null buffer");
    int rc = ensure_scratch((size_t)n * (sizeof(pt_path_segment) + 1) + 64);
    if (rc) return rc;
    rc = ensure_isect();
    if (rc) return rc;
    R.in_step = false;
    HIPCHK(hipMemcpyAsync(R.scratch, host_paths, (size_t)n * sizeof(pt_path_segment), hipMemcpyHostToDevice, R.stream));
    hipLaunchKernelGGL(k_import_paths, dim3((n + 255) / 256), dim3(256), 0, R.stream, R.pool[0],
                       (const pt_path_segment *)R.scratch, (uint32_t)n);
    HIPCHK(hipGetLastError());
    launch_intersect(R.pool[0], nullptr, (uint32_t)n, tile_dir(-1), nullptr);
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
    if (R.copy_stream) HIPCHK(hipStreamSynchronize(R.copy_stream));
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
    R.image_epoch++;
    return PT_OK;
}

// Resume an accumulation: the running sum is the whole state the reference carries between iterations (dev_image,
// pathtrace.cu:71,84,389).  Everything in flight comes first: batches still tracing add into the buffer being replaced.
int pt_set_image(const float *host_image_sum) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_set_image: not initialised");
    if (!host_image_sum) return fail(PT_ERR_INVALID, "pt_set_image: null buffer");
    HIPCHK(hipStreamSynchronize(R.stream));
    if (R.copy_stream) HIPCHK(hipStreamSynchronize(R.copy_stream));
    HIPCHK(hipMemcpy(R.image, host_image_sum, (size_t)R.npix * 12, hipMemcpyHostToDevice));
    R.ov_active = false;
    R.image_epoch++;
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

int pt_get_bvh_info(pt_bvh_info *out) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_get_bvh_info: not initialised");
    if (R.mesh_mode != MESH_BVH) return fail(PT_ERR_INVALID, "pt_get_bvh_info: PT_MESH_BVH is off or the scene has no mesh");
    if (out) *out = R.bvh_info;
    return PT_OK;
}

int pt_bvh_build(const pt_triangle *triangles, int count, float *nodes, int node_capacity, int32_t *order, float *grid) {
    if (count < 0 || (count > 0 && !triangles)) return fail(PT_ERR_INVALID, "pt_bvh_build: bad triangle list");
    ptbvh::Tree tree;
    ptbvh::build(reinterpret_cast<const float *>(triangles), count, tree);
    if (tree.num_nodes() > node_capacity || !nodes) return tree.num_nodes();
    memcpy(nodes, tree.nodes.data(), tree.nodes.size() * 4);
    if (order && count > 0) memcpy(order, tree.order.data(), (size_t)count * 4);
    if (grid) { for (int a = 0; a < 3; ++a) { grid[a] = tree.origin[a]; grid[3 + a] = tree.step[a]; } grid[6] = tree.pad; grid[7] = tree.prune; }
    return tree.num_nodes();
}

int pt_tri_bounds(const pt_triangle *triangles, int count, float origin_bound, float *bounds) {
    if (count < 0 || (count > 0 && !triangles) || !bounds) return fail(PT_ERR_INVALID, "pt_tri_bounds: bad argument");
    make_tri_bounds(triangles, count, (double)origin_bound, bounds);
    return (count + 3) & ~3;
}

int pt_cull_boxes(const pt_geom *geoms, int count, const float *eye, float *boxes, float *origin_bound, float *reject) {
    if (count < 0 || (count > 0 && !geoms) || !boxes) return fail(PT_ERR_INVALID, "pt_cull_boxes: bad argument");
    std::vector<const float *> inv((size_t)std::max(1, count));
    std::vector<char> sph((size_t)std::max(1, count)), skip((size_t)std::max(1, count));
    for (int i = 0; i < count; ++i) {
        inv[(size_t)i] = &geoms[i].inverseTransform.m[0][0];
        sph[(size_t)i] = geoms[i].type == PT_SPHERE;
        skip[(size_t)i] = geoms[i].type == PT_TRIANGLE_MESH;
    }
    const double e[3] = {eye ? (double)eye[0] : 0.0, eye ? (double)eye[1] : 0.0, eye ? (double)eye[2] : 0.0};
    std::vector<ptcull::Box> bx;
    const float r = ptcull::make_boxes(inv.data(), reinterpret_cast<const bool *>(sph.data()),
                                       reinterpret_cast<const bool *>(skip.data()), count, e, 1, bx);
    for (int i = 0; i < count; ++i)
        for (int k = 0; k < 3; ++k) { boxes[6 * i + k] = bx[(size_t)i].lo[k]; boxes[6 * i + 3 + k] = bx[(size_t)i].hi[k]; }
    if (origin_bound) *origin_bound = r;
    if (reject)
        for (int i = 0; i < count; ++i) {
            float row[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            const int ax = geoms[i].type == PT_CUBE ? ptcull::reject_row(&geoms[i].inverseTransform.m[0][0], row) : 3;
            reject[5 * i] = (float)ax;
            for (int k = 0; k < 4; ++k) reject[5 * i + 1 + k] = row[k];
        }
    return PT_OK;
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

}  // namespace one

#include "pt_probe.hpp"
#include "pt_multi.hpp"

// diagnostics (not in include/ptmi355.h): the block counts an instrumented kernel build has added up since pt_init or the
// last call (profiles/tools/isa_count.py); reads and clears.  Single-device sessions.
extern "C" int ptdbg_counts(unsigned int *out, int words) {
    if (!g_single.live || !g_single.dbg_counts || words < 0 || (size_t)words > g_single.dbg_words) return -1;
    if (hipStreamSynchronize(g_single.stream) != hipSuccess) return -1;
    if (hipMemcpy(out, g_single.dbg_counts, (size_t)words * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (hipMemset(g_single.dbg_counts, 0, g_single.dbg_words * 4) != hipSuccess) return -1;
    return words;
}

#ifdef PT_WAVE_TIMES
// diagnostic build only (not in include/ptmi355.h): per-wave start / end ticks and hardware ids of k_bounce's last launches
extern "C" int ptdbg_wave_times(unsigned long long *times /* [8][8192][2] */, uint32_t *hw /* [8][8192] */) {
    if (hipMemcpyFromSymbol(times, HIP_SYMBOL(g_wave_times), sizeof(unsigned long long) * 8 * 8192 * 2) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_wave_hw), sizeof(uint32_t) * 8 * 8192) != hipSuccess) return -1;
    return 0;
}
#endif
