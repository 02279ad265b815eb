// pt_h_session.hpp -- one context's state (Renderer), error reporting, event profiling, small helpers
// (one of the host-side headers of libptmi355.so, included by ptmi355.hip -- the only translation unit -- in dependency order)
#pragma once

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
            return fail(PT_ERR_DEVICE, "HIP error (%s:%d): %s: %s", __builtin_strrchr(__FILE__, '/') ? __builtin_strrchr(__FILE__, '/') + 1 : __FILE__, __LINE__, #expr, \
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
    float *d_tri_bound = nullptr;  // every-triangle loop, stage 1: the triangles' 64-byte records for the matrix pipe (upload_tri_bounds)
    size_t tri_bound_words = 0;
    std::vector<float> grec_frames;   // per geom {g, 1 / Rm} of a mesh's records (device copy: geom record words G_INV + 7 .. + 10)
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
        hipStream_t la_stream = nullptr;          // PT_LOOKAHEAD: its stream of the tracing compute units (ensure_la_masks)
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
    // PT_LOOKAHEAD: windows of consecutive iterations traced ahead of pt_trace's caller (pt_h_api.hpp: la_*).  A ring of
    // LA_SLOTS windows, slot j on lane j (consecutive slots on alternating launch streams): la_cur is the window being
    // consumed, the slots after it hold the windows that follow it -- up to LA_AHEAD of them traced ahead, overlapping on
    // the two launch streams like asynchronous batches; `next` = the sample the next consecutive call consumes; a window is
    // only ever served to calls whose camera / depth / lens are byte-equal to what it was traced with.  Three ahead on four
    // lanes measured 0.066 ms per call against 0.070 for two (and 0.054 against 0.056 without a host image); six lanes with
    // four or five ahead: the same as three (profiles/r06/lookahead_ab_windows_ahead.txt).
    static constexpr int LA_SLOTS = 4, LA_AHEAD = 3;
    struct LaWindow {
        bool valid = false;       // enqueued and not yet consumed or discarded
        bool inflight = false;    // enqueued, and the launch stream has not been ordered behind its last launch yet
        int iter0 = 0, count = 0, next = 0;
        uint32_t stamp = 0;       // of its final colours
        pt_camera cam{}; int depth = 0; Lens lens{0, 0.0f, 0.0f};
        Control *ctl = nullptr;   // its lane's control block (statistics of the window)
        bool masked = false;      // traced on the lane's masked stream
    } la[LA_SLOTS];
    int la_cur = 0;               // the slot being consumed
    // Calls that write a host image: their gathers run on la_cus compute units of their own (la_gstream, CU-masked), the
    // windows on the others (Lane::la_stream, a persistent grid of grid_la workgroups) -- DESIGN 6.13.  0: never.
    int la_cus = 24, grid_la = 0, per_cu = 0;
    bool la_masks_ready = false, la_masks_failed = false;
    uint64_t la_masked_windows = 0, la_masked_calls = 0;
    bool la_masked_last = false;  // the last batch on a lane went to its masked stream (the next one on a plain stream orders itself behind the launch stream)
    hipStream_t la_gstream = nullptr;
    hipEvent_t la_rs_event = nullptr;
    bool la_tracing = false;      // the batch being enqueued is a window: no k_gather, only its counters (enqueue_end)
    uint64_t la_misses = 0, la_windows = 0, la_discards = 0;   // calls that had to trace their own window first / windows enqueued / windows thrown away
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


}  // namespace
