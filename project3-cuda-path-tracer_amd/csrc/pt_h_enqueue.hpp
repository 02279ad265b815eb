// pt_h_enqueue.hpp -- launch plans: a batch's kernels enqueued bounce by bounce, as one launch, or overlapped on the lanes
// (one of the host-side headers of libptmi355.so, included by ptmi355.hip -- the only translation unit -- in dependency order)
#pragma once

namespace {

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
        // this bounce's flags are spent, and zero again: the fused kernel's waves clear every word they read (run_tiles).
        // The array is the NEXT bounce's output flags.  Only a fused bounce marks the candidates of the next one (the
        // cached / unfused pipelines leave the finding to k_mesh's scan, and never touch the flags)
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
    if (R.la_tracing) {
        // a window traced ahead of the caller (PT_LOOKAHEAD): its final colours stay where they are until the calls that
        // consume them (k_gather_one, one sample each); only its counters are folded, on the lane's own stream
        hipLaunchKernelGGL(k_gather, dim3(1), dim3(BLOCK), 0, R.stream, R.image, R.final_mem, R.cap, R.map, R.step_count, R.ctl,
                           R.persist, R.trace_depth, 0u, R.whole ? 1 : 0, 1, R.fin_serial, R.iter_counts, (uint32_t)R.grid_iter_cur,
                           (HostStats *)nullptr);
        R.whole = false;
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(R.lane_cur->traced, R.stream));
        R.lane_cur->gathered_valid = false;
        return PT_OK;
    }
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
        if (l.la_stream && k < R.ov_streams) { (void)hipStreamSynchronize(l.la_stream); (void)hipStreamDestroy(l.la_stream); }
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
    if (R.la_gstream) { (void)hipStreamSynchronize(R.la_gstream); (void)hipStreamDestroy(R.la_gstream); R.la_gstream = nullptr; }
    if (R.la_rs_event) { (void)hipEventDestroy(R.la_rs_event); R.la_rs_event = nullptr; }
    R.la_masks_ready = false; R.la_masked_last = false;
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

// PT_LOOKAHEAD with a host image: compute units set aside for the calls' gathers.  A call's ~37 000 PCIe writes, issued by
// the whole chip within microseconds, queue in the memory system for the ~36 us the link needs, and the windows traced
// meanwhile run at 0.8 of their speed behind them; issued by la_cus = 24 compute units (three of each XCD: the first
// la_cus bits of a CU mask, profiles/r06/cu_mask_census.txt) they arrive at about the link's rate, and the windows -- on
// the other 232 through streams masked the other way, their persistent grid sized for those -- lose less than the 24 CUs
// cost: 0.060-0.061 against 0.068-0.072 ms per call at 800x800, 0.40 against 0.47 at 3840x2160.  Without
// a host image the masks only cost (-9 % on C3's and C5's frames): such calls keep the plain streams.  Sessions whose
// windows need k_mesh's grid (PT_MESH_BVH), devices that are not 256 compute units in 8 XCDs, and a runtime that refuses
// the masks take the plain streams as well.
bool ensure_la_masks(void) {
    if (R.la_masks_ready) return true;
    if (R.la_masks_failed || !R.ov_ready || R.la_cus < 8 || R.cus != 256 || R.mesh_mode == MESH_BVH) return false;
    uint32_t trace_mask[8], gather_mask[8];
    for (int b = 0; b < 8; ++b) { trace_mask[b] = 0xffffffffu; gather_mask[b] = 0; }
    for (int b = 0; b < R.la_cus; ++b) { trace_mask[b / 32] &= ~(1u << (b % 32)); gather_mask[b / 32] |= 1u << (b % 32); }
    bool ok = hipExtStreamCreateWithCUMask(&R.la_gstream, 8, gather_mask) == hipSuccess &&
              hipEventCreateWithFlags(&R.la_rs_event, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; ok && k < R.ov_streams; ++k) ok = hipExtStreamCreateWithCUMask(&R.lane[k].la_stream, 8, trace_mask) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (int k = 0; k < R.ov_streams; ++k) if (R.lane[k].la_stream) { (void)hipStreamDestroy(R.lane[k].la_stream); R.lane[k].la_stream = nullptr; }
        if (R.la_gstream) { (void)hipStreamDestroy(R.la_gstream); R.la_gstream = nullptr; }
        if (R.la_rs_event) { (void)hipEventDestroy(R.la_rs_event); R.la_rs_event = nullptr; }
        R.la_masks_failed = true;
        return false;
    }
    for (int k = R.ov_streams; k < R.ov_lanes; ++k) R.lane[k].la_stream = R.lane[k % R.ov_streams].la_stream;
    R.grid_la = (int)std::min<uint32_t>((uint32_t)R.grid, (uint32_t)(R.cus - R.la_cus) * (uint32_t)std::max(1, R.per_cu));
    R.la_masks_ready = true;
    return true;
}

// the fused pipelines only: the unfused / two-kernel-sort / fake-shader ones keep intersection planes and sort tables
// (one set), the first-bounce cache is filled by the first batch that needs it
bool overlap_eligible(int count) {
    return R.ov_ok && R.ov_enabled && !R.capturing && !R.use_graphs && !R.profiling && !R.epi_host && !R.dbg_counts &&
           !(R.flags & (PT_UNFUSED | PT_FAKE_SHADER | PT_CACHE_FIRST)) && (!(R.flags & PT_SORT_MATERIAL) || R.sort_keys > 0) &&
           count >= 1 && count <= R.max_batch;
}

int enqueue_batch_serial(int iter0, int count);

// before a batch goes to a lane: the stamp's wrap, and whatever the launch stream holds
int enter_lanes(void) {
    if (R.fin_serial == 0xffffffffu) {      // the stamp is about to wrap: nothing may be in flight while every lane's colours are forgotten
        HIPCHK(hipStreamSynchronize(R.stream));
        for (int j = 0; j < R.ov_lanes; ++j) HIPCHK(hipStreamSynchronize(R.lane[j].stream));
        for (int j = 0; j < R.ov_lanes; ++j) if (R.lane[j].la_stream) HIPCHK(hipStreamSynchronize(R.lane[j].la_stream));
        for (int j = 0; j < R.ov_lanes; ++j) HIPCHK(hipMemsetAsync(R.lane[j].b.final_mem, 0, R.final_bytes, R.stream));
        HIPCHK(hipStreamSynchronize(R.stream));
        R.fin_serial = 0;
        R.ov_active = false;
        for (auto &w : R.la) { w.valid = false; w.inflight = false; }      // (their colours are gone)
    }
    if (!R.ov_active) {
        // whatever the launch stream holds (uploads, masks, serial batches on the session's buffers) comes first
        HIPCHK(hipEventRecord(R.ov_enter, R.stream));
        for (int k = 0; k < R.ov_lanes; ++k) {
            HIPCHK(hipStreamWaitEvent(R.lane[k].stream, R.ov_enter, 0));
            if (R.lane[k].la_stream && k < R.ov_streams) HIPCHK(hipStreamWaitEvent(R.lane[k].la_stream, R.ov_enter, 0));
            R.lane[k].gathered_valid = false;
        }
    }
    return PT_OK;
}

// one batch on lane `l`: its buffers and stream stand in for the session's while the serial enqueue code runs
int enqueue_on_lane(Renderer::Lane &l, int iter0, int count) {
    if (l.gathered_valid) HIPCHK(hipStreamWaitEvent(l.stream, l.gathered, 0));
    const Renderer::Bufs home = take_bufs();
    R.lane_main = R.stream; R.lane_cur = &l;
    put_bufs(l.b); R.stream = l.stream;
    const int rc = enqueue_batch_serial(iter0, count);                  // its gather goes to the launch stream (enqueue_end)
    R.stream = R.lane_main; put_bufs(home);
    R.lane_cur = nullptr; R.lane_main = nullptr;
    R.ov_active = rc == PT_OK;
    return rc;
}

int enqueue_batch_direct(int iter0, int count) {
    if (!overlap_eligible(count)) return enqueue_batch_serial(iter0, count);
    int rc = ensure_lanes();
    if (rc) return rc;
    if (!R.ov_enabled) return enqueue_batch_serial(iter0, count);       // the lanes do not fit the budget
    if (R.la_masked_last) { R.ov_active = false; R.la_masked_last = false; }     // (windows on the lanes' masked streams: la_discard has put the launch stream behind them)
    rc = enter_lanes();
    if (rc) return rc;
    Renderer::Lane &l = R.lane[R.ov_next];
    R.ov_next = (R.ov_next + 1) % R.ov_lanes;
    return enqueue_on_lane(l, iter0, count);
}

// PT_LOOKAHEAD: iterations [iter0, iter0 + count) traced on lane `slot` as one pool, nothing gathered (enqueue_end under
// R.la_tracing): the window's final colours wait in the lane's buffer for the calls that consume them.  The caller has
// made sure that nothing still reads that lane's buffers (its previous window is consumed or discarded).
// `masked`: on the lane's stream of the tracing compute units, with the persistent grid that fits them (ensure_la_masks).  A
// lane's two streams are ordered against each other through the launch stream: whoever changes the kind starts from
// enter_lanes' event (every batch's gather, and every window some call has consumed or discarded, is behind that).
int enqueue_window(int slot, int iter0, int count, bool masked) {
    if (masked != R.la_masked_last) R.ov_active = false;
    int rc = enter_lanes();
    if (rc) return rc;
    Renderer::Lane &l = R.lane[slot];
    const hipStream_t plain = l.stream;
    const int grid = R.grid;
    if (masked) { l.stream = l.la_stream; R.grid = R.grid_la; }
    R.la_tracing = true;
    rc = enqueue_on_lane(l, iter0, count);
    R.la_tracing = false;
    l.stream = plain; R.grid = grid;
    R.la_masked_last = masked;
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

}  // namespace
