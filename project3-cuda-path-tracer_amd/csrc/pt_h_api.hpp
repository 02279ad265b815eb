// pt_h_api.hpp -- the entry points of ONE context (namespace one): the C-ABI of include/ptmi355.h applied to the calling thread's context
// (one of the host-side headers of libptmi355.so, included by ptmi355.hip -- the only translation unit -- in dependency order)
#pragma once

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
    // batches and windows may still be running on the lanes (lane 0's buffers are the session's own, freed next)
    for (int k = 0; k < OV_MAX_LANES; ++k) {
        if (R.lane[k].stream) (void)hipStreamSynchronize(R.lane[k].stream);
        if (R.lane[k].la_stream) (void)hipStreamSynchronize(R.lane[k].la_stream);
    }
    if (R.la_gstream) (void)hipStreamSynchronize(R.la_gstream);
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
int la_discard(int how);            // PT_LOOKAHEAD: forget the windows traced ahead (below, with pt_trace)
enum { LA_STREAM = 0, LA_HOST = 1, LA_LATER = 2 };

}  // namespace one

#include "pt_h_scene.hpp"

namespace one {

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
    R.grec_frames.assign((size_t)std::max(1, d->num_geoms) * 4, 0.0f);
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
    R.per_cu = per_cu;
    if (const char *e = pt_experiment("PTMI355_LA_CUS")) R.la_cus = std::max(0, std::min(64, atoi(e) & ~7));
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
    if (memcmp(&R.cam, camera, sizeof R.cam) != 0) {
        R.cache_valid = false; drop_graphs();                  // refill the bounce-0 cache
        // windows traced ahead for the old camera are void, and what follows rewrites masks and boxes their launches read
        const int rc = la_discard(LA_HOST);
        if (rc) return rc;
    }
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
    if (R.flags & PT_LOOKAHEAD) { const int rc = la_discard(LA_STREAM); if (rc) return rc; }
    R.in_step = false;
    R.ov_ok = true;
    const int rc = enqueue_batch(iter0, count);
    R.ov_ok = false;
    return rc;
}

int pt_trace_batch(int iter0, int count, float *host_image_sum) {
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace_batch: not initialised");
    if (R.flags & PT_LOOKAHEAD) { const int rc = la_discard(LA_STREAM); if (rc) return rc; }
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

// ---------------------------------------------------------------------------------------------------------------------
// PT_LOOKAHEAD (include/ptmi355.h): pt_trace traces ahead of its caller.
//
// The reference's host asks for ONE iteration per call (main.cpp:130-140) and wants state.image complete when the call
// returns (pathtrace.cu:389-392).  One iteration is eight DEPENDENT bounces of ~10 us each however few paths are left
// (k_iteration, DESIGN 6.11: 122 us per call, issue 0.50), while 64 iterations traced as one pool cost 48 us each
// (k_bounce).  A path is a function of (iteration, pixelIndex, depth) and nothing else, so the iterations the host is
// going to ask for can be traced before it asks: a WINDOW of consecutive iterations goes through the batched pipeline on
// a lane (enqueue_window), every sample's final colours stay in that lane's buffer, and the call for iteration i runs
// finalGather for sample i alone (k_gather_one) -- the same single float addition per pixel and iteration, in iteration
// order -- with the host-image write and the tonemap folded into that launch.  While window w is consumed, window w + 1
// is traced on the other lane.  Window sizes grow 4, 16, 64, .. up to max_batch: the first image after a camera move
// does not wait for 64 iterations.
// ---------------------------------------------------------------------------------------------------------------------
bool la_possible(void) {
    return (R.flags & PT_LOOKAHEAD) && !(R.flags & (PT_UNFUSED | PT_FAKE_SHADER | PT_CACHE_FIRST | PT_ASYNC_IMAGE)) &&
           (!(R.flags & PT_SORT_MATERIAL) || R.sort_keys > 0) && R.map.tile_count == 1 && R.max_batch >= 2 && R.ov_enabled &&
           !R.use_graphs && !R.profiling && !R.dbg_counts;
}

// Forget the windows.  One that may still be tracing keeps its lane's buffers busy -- lane 0's are the session's own --
// so whatever is enqueued on the launch stream afterwards has to wait for it: LA_STREAM (the default).  LA_HOST: the HOST
// waits, before it rewrites scene records the launches in flight read (pt_set_camera).  LA_LATER: nothing waits yet --
// la_trace itself, which goes on with another window on the lanes (stream order does the rest); the window stays
// marked `inflight` for whoever comes next.
int la_discard(int how) {
    for (int j = 0; j < Renderer::LA_SLOTS; ++j) {
        Renderer::LaWindow &w = R.la[j];
        if (w.valid) { w.valid = false; R.la_discards++; }
        if (!w.inflight || how == LA_LATER || !R.ov_ready) continue;
        if (how == LA_HOST) HIPCHK(hipStreamSynchronize(w.masked ? R.lane[j].la_stream : R.lane[j].stream));
        else HIPCHK(hipStreamWaitEvent(R.stream, R.lane[j].traced, 0));
        w.inflight = false;
    }
    return PT_OK;
}

static bool la_matches(const Renderer::LaWindow &w, int iter) {
    return w.valid && iter == w.iter0 + w.next && w.depth == R.trace_depth && memcmp(&w.cam, &R.cam, sizeof w.cam) == 0 &&
           memcmp(&w.lens, &R.lens, sizeof w.lens) == 0;
}

static int la_trace_window(int slot, int iter0, int count, bool masked) {
    count = (int)std::min<int64_t>((int64_t)count, (int64_t)0x7fffffff - (int64_t)iter0 + 1);       // enqueue_begin's range of iteration numbers
    const int rc = enqueue_window(slot, iter0, count, masked);
    if (rc) return rc;
    Renderer::LaWindow &w = R.la[slot];
    w.valid = true; w.inflight = true; w.iter0 = iter0; w.count = count; w.next = 0; w.stamp = R.fin_serial;
    w.cam = R.cam; w.depth = R.trace_depth; w.lens = R.lens; w.ctl = R.last_ctl; w.masked = masked;
    R.la_windows++;
    if (masked) R.la_masked_windows++;
    return PT_OK;
}

// the windows that follow the one being consumed: up to LA_AHEAD of them, each four times its predecessor's size up to
// max_batch, on the ring's next slots -- overlapping on the two launch streams, like asynchronous batches
static int la_trace_ahead(void) {
    const Renderer::LaWindow *last = &R.la[R.la_cur];
    for (int k = 1; k <= Renderer::LA_AHEAD; ++k) {
        const int slot = (R.la_cur + k) % Renderer::LA_SLOTS;
        Renderer::LaWindow &w = R.la[slot];
        const int64_t iter0 = (int64_t)last->iter0 + last->count;
        if (!w.valid) {
            if (iter0 > 0x7fffffff) break;
            const int rc = la_trace_window(slot, (int)iter0, (int)std::min<int64_t>((int64_t)R.max_batch, (int64_t)last->count * 4), R.la[R.la_cur].masked);
            if (rc) return rc;
        }
        last = &w;
    }
    return PT_OK;
}

// `handled`: the call was served here (else the caller goes on with the plain path)
int la_trace(uint8_t *pbo_rgba, int iter, float *host_image_sum, bool *handled) {
    *handled = false;
    if (!la_possible() || iter < 1) return la_discard(LA_STREAM);
    int rc = ensure_lanes();
    if (rc) return rc;
    if (!R.ov_enabled || R.ov_lanes < Renderer::LA_SLOTS) return PT_OK;             // the lanes do not fit: the plain path
    *handled = true;
    // the host image: the launch writes the sums that changed into the caller's page-locked buffer when that buffer holds
    // exactly the accumulation buffer's content as of the previous call (PT_HOST_SPARSE); otherwise every pixel is copied
    float *mapped = host_image_sum ? map_host(host_image_sum, (size_t)R.npix * 12) : nullptr;
    const bool host_current = mapped && R.host_sparse_enabled && R.own_image && R.host_synced == mapped && R.host_epoch == R.image_epoch;
    if (!la_matches(R.la[R.la_cur], iter)) {
        // not the next sample of the window being consumed (whose successors on the ring continue it, so none of them
        // starts at `iter` either): everything traced ahead is void.  The new chain of windows goes to the lanes' masked
        // streams when its calls are going to write a host image (ensure_la_masks); a lane's two streams know nothing of
        // each other, so a change of kind waits for what is in flight (through the launch stream: enqueue_window)
        const bool masked = mapped && R.host_sparse_enabled && R.own_image && ensure_la_masks();
        rc = la_discard(masked != R.la_masked_last ? LA_STREAM : LA_LATER);
        if (rc) return rc;
        rc = la_trace_window(R.la_cur, iter, std::min(R.max_batch, 4), masked);
        if (rc) return rc;
        R.la_misses++;
        if (masked) {                                     // the gathers' stream starts behind whatever the launch stream holds
            HIPCHK(hipEventRecord(R.la_rs_event, R.stream));
            HIPCHK(hipStreamWaitEvent(R.la_gstream, R.la_rs_event, 0));
        }
    }
    Renderer::LaWindow &w = R.la[R.la_cur];
    const int s = w.next;
    Renderer::Lane &lane = R.lane[R.la_cur];
    // a call that writes the host image itself launches on the gathers' compute units when its window was traced on the
    // others; every call ends with its stream drained, so consecutive calls may use different ones
    hipStream_t gs = (w.masked && host_current) ? R.la_gstream : R.stream;
    if (w.inflight) {
        HIPCHK(hipStreamWaitEvent(R.stream, lane.traced, 0));
        if (gs != R.stream) HIPCHK(hipStreamWaitEvent(gs, lane.traced, 0));
        w.inflight = false;
    }
    if (gs != R.stream) R.la_masked_calls++;
    if (host_current && R.dma_last) { HIPCHK(hipStreamWaitEvent(gs, R.dma_last, 0)); R.dma_last = nullptr; }
    {
        const float4 *fin = reinterpret_cast<const float4 *>(lane.b.final_mem) + (size_t)s * (size_t)R.npix;
        dim3 grid((unsigned)((R.npix + (int)LA_UNROLL * BLOCK - 1) / ((int)LA_UNROLL * BLOCK)));
        // on its own compute units: a round and a half of what they hold at once (a grid-stride kernel; 0.058 against 0.060 ms with one round)
        if (gs != R.stream) grid = dim3((unsigned)std::min<int>((int)grid.x, R.la_cus * (pt_experiment("PTMI355_LA_GWGS") ? std::max(1, atoi(pt_experiment("PTMI355_LA_GWGS"))) : 12)));
        float *host_dev = host_current ? mapped : (float *)nullptr;
        if (pbo_rgba) hipLaunchKernelGGL(k_gather_one<true>, grid, dim3(BLOCK), 0, gs, R.image, fin, host_dev, w.stamp, (uint32_t)R.npix, pbo_rgba, iter);
        else hipLaunchKernelGGL(k_gather_one<false>, grid, dim3(BLOCK), 0, gs, R.image, fin, host_dev, w.stamp, (uint32_t)R.npix, pbo_rgba, iter);
    }
    HIPCHK(hipGetLastError());
    R.image_epoch++;
    if (host_image_sum && !host_current) {
        rc = enqueue_image_copy(host_image_sum);
        if (rc) return rc;
    }
    if (mapped && R.own_image) { R.host_synced = mapped; R.host_epoch = R.image_epoch; }
    w.next++;
    if (s == 0) {
        // as soon as a window starts being consumed: the windows after it (while the gather just launched runs)
        rc = la_trace_ahead();
        if (rc) return rc;
        // the window's statistics, once: its counters were folded on its lane before `traced` was recorded
        R.last_ctl = w.ctl; R.host_stats_serial = 0; R.step_count = w.count; R.step_iter0 = w.iter0;
        rc = collect_stats();
        if (gs != R.stream) HIPCHK(hipStreamSynchronize(gs));
    } else {
        HIPCHK(hipStreamSynchronize(gs));
        R.stats.bounces = 0; R.stats.rays = 0;
        memset(R.stats.live, 0, sizeof R.stats.live);
    }
    if (w.next >= w.count) { w.valid = false; R.la_cur = (R.la_cur + 1) % Renderer::LA_SLOTS; }     // consumed: on to the window traced meanwhile
    return rc;
}

int pt_trace(uint8_t *pbo_rgba, int frame, int iter, float *host_image_sum) {
    (void)frame;                                          // unused in the reference too (main.cpp:136)
    if (!R.live) return fail(PT_ERR_INVALID, "pt_trace: not initialised");
    R.in_step = false;
    if (R.flags & PT_LOOKAHEAD) {
        bool handled = false;
        const int rc = la_trace(pbo_rgba, iter, host_image_sum, &handled);
        if (rc || handled) return rc;
    }
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
    if (R.flags & PT_LOOKAHEAD) { const int rc = la_discard(LA_STREAM); if (rc) return rc; }
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
    if (R.flags & PT_LOOKAHEAD) { const int rc = la_discard(LA_STREAM); if (rc) return rc; }
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
    if (R.flags & PT_LOOKAHEAD) { const int rc = la_discard(LA_STREAM); if (rc) return rc; }
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
    if (R.flags & PT_LOOKAHEAD) { const int rc = la_discard(LA_STREAM); if (rc) return rc; }
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

int pt_tri_records(const pt_triangle *triangles, int count, float origin_bound, uint16_t *records, float frame[4]) {
    if (count < 0 || (count > 0 && !triangles) || !records || !frame) return fail(PT_ERR_INVALID, "pt_tri_records: bad argument");
    const size_t n64 = (size_t)((count + 63) & ~63);
    std::vector<float> sph(std::max<size_t>(n64, 1) * 4, 0.0f);
    make_tri_bounds(triangles, count, (double)origin_bound, sph.data());
    for (size_t i = (size_t)((count + 3) & ~3); i < n64; ++i) sph[i * 4 + 3] = -1.0f;
    static_assert(sizeof(_Float16) == sizeof(uint16_t), "binary16");
    make_tri_records(sph.data(), (int)n64, reinterpret_cast<_Float16 *>(records), frame);
    return (int)n64;
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
