// pt_h_image.hpp -- the caller's host image (page-locking, mapping, copies) and the statistics of a call
// (one of the host-side headers of libptmi355.so, included by ptmi355.hip -- the only translation unit -- in dependency order)
#pragma once

namespace {

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
