// pt_k_image.hpp -- kernels around the image: k_raygen (generateRayFromCamera), k_shade_fake, k_gather (finalGather), k_tonemap (sendImageToPBO), k_copy_out, AoS import / export
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

// ---------------------------------------------------------------------------
// generateRayFromCamera -> SoA pool, `count` samples (stepping interface; the
// batch path generates rays inside bounce 0)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_raygen(Pool p, pt_camera cam, Lens lens, TileMap map, int count,
                                                  int iter0, int trace_depth, Control *ctl) {
    uint32_t total = (uint32_t)map.tile_pixels * (uint32_t)count;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0) { ctl->nlive[0] = total; }
    if (i >= total) return;
    const uint32_t smp = i / (uint32_t)map.tile_pixels;
    const uint32_t j = i - smp * (uint32_t)map.tile_pixels;
    if (iter0 < 0) iter0 = (int)ctl->iter0;                  // graph replay
    f3 o, d;
    camera_ray(cam, lens, trace_depth, iter0 + (int)smp, local_to_pixel(map, (int)j), map.W, o, d);
    const SlotPtr q = p.slot(i);
    pf(q, 0) = o.x; pf(q, 1) = o.y; pf(q, 2) = o.z;
    pf(q, 3) = d.x; pf(q, 4) = d.y; pf(q, 5) = d.z;
    pf(q, 6) = 1.0f; pf(q, 7) = 1.0f; pf(q, 8) = 1.0f;
    ppid(q) = i;
}


// shadeFakeMaterial (pathtrace.cu:224-266): one bounce, never spawns a ray
__global__ __launch_bounds__(BLOCK) void k_shade_fake(Pool p, Isect is, const float *mats_g, TileMap map,
                                                      int iter0, uint32_t n, float *fin, uint32_t stamp_arg, const Control *ctl) {
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
    put_final(fin, pid, c, batch_stamp(stamp_arg, ctl));
}

// finalGather (pathtrace.cu:269-278): image[pixelIndex] += colour, one add per
// pixel per iteration, samples added in iteration order
__global__ __launch_bounds__(BLOCK) void k_gather(float *image, const float *fin, uint32_t cap, TileMap map,
                                                  int count, Control *ctl, Persist *per, int depths,
                                                  uint32_t fake_rays, int partial_counts, int counters_only, uint32_t stamp_arg,
                                                  const uint32_t *iter_counts, uint32_t iter_grid, HostStats *host_stats) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (partial_counts) {            // the batch ran as ONE launch (k_iteration): its per-workgroup counts are added up here
        __shared__ uint32_t fold_lds[BLOCK / 32];
        if (blockIdx.x == 0)
            fold_iter_counts(iter_counts, iter_grid, depths, ctl, per, host_stats, (uint32_t)count, batch_stamp(stamp_arg, ctl), fold_lds);
    } else if (j == 0) {             // fold this batch's ray count into the persistent counters (batches of different lanes may
                                     // run side by side, hence atomics)
        unsigned long long r = fake_rays;
        for (int d = 0; d < depths; ++d) r += ctl->alive[d];
        atomicAdd(&per->rays, r);
        atomicAdd(&per->iterations, (unsigned long long)count);
        atomicAdd(&per->first_rays, (unsigned long long)(depths > 0 ? ctl->alive[0] : fake_rays));
    }
    if (counters_only || j >= (uint32_t)map.tile_pixels) return;
    const int pix = local_to_pixel(map, (int)j);
    float r = image[3 * pix + 0], g = image[3 * pix + 1], b = image[3 * pix + 2];
    // samples are added in iteration order (one add per pixel per iteration, as the reference does); the loads of
    // eight samples are issued together, the adds stay in order
    // an entry counts when it carries this batch's stamp; the others are paths that ended with colour 0 (put_final)
    const uint32_t stamp = batch_stamp(stamp_arg, ctl);
    const float4 *f4 = reinterpret_cast<const float4 *>(fin) + j;
    int s = 0;
    for (; s + 8 <= count; s += 8) {
        float4 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = f4[(size_t)(s + u) * map.tile_pixels];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (__float_as_uint(c[u].w) == stamp) { r += c[u].x; g += c[u].y; b += c[u].z; }
    }
    for (; s < count; ++s) {
        const float4 c = f4[(size_t)s * map.tile_pixels];
        if (__float_as_uint(c.w) == stamp) { r += c.x; g += c.y; b += c.z; }
    }
    image[3 * pix + 0] = r; image[3 * pix + 1] = g; image[3 * pix + 2] = b;
}

// PT_ASYNC_IMAGE: the snapshot of the running sum -> the caller's page-locked image (device-mapped), 16 B per lane,
// whole 256-B lines per quarter wave for the PCIe write combiner; a few workgroups on the copy stream beside the next
// call's tracing (the state.image hand-over of pathtrace.cu:389-390, off the critical path)
__global__ __launch_bounds__(BLOCK) void k_copy_out(float4 *dst, const float4 *src, uint32_t n4, float *dst_tail, const float *src_tail, uint32_t ntail) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n4; i += gridDim.x * BLOCK) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < ntail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// sendImageToPBO (pathtrace.cu:48-68) for one pixel's running sum
__device__ __forceinline__ uchar4 tonemap_pixel(float r, float g, float b, int iter) {
    const float s[3] = {r, g, b};
    int c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double v = (double)(s[k] / (float)iter) * 255.0;
        int q = (int)v;                      // v_cvt_i32_f64: saturating, NaN -> 0
        c[k] = q < 0 ? 0 : (q > 255 ? 255 : q);
    }
    uchar4 o;
    o.x = (unsigned char)c[0]; o.y = (unsigned char)c[1]; o.z = (unsigned char)c[2]; o.w = 0;
    return o;
}

__global__ __launch_bounds__(BLOCK) void k_tonemap(uint8_t *pbo, const float *image, int npix, int iter) {
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= npix) return;
    reinterpret_cast<uchar4 *>(pbo)[i] = tonemap_pixel(image[3 * i + 0], image[3 * i + 1], image[3 * i + 2], iter);
}

// PT_LOOKAHEAD: finalGather (pathtrace.cu:269-278) for ONE iteration of a window traced ahead of the caller.  `fin` is that
// sample's slice of the window's final colours (float4[pixels], index = pixel: such sessions own the whole frame): an entry
// that carries the window's stamp is a path that ended with a non-zero colour -- image[pixel] += colour, the one addition
// per pixel and iteration of the reference, in iteration order because the calls come in iteration order.  The other
// pixels' sums do not change and are not even read, unless a PBO wants every pixel tonemapped (PBO = true).  `host` (the
// caller's page-locked state.image, device-mapped; PT_HOST_SPARSE) receives the sums that changed: ~6 % of the pixels of a
// Cornell iteration, the only bytes of the call that cross PCIe (~30 000 lines of 64 B: ~36 us).
// SIXTEEN scalar registers: this launch runs BESIDE the persistent grid that traces the next windows, and what that grid
// leaves free on a SIMD is not wave slots (2 of 8) or vector registers (32 of 512) but scalar ones -- six waves of
// k_bounce at 106 SGPRs hold 6 x 128 of the 800 (a wave is granted its count rounded up to 16, plus 16); a wave fits
// into the remaining 32 only with at most 16.  With the default allocation (24) every such launch waited for the bounce
// kernel beside it to END: 400-650 us instead of 36 (profiles/r06/lookahead_before_sgpr16.txt).
constexpr uint32_t LA_UNROLL = 2;
template <bool PBO>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(16), amdgpu_num_vgpr(32))) void k_gather_one(
        float *__restrict__ image, const float4 *__restrict__ fin, float *__restrict__ host, uint32_t stamp, uint32_t n,
        uint8_t *__restrict__ pbo, int iter) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t j0 = blockIdx.x * BLOCK + threadIdx.x; j0 < n; j0 += LA_UNROLL * stride) {
        float4 c[LA_UNROLL];
#pragma unroll
        for (int u = 0; u < (int)LA_UNROLL; ++u) {
            const uint32_t j = j0 + (uint32_t)u * stride;
            c[u] = j < n ? fin[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);       // (no entry carries stamp 0)
        }
#pragma unroll
        for (int u = 0; u < (int)LA_UNROLL; ++u) {
            const uint32_t j = j0 + (uint32_t)u * stride;
            const bool changed = __float_as_uint(c[u].w) == stamp;
            if (PBO ? j >= n : !changed) continue;
            float r = image[3 * j + 0], g = image[3 * j + 1], b = image[3 * j + 2];
            if (changed) {
                r += c[u].x; g += c[u].y; b += c[u].z;
                image[3 * j + 0] = r; image[3 * j + 1] = g; image[3 * j + 2] = b;
                if (host) { host[3 * j + 0] = r; host[3 * j + 1] = g; host[3 * j + 2] = b; }
            }
            if (PBO) reinterpret_cast<uchar4 *>(pbo)[j] = tonemap_pixel(r, g, b, iter);
        }
    }
}

// pool <-> reference AoS (debug / parity export and pt_intersect_once)
__global__ void k_export_paths(Pool p, TileMap map, uint32_t n_total, uint32_t n_live, int remaining,
                               pt_path_segment *out, RangeDir dir, uint32_t span) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    uint32_t src = i;
    if (dir.mem) {                        // logical -> physical: largest r with base[r] <= i
        const uint32_t *base = dir.base();
        uint32_t lo = 0, hi = dir.nr - 1;
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

}  // namespace
