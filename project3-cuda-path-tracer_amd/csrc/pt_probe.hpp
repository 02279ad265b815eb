// pt_probe.hpp -- known-answer probes of libptmi355.so (included by ptmi355.hip only): the device functions of
// pt_device.hpp that restate third-party arithmetic the reference merely calls -- thrust's minstd_rand + u01
// (pathtrace.cu:41-45, interactions.h:12-13), the sin / cos binding of interactions.h:40-41 and
// calculateRandomDirectionInHemisphere (interactions.h:10-42) -- run on caller data, so that a test can hold them
// against published constants and against the oracle one function at a time instead of through whole images.
// No session needed: the probes run on the calling thread's current HIP device.
#pragma once

namespace {

__global__ void k_probe_rng(const uint32_t *seeds, int n, int draws, uint32_t *state, float *u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t st = ptd::lcg_seed(seeds[i]);
    float last = 0.0f;
    for (int k = 0; k < draws; ++k) last = ptd::u01(st);
    if (state) state[i] = st;
    if (u) u[i] = last;
}

__global__ void k_probe_sincos(const float *x, uint32_t first_bits, uint32_t n, float *s, float *c, unsigned long long *sum) {
    unsigned long long as = 0, ac = 0;
    // 64-bit index: n may be anything up to 2^32 - 1 (the whole binary32 range) and the stride must not wrap
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const float v = x ? x[k] : __uint_as_float(first_bits + (uint32_t)k);
        float sv, cv;
        ptd::sincos_shared(v, sv, cv);
        if (s) s[k] = sv;
        if (c) c[k] = cv;
        as += (unsigned long long)__float_as_uint(sv) * (2ull * k + 1ull);
        ac += (unsigned long long)__float_as_uint(cv) * (2ull * k + 1ull);
    }
    if (sum) { atomicAdd(&sum[0], as); atomicAdd(&sum[1], ac); }
}

__global__ void k_probe_sqrt(uint32_t first_bits, uint32_t n, unsigned long long *bad) {
    unsigned long long b0 = 0, b1 = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first_bits + (uint32_t)k);
        const float s = ptd::sqrt_normal_range(x);
        // (as normalize_unit chooses: the four-addition form inside its gate)
        const float q = (x >= ptd::NEAR_ONE_LO && x <= ptd::NEAR_ONE_HI) ? ptd::rsqrt_near_one(x) : ptd::rsqrt_of_root(x);
        const float s_ref = __builtin_sqrtf(x);                  // (hipcc: correctly rounded by default)
        const float q_ref = 1.0f / s_ref;
        b0 += __float_as_uint(s) != __float_as_uint(s_ref);
        b1 += __float_as_uint(q) != __float_as_uint(q_ref);
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
}

__global__ void k_probe_hemisphere(const float *normals, const uint32_t *seeds, int n, float *dirs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t st = ptd::lcg_seed(seeds[i]);
    const f3 d = ptd::hemisphere(ptd::mk(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]), st);
    dirs[3 * i] = d.x; dirs[3 * i + 1] = d.y; dirs[3 * i + 2] = d.z;
}

// the shader clock while whatever else is running runs: one wave counts its cycle counter (s_memtime) against the constant
// 100-MHz counter (s_memrealtime) for `ticks` of the latter.  Sixteen scalar registers: it has to fit beside a persistent
// grid that leaves 32 of a SIMD's 800 free (pt_k_image.hpp: k_gather_one).  Ends by itself: the real-time counter advances.
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(16), amdgpu_num_vgpr(32))) void k_probe_clock(unsigned long long *out, unsigned ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64(), c0 = clock64();
    unsigned long long t1 = t0;
    while (t1 - t0 < (unsigned long long)ticks) { __builtin_amdgcn_s_sleep(8); t1 = wall_clock64(); }
    const unsigned long long c1 = clock64();
    out[0] = c1 - c0; out[1] = t1 - t0;
}

// device scratch of one probe call: freed on every exit path
struct ProbeBufs {
    std::vector<void *> mem;
    ~ProbeBufs() { for (void *p : mem) if (p) (void)hipFree(p); }
    void *get(size_t bytes, const void *init) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
        mem.push_back(p);
        if (init ? hipMemcpy(p, init, bytes, hipMemcpyHostToDevice) != hipSuccess : hipMemset(p, 0, bytes ? bytes : 4) != hipSuccess) return nullptr;
        return p;
    }
};

}  // namespace

namespace one {

int pt_probe_rng(const uint32_t *seeds, int n, int draws, uint32_t *state, float *u) {
    if (n < 0 || draws < 0 || (n > 0 && !seeds)) return fail(PT_ERR_INVALID, "pt_probe_rng: bad argument");
    if (n == 0) return PT_OK;
    ProbeBufs b;
    uint32_t *d_seeds = (uint32_t *)b.get((size_t)n * 4, seeds);
    uint32_t *d_state = (uint32_t *)b.get((size_t)n * 4, nullptr);
    float *d_u = (float *)b.get((size_t)n * 4, nullptr);
    if (!d_seeds || !d_state || !d_u) { (void)hipGetLastError(); return fail(PT_ERR_DEVICE, "pt_probe_rng: no HIP device / out of memory (this library has no CPU fallback)"); }
    hipLaunchKernelGGL(k_probe_rng, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_seeds, n, draws, d_state, d_u);
    HIPCHK(hipGetLastError());
    if (state) HIPCHK(hipMemcpy(state, d_state, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (u) HIPCHK(hipMemcpy(u, d_u, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipDeviceSynchronize());
    return PT_OK;
}

int pt_probe_sincos(const float *x, uint32_t first_bits, uint32_t n, float *s, float *c, uint64_t sum[2]) {
    if (n == 0) { if (sum) sum[0] = sum[1] = 0; return PT_OK; }
    // arrays of n floats (inputs or outputs) are bounded; the array-free form (arguments from first_bits, checksums only)
    // may sweep every binary32 value
    if ((x || s || c) && n > (1u << 28)) return fail(PT_ERR_INVALID, "pt_probe_sincos: %u elements with arrays (at most 2^28)", n);
    ProbeBufs b;
    float *d_x = x ? (float *)b.get((size_t)n * 4, x) : nullptr;
    float *d_s = s ? (float *)b.get((size_t)n * 4, nullptr) : nullptr;
    float *d_c = c ? (float *)b.get((size_t)n * 4, nullptr) : nullptr;
    unsigned long long *d_sum = (unsigned long long *)b.get(16, nullptr);
    if ((x && !d_x) || (s && !d_s) || (c && !d_c) || !d_sum) { (void)hipGetLastError(); return fail(PT_ERR_DEVICE, "pt_probe_sincos: no HIP device / out of memory (this library has no CPU fallback)"); }
    const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_probe_sincos, dim3(blocks), dim3(256), 0, 0, d_x, first_bits, n, d_s, d_c, d_sum);
    HIPCHK(hipGetLastError());
    if (s) HIPCHK(hipMemcpy(s, d_s, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (c) HIPCHK(hipMemcpy(c, d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (sum) HIPCHK(hipMemcpy(sum, d_sum, 16, hipMemcpyDeviceToHost));
    HIPCHK(hipDeviceSynchronize());
    return PT_OK;
}

int pt_probe_sqrt(uint32_t first_bits, uint32_t n, uint64_t mismatch[2]) {
    if (!mismatch) return fail(PT_ERR_INVALID, "pt_probe_sqrt: null result");
    mismatch[0] = mismatch[1] = 0;
    if (n == 0) return PT_OK;
    ProbeBufs b;
    unsigned long long *d_bad = (unsigned long long *)b.get(16, nullptr);
    if (!d_bad) { (void)hipGetLastError(); return fail(PT_ERR_DEVICE, "pt_probe_sqrt: no HIP device / out of memory (this library has no CPU fallback)"); }
    const unsigned blocks = (unsigned)std::min<uint64_t>(((uint64_t)n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_probe_sqrt, dim3(blocks), dim3(256), 0, 0, first_bits, n, d_bad);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(mismatch, d_bad, 16, hipMemcpyDeviceToHost));
    HIPCHK(hipDeviceSynchronize());
    return PT_OK;
}

int pt_probe_clock(int microseconds, double *ghz) {
    if (!ghz || microseconds < 1 || microseconds > 100000) return fail(PT_ERR_INVALID, "pt_probe_clock: bad argument");
    *ghz = 0.0;
    unsigned long long *h = nullptr, *d = nullptr;
    hipStream_t st = nullptr;
    int lo = 0, hi = 0;
    if (hipHostMalloc((void **)&h, 16, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void **)&d, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (h) (void)hipHostFree(h);
        return fail(PT_ERR_DEVICE, "pt_probe_clock: no HIP device / no mappable host memory (this library has no CPU fallback)");
    }
    h[0] = h[1] = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
    hipError_t e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_probe_clock, dim3(1), dim3(64), 0, st, d, (unsigned)microseconds * 100u);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);       // (this stream only: whatever else is enqueued keeps running)
        (void)hipStreamDestroy(st);
    }
    const unsigned long long cycles = h[0], ticks = h[1];
    (void)hipHostFree(h);
    if (e != hipSuccess) return fail(PT_ERR_DEVICE, "pt_probe_clock: %s", hipGetErrorString(e));
    if (!ticks) return fail(PT_ERR_INTERNAL, "pt_probe_clock: the probe left no counts");
    *ghz = (double)cycles / ((double)ticks * 10.0);              // cycles per nanosecond
    return PT_OK;
}

int pt_probe_hemisphere(const float *normals, const uint32_t *seeds, int n, float *dirs) {
    if (n < 0 || (n > 0 && (!normals || !seeds || !dirs))) return fail(PT_ERR_INVALID, "pt_probe_hemisphere: bad argument");
    if (n == 0) return PT_OK;
    ProbeBufs b;
    float *d_n = (float *)b.get((size_t)n * 12, normals);
    uint32_t *d_seeds = (uint32_t *)b.get((size_t)n * 4, seeds);
    float *d_d = (float *)b.get((size_t)n * 12, nullptr);
    if (!d_n || !d_seeds || !d_d) { (void)hipGetLastError(); return fail(PT_ERR_DEVICE, "pt_probe_hemisphere: no HIP device / out of memory (this library has no CPU fallback)"); }
    hipLaunchKernelGGL(k_probe_hemisphere, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_n, d_seeds, n, d_d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(dirs, d_d, (size_t)n * 12, hipMemcpyDeviceToHost));
    HIPCHK(hipDeviceSynchronize());
    return PT_OK;
}

}  // namespace one
