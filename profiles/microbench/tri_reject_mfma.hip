// tri_reject_mfma.hip -- pricing the matrix pipe for stage 1 of the loop over every triangle (BASELINE C4 as stated;
// csrc/pt_k_trisweep.hpp).
//
// Stage 1 rejects a (ray, triangle) pair when the ray's LINE passes the triangle's bounding sphere (centre c, radius Rs)
// at more than Rs: today q = c x d - m (d = unit direction, m = o x d), |q|^2 > Rs^2 -- six fma, a three-term dot and a
// compare per pair on the vector pipe: 10 instructions, 4.1e12 pairs/s in the library's kernel.
//
// The same quantity as ONE bilinear form per pair (coordinates relative to the mesh's centre, scaled by its radius):
//     |c x d - m|^2 - Rs^2 = (|c|^2 - Rs^2) - sum_ij c_i c_j d_i d_j - 2 c . w + |m|^2          (w = d x m, |d| = 1)
// = a 11-term product of a per-TRIANGLE vector  [-cx^2 -cy^2 -cz^2 -2cxcy -2cxcz -2cycz | cx cy cz | K | 1]
//   with a per-RAY vector                       [ dx^2  dy^2  dz^2   dxdy   dxdz   dydz | -2wx -2wy -2wz | 1 | M ].
// v_mfma_f32_16x16x32_f16 has K = 32: every term as hi / lo binary16 pairs, three cross products each (hi hi, hi lo, lo hi:
// ~22-bit operands, f32 accumulation) -- 6 x 3 + 3 x 3 + 2 + 2 = 31 slots.  One MFMA = 16 triangles x 16 rays = 256 pairs; what is
// left for the vector pipe is the OR of the results' sign bits (a candidate is rare: ~1e-5 of the pairs).
//
// This file measures pairs/s of both forms on the same synthetic mesh and rays and checks that every pair the fp32 form
// keeps is kept by the MFMA form with its threshold widened by the error bound E (no candidate lost).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tri_reject_mfma tri_reject_mfma.hip && /tmp/tri_reject_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int WAVES = 4, BLOCK = 256;
constexpr float E_WIDEN = 4.0e-5f;           // error bound of the MFMA form in normalised units (31 terms, ~22-bit operands: below)

struct Sphere { float x, y, z, r2; };        // centre relative to the mesh centre / mesh radius, Rs^2 (normalised)

// ---- per-ray line quantities in the mesh's normalised frame ----------------------------------------------------------
__device__ __forceinline__ void line_of(const float *ray /* o.xyz d.xyz */, float &dx, float &dy, float &dz, float &mx, float &my, float &mz) {
    const float ox = ray[0], oy = ray[1], oz = ray[2];
    float ax = ray[3], ay = ray[4], az = ray[5];
    const float s = 1.0f / sqrtf(ax * ax + ay * ay + az * az);
    dx = ax * s; dy = ay * s; dz = az * s;
    mx = oy * dz - oz * dy; my = oz * dx - ox * dz; mz = ox * dy - oy * dx;
}

// ---- today's form: every lane its own ray, spheres wave-uniform from LDS ------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_valu(const float *__restrict__ rays, int nrays, const Sphere *__restrict__ sph, int ntri,
                                               unsigned long long *__restrict__ cand_count, uint32_t *__restrict__ bits) {
    __shared__ float4 stage[WAVES][2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray = (blockIdx.x * WAVES + wv) * 64 + lane;
    float dx, dy, dz, mx, my, mz;
    line_of(rays + 6 * (size_t)min(ray, nrays - 1), dx, dy, dz, mx, my, mz);
    unsigned long long found = 0;
    const float4 *tb = reinterpret_cast<const float4 *>(sph);
    const int ngroups = ntri / 64;
    float4 g_next = tb[lane];
    for (int g = 0; g < ngroups; ++g) {
        float4 *buf = stage[wv][g & 1];
        buf[lane] = g_next;
        if (g + 1 < ngroups) g_next = tb[(size_t)(g + 1) * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const float4 t = buf[j];
            const float qx = __builtin_fmaf(t.y, dz, __builtin_fmaf(-t.z, dy, -mx));
            const float qy = __builtin_fmaf(t.z, dx, __builtin_fmaf(-t.x, dz, -my));
            const float qz = __builtin_fmaf(t.x, dy, __builtin_fmaf(-t.y, dx, -mz));
            const float qq = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
            const bool keep = !(qq > t.w);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (__builtin_expect(m != 0, 0)) {
                found += (unsigned long long)__popcll(m);
                if (bits && keep && ray < nrays) atomicOr(&bits[(size_t)ray * (ntri / 32) + (g * 64 + j) / 32], 1u << ((g * 64 + j) & 31));
            }
        }
    }
    if (lane == 0 && found) atomicAdd(cand_count, found);
}

// ---- the MFMA form ---------------------------------------------------------------------------------------------------------
// slot layout (k = 0..31): a_k on the triangle side, b_k on the ray side; value v -> (hi, lo) binary16; term t occupies three
// slots (a_hi b_hi), (a_hi b_lo), (a_lo b_hi)
__host__ __device__ inline void split16(float v, _Float16 &hi, _Float16 &lo) { hi = (_Float16)v; lo = (_Float16)(v - (float)hi); }

// the 32 triangle-side slots of one triangle (host, at "pt_init")
static void tri_slots(const Sphere &s, _Float16 *a) {
    const float c[3] = {s.x, s.y, s.z};
    const float quad[6] = {-c[0] * c[0], -c[1] * c[1], -c[2] * c[2], -2 * c[0] * c[1], -2 * c[0] * c[2], -2 * c[1] * c[2]};
    int k = 0;
    auto term = [&](float v) { _Float16 h, l; split16(v, h, l); a[k++] = h; a[k++] = h; a[k++] = l; };
    for (int i = 0; i < 6; ++i) term(quad[i]);
    for (int i = 0; i < 3; ++i) term(c[i]);
    {   // K = |c|^2 - Rs^2 + E (the widening rides in the constant), against b = 1: two slots
        _Float16 h, l; split16((c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) - s.r2, h, l); a[k++] = h; a[k++] = l;
    }
    a[k++] = (_Float16)1.0f; a[k++] = (_Float16)1.0f;            // x M (hi, lo)
    a[k++] = (_Float16)0.0f;
}

// the 32 ray-side slots of one ray
__device__ __forceinline__ void ray_slots(float dx, float dy, float dz, float mx, float my, float mz, _Float16 *b) {
    const float wx = dy * mz - dz * my, wy = dz * mx - dx * mz, wz = dx * my - dy * mx;
    const float quad[6] = {dx * dx, dy * dy, dz * dz, dx * dy, dx * dz, dy * dz};
    const float lin[3] = {-2.0f * wx, -2.0f * wy, -2.0f * wz};
    int k = 0;
    auto term = [&](float v) { _Float16 h, l; split16(v, h, l); b[k++] = h; b[k++] = l; b[k++] = h; };
    for (int i = 0; i < 6; ++i) term(quad[i]);
    for (int i = 0; i < 3; ++i) term(lin[i]);
    b[k++] = (_Float16)1.0f; b[k++] = (_Float16)1.0f;            // x K (hi, lo)
    { _Float16 h, l; split16((mx * mx + my * my + mz * mz) - E_WIDEN, h, l); b[k++] = h; b[k++] = l; }   // M - E: keep when val <= E
    b[k++] = (_Float16)0.0f;
}

constexpr int STAGE_TRIS = 64;               // triangles per LDS stage (4 KiB), shared by the workgroup's four waves
__global__ __launch_bounds__(BLOCK) void k_mfma(const float *__restrict__ rays, int nrays, const half8 *__restrict__ tri_a /* [ntri][4] */, int ntri,
                                               unsigned long long *__restrict__ cand_count, uint32_t *__restrict__ bits) {
    __shared__ half8 stage[2][STAGE_TRIS * 4];                   // [triangle][k-block of 8]
    __shared__ _Float16 rb[WAVES][64 * 32];                      // this wave's rays' slots, [ray][32]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray0 = (blockIdx.x * WAVES + wv) * 64;
    {
        float dx, dy, dz, mx, my, mz;
        line_of(rays + 6 * (size_t)min(ray0 + lane, nrays - 1), dx, dy, dz, mx, my, mz);
        _Float16 b[32];
        ray_slots(dx, dy, dz, mx, my, mz, b);
        for (int k = 0; k < 32; ++k) rb[wv][lane * 32 + k] = b[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // B fragments: ray group gI = rays 16 gI .. 16 gI + 15; lane l holds column l & 15, k-block l >> 4
    half8 bf[4];
#pragma unroll
    for (int gI = 0; gI < 4; ++gI) bf[gI] = *reinterpret_cast<const half8 *>(&rb[wv][(16 * gI + (lane & 15)) * 32 + 8 * (lane >> 4)]);
    unsigned long long found = 0;
    const int nstages = ntri / STAGE_TRIS;
    // stage 0
    stage[0][threadIdx.x] = tri_a[threadIdx.x];
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
        half8 nxt;
        if (st + 1 < nstages) nxt = tri_a[(size_t)(st + 1) * STAGE_TRIS * 4 + threadIdx.x];
        const half8 *buf = stage[st & 1];
#pragma unroll
        for (int tg = 0; tg < STAGE_TRIS / 16; ++tg) {
            // A fragment: lane l holds row (triangle) l & 15, k-block l >> 4
            const half8 af = buf[(tg * 16 + (lane & 15)) * 4 + (lane >> 4)];
            float4v acc[4];
#pragma unroll
            for (int gI = 0; gI < 4; ++gI) {
                const float4v z = {0.0f, 0.0f, 0.0f, 0.0f};
                acc[gI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[gI], z, 0, 0, 0);
            }
            // a pair is KEPT when its value is <= 0 (sign bit set, or zero): the common case is "all positive"
            uint32_t any = 0;
#pragma unroll
            for (int gI = 0; gI < 4; ++gI) {
                const uint32_t s01 = __float_as_uint(acc[gI][0]) | __float_as_uint(acc[gI][1]);
                const uint32_t s23 = __float_as_uint(acc[gI][2]) | __float_as_uint(acc[gI][3]);
                any |= s01 | s23;
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64((int)any < 0);
            if (__builtin_expect(m != 0, 0)) {
                // rare: which pairs.  D layout: lane l holds column (ray) l & 15 of group gI, rows (triangles) 4 (l >> 4) + r
#pragma unroll
                for (int gI = 0; gI < 4; ++gI)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (!(acc[gI][r] > 0.0f)) {
                            const int ray = ray0 + 16 * gI + (lane & 15), tri = st * STAGE_TRIS + tg * 16 + 4 * (lane >> 4) + r;
                            if (ray < nrays) {
                                found += 1;                    // (the library would append (ray, triangle) to the wave's candidate ring here)
                                if (bits) atomicOr(&bits[(size_t)ray * (ntri / 32) + tri / 32], 1u << (tri & 31));
                            }
                        }
            }
        }
        if (st + 1 < nstages) stage[(st + 1) & 1][threadIdx.x] = nxt;
        __syncthreads();
    }
    if (found) atomicAdd(cand_count, found);
}

// ---- the MFMA form without a workgroup barrier: every wave fetches its own triangle records (what k_bounce's independent
// waves would do), TILES tiles of 64 rays per pass over the triangles ---------------------------------------------------
template <int TILES>
__global__ __launch_bounds__(BLOCK) void k_mfma_wave(const float *__restrict__ rays, int nrays, const half8 *__restrict__ tri_a, int ntri,
                                                    unsigned long long *__restrict__ cand_count) {
    __shared__ _Float16 rb[WAVES][64 * 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray0 = (blockIdx.x * WAVES + wv) * 64 * TILES;
    half8 bf[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        float dx, dy, dz, mx, my, mz;
        line_of(rays + 6 * (size_t)min(ray0 + 64 * t + lane, nrays - 1), dx, dy, dz, mx, my, mz);
        _Float16 b[32];
        ray_slots(dx, dy, dz, mx, my, mz, b);
        for (int k = 0; k < 32; ++k) rb[wv][lane * 32 + k] = b[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int gI = 0; gI < 4; ++gI) bf[t][gI] = *reinterpret_cast<const half8 *>(&rb[wv][(16 * gI + (lane & 15)) * 32 + 8 * (lane >> 4)]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    unsigned long long found = 0;
    const int ngroups = ntri / 16;
    // lane l fetches row (triangle) l & 15, k-block l >> 4 of group g: 16 B at tri_a[(16 g + (l & 15)) * 4 + (l >> 4)]; four groups in flight
    constexpr int AHEAD = 4;
    half8 af[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) af[u] = tri_a[(size_t)(16 * u + (lane & 15)) * 4 + (lane >> 4)];
    for (int g = 0; g < ngroups; g += AHEAD) {
        half8 nx[AHEAD];
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            const int gn = min(g + AHEAD + u, ngroups - 1);
            nx[u] = tri_a[(size_t)(16 * gn + (lane & 15)) * 4 + (lane >> 4)];
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            uint32_t any = 0;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int gI = 0; gI < 4; ++gI) {
                    const float4v z = {0.0f, 0.0f, 0.0f, 0.0f};
                    const float4v acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], bf[t][gI], z, 0, 0, 0);
                    any |= (__float_as_uint(acc[0]) | __float_as_uint(acc[1])) | (__float_as_uint(acc[2]) | __float_as_uint(acc[3]));
                }
            const unsigned long long m = __builtin_amdgcn_ballot_w64((int)any < 0);
            if (__builtin_expect(m != 0, 0)) found += (unsigned long long)__popcll(m);       // (the library re-runs the group to find the pairs)
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) af[u] = nx[u];
    }
    if (lane == 0 && found) atomicAdd(cand_count, found);
}

int main(int argc, char **argv) {
    const int ntri = 100032 / 64 * 64;       // 100 032 is a multiple of 64 already (BASELINE C4's mesh)
    const int nrays = argc > 1 ? atoi(argv[1]) : 1 << 20;
    const int nverify = 8192;
    // a UV sphere of radius 1 (normalised frame) with ~100 k small triangles: bounding spheres of triangle size
    std::vector<Sphere> sph((size_t)ntri);
    srand(7);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
    for (int i = 0; i < ntri; ++i) {
        const float u = 2.0f * rnd() - 1.0f, ph = 6.2831853f * rnd(), r = sqrtf(1.0f - u * u);
        const float edge = 0.013f * (0.5f + rnd());              // ~ 2 pi / 521 of a unit sphere
        sph[(size_t)i] = Sphere{r * cosf(ph), r * sinf(ph), u, edge * edge};
    }
    // rays: origins a few radii away, directions towards points of the ball of radius 1.3 (most lines pass near the mesh)
    std::vector<float> rays((size_t)nrays * 6);
    for (int i = 0; i < nrays; ++i) {
        float o[3], p[3];
        for (int k = 0; k < 3; ++k) { o[k] = 6.0f * (rnd() - 0.5f); p[k] = 2.6f * (rnd() - 0.5f); }
        for (int k = 0; k < 3; ++k) { rays[(size_t)i * 6 + k] = o[k]; rays[(size_t)i * 6 + 3 + k] = (p[k] - o[k]) * (0.3f + rnd()); }
    }
    std::vector<_Float16> ta((size_t)ntri * 32);
    for (int i = 0; i < ntri; ++i) tri_slots(sph[(size_t)i], &ta[(size_t)i * 32]);

    float *d_rays; Sphere *d_sph; half8 *d_ta; unsigned long long *d_cnt; uint32_t *d_bits_v, *d_bits_m;
    CHK(hipMalloc(&d_rays, rays.size() * 4)); CHK(hipMalloc(&d_sph, sph.size() * sizeof(Sphere))); CHK(hipMalloc(&d_ta, ta.size() * 2));
    CHK(hipMalloc(&d_cnt, 16));
    const size_t bit_words = (size_t)nverify * (ntri / 32);
    CHK(hipMalloc(&d_bits_v, bit_words * 4)); CHK(hipMalloc(&d_bits_m, bit_words * 4));
    CHK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_sph, sph.data(), sph.size() * sizeof(Sphere), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_ta, ta.data(), ta.size() * 2, hipMemcpyHostToDevice));

    // ---- no candidate lost: the first `nverify` rays, every triangle, both forms, bit for bit ----
    CHK(hipMemset(d_bits_v, 0, bit_words * 4)); CHK(hipMemset(d_bits_m, 0, bit_words * 4)); CHK(hipMemset(d_cnt, 0, 16));
    hipLaunchKernelGGL(k_valu, dim3(nverify / 256), dim3(BLOCK), 0, 0, d_rays, nverify, d_sph, ntri, d_cnt, d_bits_v);
    hipLaunchKernelGGL(k_mfma, dim3(nverify / 256), dim3(BLOCK), 0, 0, d_rays, nverify, d_ta, ntri, d_cnt + 1, d_bits_m);
    CHK(hipDeviceSynchronize());
    std::vector<uint32_t> bv(bit_words), bm(bit_words);
    CHK(hipMemcpy(bv.data(), d_bits_v, bit_words * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(bm.data(), d_bits_m, bit_words * 4, hipMemcpyDeviceToHost));
    unsigned long long nv = 0, nm = 0, lost = 0;
    for (size_t i = 0; i < bit_words; ++i) { nv += __builtin_popcount(bv[i]); nm += __builtin_popcount(bm[i]); lost += __builtin_popcount(bv[i] & ~bm[i]); }
    printf("verify: %d rays x %d triangles = %.3g pairs: fp32 form keeps %llu (%.2e of the pairs), MFMA form keeps %llu (%.2fx), kept by fp32 but NOT by MFMA: %llu\n",
           nverify, ntri, (double)nverify * ntri, nv, (double)nv / ((double)nverify * ntri), nm, nv ? (double)nm / nv : 0.0, lost);

    // ---- rates ----
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char *names[4] = {"fp32 form (vector pipe)", "bilinear form (MFMA f16x32)", "MFMA, per-wave records", "MFMA, per-wave, 2 tiles"};
    for (int which = 0; which < 4; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipMemset(d_cnt, 0, 16));
            CHK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k_valu, dim3(nrays / 256), dim3(BLOCK), 0, 0, d_rays, nrays, d_sph, ntri, d_cnt, (uint32_t *)nullptr);
            else if (which == 1) hipLaunchKernelGGL(k_mfma, dim3(nrays / 256), dim3(BLOCK), 0, 0, d_rays, nrays, d_ta, ntri, d_cnt, (uint32_t *)nullptr);
            else if (which == 2) hipLaunchKernelGGL(k_mfma_wave<1>, dim3(nrays / 256), dim3(BLOCK), 0, 0, d_rays, nrays, d_ta, ntri, d_cnt);
            else hipLaunchKernelGGL(k_mfma_wave<2>, dim3(nrays / 512), dim3(BLOCK), 0, 0, d_rays, nrays, d_ta, ntri, d_cnt);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c = 0; CHK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
            printf("%-28s %8.2f ms  %.3e pairs/s  (%llu kept)\n", names[which], ms,
                   (double)nrays * ntri / (ms * 1e-3), c);
        }
    }
    return 0;
}
