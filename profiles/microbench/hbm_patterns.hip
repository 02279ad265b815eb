// hbm_patterns.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the ACCESS PATTERNS of k_bounce (round 4,
// VERDICT r03 item 3).  MI355X_MICROARCH.md calibrates the counters for 16-B-per-lane streaming only ("other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute");
// k_bounce moves its path pool as DWORD-per-lane rows of 256 B (ten per 64-path tile), appends survivors as partial
// rows and drops final colours as scattered 16-B stores.  Every kernel below moves a known number of bytes in one of
// those patterns over buffers far larger than the 256-MiB Infinity Cache; run it once under
//     rocprofv3 --pmc FETCH_SIZE -- ./hbm_patterns      and once under      rocprofv3 --pmc WRITE_SIZE -- ./hbm_patterns
// and divide (profiles/tools/hbm_calibrate.py).  The program prints the algorithmic bytes of every kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -o hbm_patterns hbm_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(r_)); exit(1); } } while (0)

constexpr int TILE_BYTES = 2560;          // ten 256-B rows: csrc/pt_types.hpp Pool

__device__ __forceinline__ char *slot(float *base, uint32_t s) {
    return reinterpret_cast<char *>(base) + (size_t)(s >> 6) * TILE_BYTES + ((s & 63u) << 2);
}

// 1/2: a wave reads the ten rows of 64 consecutive SLOTS starting at slot `tile * 64 + off` (off = 0: whole aligned
// rows; off = 20: every row is 44 dwords of one physical tile and 20 of the next, as a logical tile of a packed pool is)
__global__ __launch_bounds__(256) void k_read_rows(float *pool, uint32_t tiles, uint32_t off, float *sink) {
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63, waves = gridDim.x * 4;
    float acc = 0.0f;
    for (uint32_t t = wave; t < tiles; t += waves) {
        char *p = slot(pool, t * 64 + off + lane);
#pragma unroll
        for (int k = 0; k < 10; ++k) acc += *reinterpret_cast<float *>(p + k * 256);
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// 3: the guide's calibrated case, 16 B per lane streaming
__global__ __launch_bounds__(256) void k_read_16(const float4 *src, size_t n4, float *sink) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) sink[0] = acc;
}
// 4/5: a wave writes rows: `keep` of every 64 paths survive and are APPENDED at the wave's running offset in its own
// span (keep = 64: whole rows), ten rows each -- the survivor stores of tile_shade
__global__ __launch_bounds__(256) void k_write_rows(float *pool, uint32_t tiles_per_wave, uint32_t keep) {
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    uint32_t packed = 0;
    const uint32_t base = wave * tiles_per_wave * 64;
    for (uint32_t t = 0; t < tiles_per_wave; ++t) {
        if (lane < keep) {
            char *p = slot(pool, base + packed + lane);
#pragma unroll
            for (int k = 0; k < 10; ++k) *reinterpret_cast<float *>(p + k * 256) = (float)(t + k);
        }
        packed += keep;
    }
}
// 5b/5c/5d: the survivor stores as k_bounce really issues them -- the surviving lanes are SCATTERED over the wave (here: a
// hash keeps ~70 %), their destinations are consecutive (packed + rank among the survivors).  keys = 1: one stream, the
// plain kernel.  keys = 4: every survivor carries a material key, destination = key's own span + the wave's count for
// that key + rank among the tile's survivors with that key -- four interleaved streams, the material-keyed kernel
// (PT_SORT_MATERIAL, fused form).  staged: the tile's survivors are first sorted by key through the wave's LDS strip, so
// that lane i stores the i-th survivor in (key, lane) order and every stream is a run of ADJACENT lanes.
__device__ __forceinline__ uint32_t rank_below64(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
template <int KEYS, bool STAGED>
__global__ __launch_bounds__(256) void k_write_keyed(float *pool, uint32_t tiles_per_wave, uint32_t key_stride, unsigned long long *written) {
    __shared__ float stage[4][11 * 64];
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float *st = stage[threadIdx.x >> 6];
    uint32_t have[KEYS];
#pragma unroll
    for (int k = 0; k < KEYS; ++k) have[k] = 0;
    const uint32_t base = wave * tiles_per_wave * 64;
    unsigned long long total = 0;
    for (uint32_t t = 0; t < tiles_per_wave; ++t) {
        uint32_t h = (wave * 8191u + t * 64u + lane) * 2654435761u;
        h ^= h >> 13;
        const bool alive = (h % 10u) < 7u;
        const int key = KEYS == 1 ? 0 : (int)((h >> 20) % (uint32_t)KEYS);
        uint32_t dst = 0, pos = 0, before = 0;
#pragma unroll
        for (int k = 0; k < KEYS; ++k) {
            const uint64_t m = __builtin_amdgcn_ballot_w64(alive && key == k);
            if (alive && key == k) { dst = (uint32_t)k * key_stride + base + have[k] + rank_below64(m); pos = before + rank_below64(m); }
            have[k] += (uint32_t)__popcll((unsigned long long)m);
            before += (uint32_t)__popcll((unsigned long long)m);
        }
        total += before;
        float f[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) f[k] = (float)(t + k) + (float)lane;
        if (!STAGED) {
            if (alive) {
                char *p = slot(pool, dst);
#pragma unroll
                for (int k = 0; k < 10; ++k) *reinterpret_cast<float *>(p + k * 256) = f[k];
            }
        } else {
            if (alive) {
#pragma unroll
                for (int k = 0; k < 10; ++k) st[k * 64 + pos] = f[k];
                st[10 * 64 + pos] = __uint_as_float(dst);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < before) {
                char *p = slot(pool, __float_as_uint(st[10 * 64 + lane]));
#pragma unroll
                for (int k = 0; k < 10; ++k) *reinterpret_cast<float *>(p + k * 256) = st[k * 64 + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (lane == 0) atomicAdd(written, total);
}
// 6: final colours: one 16-B store per ENDING path with a non-zero colour, index = pid.  `every`: one lane in `every`
// stores (pids ascend with the lane: neighbours in the wave are `every` entries apart on average, hashed a little)
__global__ __launch_bounds__(256) void k_write_fin(float4 *fin, size_t n, uint32_t every, uint32_t salt) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t h = (uint32_t)i * 2654435761u + salt;
        h ^= h >> 15;
        if (h % every == 0) fin[i] = make_float4(1.0f, 2.0f, 3.0f, __uint_as_float(salt));
    }
}
// 7: k_gather's side: read float4[pid] for every pid (16 B per lane, streaming) -- same as 3 on the fin buffer
// 8: dword-per-lane full-row streaming store (256 B per wave instruction), the plain case for WRITE_SIZE
__global__ __launch_bounds__(256) void k_write_dword(float *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (float)i;
}
__global__ __launch_bounds__(256) void k_read_dword(const float *src, size_t n, float *sink) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    const uint32_t tiles = 1u << 20;                       // 2.56 GiB pool: ten times the Infinity Cache
    const size_t pool_bytes = (size_t)(tiles + 2) * TILE_BYTES;
    float *pool, *sink;
    float4 *fin;
    const size_t nfin = (size_t)64 << 20;                  // 1 GiB of float4 entries (64 spp x 800 x 800 = 41 M; 64 M here)
    CHECK(hipMalloc(&pool, pool_bytes));
    CHECK(hipMalloc(&fin, nfin * 16));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(pool, 0, pool_bytes));
    CHECK(hipMemset(fin, 0, nfin * 16));
    CHECK(hipDeviceSynchronize());
    unsigned long long *written;
    CHECK(hipMalloc(&written, 64));
    CHECK(hipMemset(written, 0, 64));
    const int grid = 256 * 6;
    const uint32_t waves = grid * 4;
    const uint32_t tpw = tiles / waves;
    auto mb = [](double b) { return b / 1e6; };
    for (int rep = 0; rep < 2; ++rep) {                   // every kernel twice: two dispatches per name in the counter files
        hipLaunchKernelGGL(k_read_rows, dim3(grid), dim3(256), 0, 0, pool, tiles, 0u, sink);
        hipLaunchKernelGGL(k_read_rows, dim3(grid), dim3(256), 0, 0, pool, tiles, 20u, sink);
        hipLaunchKernelGGL(k_read_16, dim3(grid), dim3(256), 0, 0, (const float4 *)pool, (size_t)tiles * TILE_BYTES / 16, sink);
        hipLaunchKernelGGL(k_read_dword, dim3(grid), dim3(256), 0, 0, (const float *)pool, (size_t)tiles * TILE_BYTES / 4, sink);
        hipLaunchKernelGGL(k_write_rows, dim3(grid), dim3(256), 0, 0, pool, tpw, 64u);
        hipLaunchKernelGGL(k_write_rows, dim3(grid), dim3(256), 0, 0, pool, tpw, 45u);
        hipLaunchKernelGGL(k_write_dword, dim3(grid), dim3(256), 0, 0, pool, (size_t)tiles * TILE_BYTES / 4);
        hipLaunchKernelGGL((k_write_keyed<1, false>), dim3(grid), dim3(256), 0, 0, pool, tpw, 0u, written + 0);
        hipLaunchKernelGGL((k_write_keyed<1, true>), dim3(grid), dim3(256), 0, 0, pool, tpw, 0u, written + 1);
        hipLaunchKernelGGL((k_write_keyed<4, false>), dim3(grid), dim3(256), 0, 0, pool, tpw / 4, waves * (tpw / 4) * 64, written + 2);
        hipLaunchKernelGGL((k_write_keyed<4, true>), dim3(grid), dim3(256), 0, 0, pool, tpw / 4, waves * (tpw / 4) * 64, written + 3);
        hipLaunchKernelGGL(k_write_fin, dim3(grid), dim3(256), 0, 0, fin, nfin, 5u, 1u + rep);
        hipLaunchKernelGGL(k_write_fin, dim3(grid), dim3(256), 0, 0, fin, nfin, 1u, 3u + rep);
        hipLaunchKernelGGL(k_read_16, dim3(grid), dim3(256), 0, 0, (const float4 *)fin, nfin, sink);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipGetLastError());
    // algorithmic bytes per dispatch, in dispatch order of one repetition
    size_t fin5 = 0;
    for (size_t i = 0; i < nfin; ++i) { uint32_t h = (uint32_t)i * 2654435761u + 1u; h ^= h >> 15; fin5 += (h % 5u == 0); }
    printf("ALGO read_rows_aligned      read_MB %.1f write_MB 0\n", mb((double)tiles * TILE_BYTES));
    printf("ALGO read_rows_offset20     read_MB %.1f write_MB 0\n", mb((double)tiles * TILE_BYTES));
    printf("ALGO read_16B_pool          read_MB %.1f write_MB 0\n", mb((double)tiles * TILE_BYTES));
    printf("ALGO read_dword_pool        read_MB %.1f write_MB 0\n", mb((double)tiles * TILE_BYTES));
    printf("ALGO write_rows_keep64      read_MB 0 write_MB %.1f\n", mb((double)waves * tpw * 64 * 40));
    printf("ALGO write_rows_keep45      read_MB 0 write_MB %.1f\n", mb((double)waves * tpw * 45 * 40));
    printf("ALGO write_dword_pool       read_MB 0 write_MB %.1f\n", mb((double)tiles * TILE_BYTES));
    unsigned long long wr[4];
    CHECK(hipMemcpy(wr, written, 32, hipMemcpyDeviceToHost));
    printf("ALGO write_sparse70_1key     read_MB 0 write_MB %.1f\n", mb((double)wr[0] / 2 * 40));
    printf("ALGO write_sparse70_1key_lds read_MB 0 write_MB %.1f\n", mb((double)wr[1] / 2 * 40));
    printf("ALGO write_sparse70_4key     read_MB 0 write_MB %.1f\n", mb((double)wr[2] / 2 * 40));
    printf("ALGO write_sparse70_4key_lds read_MB 0 write_MB %.1f\n", mb((double)wr[3] / 2 * 40));
    printf("ALGO write_fin_one_in_5     read_MB 0 write_MB %.1f\n", mb((double)fin5 * 16));
    printf("ALGO write_fin_all          read_MB 0 write_MB %.1f\n", mb((double)nfin * 16));
    printf("ALGO read_16B_fin           read_MB %.1f write_MB 0\n", mb((double)nfin * 16));
    return 0;
}
