// cu_mask_census.hip -- which compute units does a stream created with hipExtStreamCreateWithCUMask really launch on?
// Every workgroup of a census launch records (XCC_ID, HW_ID) of its first wave; the host counts the distinct
// (xcc, se, cu) triples per mask.  hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_census cu_mask_census.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <set>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void census(uint32_t *out, int spin) {
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    // stay resident for a while so that the launch spreads over every unit it may use
    unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long)spin) { }
}

static int run(const char *name, hipStream_t st, uint32_t *d, int wgs) {
    CHK(hipMemsetAsync(d, 0xff, (size_t)wgs * 8, st));
    hipLaunchKernelGGL(census, dim3(wgs), dim3(256), 0, st, d, 200000);
    CHK(hipStreamSynchronize(st));
    std::vector<uint32_t> h((size_t)wgs * 2);
    CHK(hipMemcpy(h.data(), d, (size_t)wgs * 8, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus; int per_xcc[8] = {0};
    std::set<uint32_t> per[8];
    for (int b = 0; b < wgs; ++b) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx94x: se 3 bits)
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const uint32_t key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
        cus.insert(key); per[xcc & 7].insert(key);
    }
    printf("%-34s %4zu distinct compute units; per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", per[x].size());
    printf("\n");
    if (getenv("CENSUS_LIST") && cus.size() <= 64) {          // where they are: xcc.se.sh.cu
        printf("   ");
        for (uint32_t k : cus) printf(" %u.%u.%u.%u", k >> 16, (k >> 8) & 0xff, (k >> 4) & 0xf, k & 0xf);
        printf("\n");
    }
    if (getenv("CENSUS_LIST") && cus.size() > 64) {           // per XCC and shader engine: how many
        printf("    per xcc / se:");
        for (int x = 0; x < 8; ++x) {
            int per_se[8] = {0};
            for (uint32_t k : per[x]) per_se[(k >> 8) & 7]++;
            printf("  [");
            for (int e = 0; e < 8; ++e) if (per_se[e]) printf(" %d:%d", e, per_se[e]);
            printf(" ]");
        }
        printf("\n");
    }
    (void)per_xcc;
    return 0;
}

int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, words = (cus + 31) / 32;
    printf("device: %s, %d compute units\n", p.name, cus);
    uint32_t *d; const int wgs = cus * 8; CHK(hipMalloc(&d, (size_t)wgs * 8));
    hipStream_t plain; CHK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    if (run("no mask", plain, d, wgs)) return 1;
    struct { const char *name; int first, count; } cases[] = {
        {"bits 0..7 set", 0, 8}, {"bits 0..7 cleared", 8, cus - 8}, {"bit 0 set", 0, 1}, {"bits 0..31 set", 0, 32},
        {"bits 32..63 set", 32, 32}, {"bits 0..15 set", 0, 16}, {"bits 0..3 set", 0, 4}, {"bits 0..127 set", 0, 128},
        {"bits 0..23 set", 0, 24}, {"bits 0..23 cleared", 24, cus - 24}};
    for (auto &c : cases) {
        uint32_t mask[16] = {0};
        for (int b = c.first; b < c.first + c.count; ++b) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t st;
        CHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
        if (run(c.name, st, d, wgs)) return 1;
        CHK(hipStreamDestroy(st));
    }
    return 0;
}
