// host_sparse_writes.hip -- what a kernel's stores into page-locked, device-mapped HOST memory cost over PCIe when only a
// fraction of the pixels is written (12 B each, wherever they fall) against all of them in whole 256-B lines (what
// k_iteration's epilogue does today).  640 000 float3 pixels = 7.68 MB.
//   hipcc --offload-arch=gfx950 -O3 -o host_sparse_writes host_sparse_writes.hip && ./host_sparse_writes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct f3 { float x, y, z; };

// every pixel, dword k of the frame by thread k: whole lines
__global__ void k_dense(float *host, const float *src, unsigned n3) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) host[i] = src[i];
}
// flagged pixels only, one 12-B store per pixel
__global__ void k_sparse(f3 *host, const f3 *src, const unsigned char *flag, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (flag[i]) host[i] = src[i];
}
// flagged pixels only, but whole 64-B-aligned groups of 16 B x 4 ... : a pixel's 16-B-aligned neighbourhood (two stores of 8 B cover 12 B + 4 of the neighbour): not bit-safe, timing only
__global__ void k_sparse_dw(float *host, const float *src, const unsigned char *flag, unsigned n) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (flag[i]) { host[3 * i] = src[3 * i]; host[3 * i + 1] = src[3 * i + 1]; host[3 * i + 2] = src[3 * i + 2]; }
}

// round 6: the 64-B lines that hold a flagged pixel, WHOLE (16 lanes x 4 B each, one coalesced line per quarter wave): is a
// full-line write cheaper on the link than the partial one a 12-B store becomes?
__global__ void k_lines(float *host, const float *src, const unsigned char *lineflag, unsigned n3) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x)
        if (lineflag[i >> 4]) host[i] = src[i];
}

int main() {
    const unsigned n = 640000;
    float *host = nullptr, *dhost = nullptr, *src = nullptr;
    unsigned char *flag = nullptr, *lineflag = nullptr;
    CHK(hipHostMalloc((void **)&host, n * 12, hipHostMallocMapped));
    CHK(hipHostGetDevicePointer((void **)&dhost, host, 0));
    CHK(hipMalloc((void **)&src, n * 12));
    CHK(hipMemset(src, 0x3f, n * 12));
    CHK(hipMalloc((void **)&flag, n));
    CHK(hipMalloc((void **)&lineflag, n * 3 / 16 + 16));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grids[] = {256, 1280};
    for (int pct : {100, 50, 20, 10, 6, 5}) {
        std::vector<unsigned char> f(n);
        srand(7);
        unsigned cnt = 0;
        for (unsigned i = 0; i < n; ++i) { f[i] = (rand() % 100) < pct; cnt += f[i]; }
        CHK(hipMemcpy(flag, f.data(), n, hipMemcpyHostToDevice));
        std::vector<unsigned char> lf(n * 3 / 16 + 16, 0);
        unsigned lines = 0;
        for (unsigned i = 0; i < n; ++i) if (f[i]) { lf[(12 * i) >> 6] = 1; lf[(12 * i + 11) >> 6] = 1; }
        for (unsigned char c : lf) lines += c;
        CHK(hipMemcpy(lineflag, lf.data(), lf.size(), hipMemcpyHostToDevice));
        for (int g : grids) {
            float best[4] = {1e9f, 1e9f, 1e9f, 1e9f};
            for (int rep = 0; rep < 6; ++rep) {
                for (int v = 0; v < 4; ++v) {
                    CHK(hipEventRecord(e0));
                    if (v == 0) hipLaunchKernelGGL(k_dense, dim3(g), dim3(256), 0, 0, dhost, src, n * 3);
                    if (v == 1) hipLaunchKernelGGL(k_sparse, dim3(g), dim3(256), 0, 0, (f3 *)dhost, (const f3 *)src, flag, n);
                    if (v == 2) hipLaunchKernelGGL(k_sparse_dw, dim3(g), dim3(256), 0, 0, dhost, src, flag, n);
                    if (v == 3) hipLaunchKernelGGL(k_lines, dim3(g), dim3(256), 0, 0, dhost, src, lineflag, n * 3);
                    CHK(hipEventRecord(e1));
                    CHK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best[v]) best[v] = ms;
                }
            }
            printf("%3d %% of the pixels (%6u, %5.2f MB, %6u lines)  grid %4d:  dense whole frame %6.1f us   sparse 12-B stores %6.1f us   sparse 3 dword stores %6.1f us   whole lines %6.1f us\n",
                   pct, cnt, cnt * 12 / 1e6, lines, g, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3);
        }
    }
    return 0;
}
