// valu_peak.hip -- what ONE gfx950 SIMD issues per cycle, measured (VERDICT r01 "settle the VALU roof").
//
// Every CU runs k workgroups of 256 threads (k waves per SIMD; residency forced by the dynamic-LDS request:
// k requests fit a CU, k + 1 do not), each wave executes N x 32 independent instructions of one kind between
// two s_memtime / s_memrealtime stamps.  Reported per kind and k: SIMD cycles per wave-instruction over the span
// from the first stamp to the last (all waves), the waves actually in flight per SIMD (sum of wave durations /
// span), what ONE wave sustains (its own stamped cycles per instruction), the in-kernel clock (s_memtime /
// s_memrealtime) and the chip-wide lane-operation rate from the HIP-event wall time.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_peak valu_peak.hip && ./valu_peak > valu_peak.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind { FMA = 0, MUL_ADD, PK_FMA, PK_MUL, RCP, SQRT, MINMAX, CNDMASK, BPERMUTE, FMA64, MUL64, MAD_U64, ADD_U32, DS_READ, CND_SGPR, CMP_CND, MAX3, MED3, FMA_RCP, DS_READ128, DS_WRITE, MBCNT, CVT, KINDS };
static const char *kind_name[KINDS] = {"v_fma_f32", "v_mul_f32+v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_rcp_f32", "v_sqrt_f32",
                                       "v_min_f32+v_max_f32", "v_cndmask_b32", "ds_bpermute_b32", "v_fma_f64", "v_mul_f64",
                                       "v_mad_u64_u32", "v_add_u32", "ds_read_b32", "v_cndmask_b32 (sgpr-pair mask, e64)", "v_cmp_lt_f32+v_cndmask_b32", "v_max3_f32", "v_med3_f32", "3 v_fma_f32 : 1 v_rcp_f32", "ds_read_b128", "ds_write_b32", "v_mbcnt_lo+hi", "v_cvt_f32_u32"};

// 16 independent 32-bit chains a0..a15 (or 8 64-bit ones); one asm statement = 32 instructions
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_issue(unsigned long long *stamps, float *sink, int iters, float seed) {
    extern __shared__ float lds[];
    float a[16];
    double d[8];
    unsigned long long q[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = (double)a[i]; q[i] = (unsigned long long)threadIdx.x + i; }
    const float b = 1.0000001f, c = 1e-9f;
    const double bd = 1.0000001, cd = 1e-9;
    int addr = (threadIdx.x * 4) & 255;
    lds[threadIdx.x] = seed;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (KIND == FMA) {
#define OP(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
            asm volatile(R16(OP) R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
#undef OP
        } else if (KIND == MUL_ADD) {
#define OP(i) "v_mul_f32 %" #i ", %" #i ", %16\n\t"
#define OQ(i) "v_add_f32 %" #i ", %" #i ", %17\n\t"
            asm volatile(R16(OP) R16(OQ) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
#undef OP
#undef OQ
        } else if (KIND == PK_FMA || KIND == PK_MUL) {
            // 8 independent 64-bit (two-float) chains, 32 packed instructions per statement
            float2 p[8], pb = make_float2(b, b), pc = make_float2(c, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = make_float2(a[2 * i], a[2 * i + 1]);
            if (KIND == PK_FMA) {
#define OP(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
                asm volatile(R8(OP) R8(OP) R8(OP) R8(OP) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pb), "v"(pc));
#undef OP
            } else {
#define OP(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n\t"
                asm volatile(R8(OP) R8(OP) R8(OP) R8(OP) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(pb), "v"(pc));
#undef OP
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[2 * i] = p[i].x; a[2 * i + 1] = p[i].y; }
        } else if (KIND == RCP || KIND == SQRT) {
#define OP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define OQ(i) "v_sqrt_f32 %" #i ", %" #i "\n\t"
            if (KIND == RCP)
                asm volatile(R16(OP) R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                             "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
            else
                asm volatile(R16(OQ) R16(OQ) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                             "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
#undef OP
#undef OQ
        } else if (KIND == MINMAX) {
#define OP(i) "v_min_f32 %" #i ", %" #i ", %16\n\t"
#define OQ(i) "v_max_f32 %" #i ", %" #i ", %17\n\t"
            asm volatile(R16(OP) R16(OQ) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
#undef OP
#undef OQ
        } else if (KIND == CNDMASK) {
#define OP(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n\t"
            asm volatile(R16(OP) R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b) : "vcc");
#undef OP
        } else if (KIND == BPERMUTE) {
#define OP(i) "ds_bpermute_b32 %" #i ", %16, %" #i "\n\t"
            asm volatile(R16(OP) "s_waitcnt lgkmcnt(0)\n\t" R16(OP) "s_waitcnt lgkmcnt(0)\n\t"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(addr));
#undef OP
        } else if (KIND == DS_READ) {
#define OP(i) "ds_read_b32 %" #i ", %16\n\t"
            asm volatile(R16(OP) "s_waitcnt lgkmcnt(0)\n\t" R16(OP) "s_waitcnt lgkmcnt(0)\n\t"
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]),
                         "=&v"(a[8]), "=&v"(a[9]), "=&v"(a[10]), "=&v"(a[11]), "=&v"(a[12]), "=&v"(a[13]), "=&v"(a[14]), "=&v"(a[15]) : "v"(addr));
#undef OP
        } else if (KIND == FMA64 || KIND == MUL64) {
#define OP(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n\t"
#define OQ(i) "v_mul_f64 %" #i ", %" #i ", %8\n\t"
            if (KIND == FMA64)
                asm volatile(R8(OP) R8(OP) R8(OP) R8(OP) : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(bd), "v"(cd));
            else
                asm volatile(R8(OQ) R8(OQ) R8(OQ) R8(OQ) : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(bd), "v"(cd));
#undef OP
#undef OQ
        } else if (KIND == MAD_U64) {
#define OP(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n\t"
            unsigned int m0 = 48271u, m1 = (unsigned int)threadIdx.x | 1u;
            asm volatile(R8(OP) R8(OP) R8(OP) R8(OP) : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(m0), "v"(m1) : "vcc");
#undef OP
        } else if (KIND == CND_SGPR) {
#define OP(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, %17\n\t"
            unsigned long long mask = 0x5555aaaa3333ccccull ^ (unsigned long long)iters;
            asm volatile(R16(OP) R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "s"(mask));
#undef OP
        } else if (KIND == CMP_CND) {
#define OP(i) "v_cmp_lt_f32 vcc, %" #i ", %16\n\tv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n\t"
            asm volatile(R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c) : "vcc");
#undef OP
        } else if (KIND == MAX3 || KIND == MED3) {
#define OP(i) "v_max3_f32 %" #i ", %" #i ", %16, %17\n\t"
#define OQ(i) "v_med3_f32 %" #i ", %" #i ", %16, %17\n\t"
            if (KIND == MAX3)
                asm volatile(R16(OP) R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                             "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
            else
                asm volatile(R16(OQ) R16(OQ) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                             "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
#undef OP
#undef OQ
        } else if (KIND == FMA_RCP) {
            // 24 fma + 8 rcp, interleaved 3:1 -- do transcendentals issue beside the plain VALU stream?
#define OP(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n\t"
#define OQ(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
            asm volatile(OP(0) OP(1) OP(2) OQ(12) OP(3) OP(4) OP(5) OQ(13) OP(6) OP(7) OP(8) OQ(14) OP(9) OP(10) OP(11) OQ(15)
                         OP(0) OP(1) OP(2) OQ(12) OP(3) OP(4) OP(5) OQ(13) OP(6) OP(7) OP(8) OQ(14) OP(9) OP(10) OP(11) OQ(15)
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(b), "v"(c));
#undef OP
#undef OQ
        } else if (KIND == DS_READ128) {
            // 8 x ds_read_b128 (lane-contiguous 16-B reads, conflict-free) per statement, counted as 8
            float4 w[8];
            int addr16 = (threadIdx.x * 16) & 1023;
#define OP(i) "ds_read_b128 %" #i ", %8\n\t"
            asm volatile(R8(OP) "s_waitcnt lgkmcnt(0)\n\t" R8(OP) "s_waitcnt lgkmcnt(0)\n\t" R8(OP) "s_waitcnt lgkmcnt(0)\n\t" R8(OP) "s_waitcnt lgkmcnt(0)\n\t"
                         : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]) : "v"(addr16));
#undef OP
            a[0] += w[0].x + w[7].w;
        } else if (KIND == DS_WRITE) {
#define OP(i) "ds_write_b32 %16, %" #i "\n\t"
            asm volatile(R16(OP) R16(OP) "s_waitcnt lgkmcnt(0)\n\t"
                         :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
                         "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(addr) : "memory");
#undef OP
        } else if (KIND == MBCNT) {
#define OP(i) "v_mbcnt_lo_u32_b32 %" #i ", %16, 0\n\tv_mbcnt_hi_u32_b32 %" #i ", %17, %" #i "\n\t"
            unsigned int mlo = 0x5555aaaau ^ (unsigned)iters, mhi = 0x3333ccccu;
            asm volatile(R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "s"(mlo), "s"(mhi));
#undef OP
        } else if (KIND == CVT) {
#define OP(i) "v_cvt_f32_u32 %" #i ", %" #i "\n\t"
            asm volatile(R16(OP) R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
#undef OP
        } else if (KIND == ADD_U32) {
#define OP(i) "v_add_u32 %" #i ", %" #i ", %16\n\t"
            asm volatile(R16(OP) R16(OP) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                         "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]) : "v"(addr));
#undef OP
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)d[i] + (float)q[i];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { stamps[4 * wave] = t1 - t0; stamps[4 * wave + 1] = r1 - r0; stamps[4 * wave + 2] = r0; stamps[4 * wave + 3] = r1; }
    if (s == 12345.678f) sink[0] = s;     // keeps the chains alive
}

template <int KIND>
static void run(int cus, int k, int iters, unsigned long long *d_st, float *d_sink, bool first) {
    const int blocks = cus * k;
    // residency: exactly k workgroups per CU -- k requests fit the CU's 160 KiB with 4 KiB to spare, k + 1 do not
    size_t lds_req = ((size_t)(156 * 1024) / (size_t)k) & ~(size_t)511;
    if (lds_req > 64 * 1024) lds_req = 64 * 1024;            // k = 1, 2: two would fit; the dispatcher spreads 256 / 512 blocks over 256 CUs
    CHK(hipFuncSetAttribute((const void *)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {              // first launch warms up (clock ramp, code fetch)
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), lds_req, 0, d_st, d_sink, iters, 1.0f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st((size_t)blocks * 4 * 4);
    CHK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    unsigned long long rmin = ~0ull, rmax = 0;
    double busy = 0;
    for (int w = 0; w < blocks * 4; ++w) {
        cyc.push_back((double)st[4 * w]); clk.push_back((double)st[4 * w] / (double)st[4 * w + 1] * 100e6);
        rmin = std::min(rmin, st[4 * w + 2]); rmax = std::max(rmax, st[4 * w + 3]);
        busy += (double)st[4 * w + 1];
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double med_cyc = cyc[cyc.size() / 2], med_clk = clk[clk.size() / 2];
    const int per_stmt = (KIND == DS_READ128) ? 32 : 32;
    const double insts = (double)iters * per_stmt;
    const double span_s = (double)(rmax - rmin) / 100e6;                       // first stamp to last stamp, all waves
    const double overlap = busy / 100e6 / span_s / (double)(cus * 4);          // average waves in flight per SIMD
    const double per_simd_span = (double)blocks * 4 * insts / (span_s * med_clk) / (double)(cus * 4);   // wave-insts / cycle / SIMD over the span
    const double lane_ops = (double)blocks * 4 * insts * 64.0 / (ms * 1e-3);
    printf("%s{\"kind\": \"%s\", \"waves_per_simd\": %d, \"waves_in_flight_per_simd\": %.2f, \"cycles_per_wave_inst_per_simd\": %.3f, "
           "\"one_wave_cycles_per_inst\": %.3f, \"in_kernel_clock_ghz\": %.3f, \"chip_lane_ops_per_s_T\": %.2f, \"kernel_ms\": %.3f}",
           first ? "" : ",\n  ", kind_name[KIND], k, overlap, 1.0 / per_simd_span, med_cyc / insts, med_clk / 1e9, lane_ops / 1e12, ms);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 4096;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d_st; float *d_sink;
    CHK(hipMalloc(&d_st, (size_t)cus * 8 * 4 * 4 * 8));
    CHK(hipMalloc(&d_sink, 64));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters\": %d, \"insts_per_wave\": %d, \"rows\": [\n  ", prop.gcnArchName, cus, prop.clockRate / 1000, iters, iters * 32);
    const int ks[] = {1, 2, 4, 5, 8};
    bool first = true;
    for (int k : ks) {
        run<FMA>(cus, k, iters, d_st, d_sink, first); first = false;
        run<MUL_ADD>(cus, k, iters, d_st, d_sink, false);
        run<PK_FMA>(cus, k, iters, d_st, d_sink, false);
        run<PK_MUL>(cus, k, iters, d_st, d_sink, false);
        run<MINMAX>(cus, k, iters, d_st, d_sink, false);
        run<CNDMASK>(cus, k, iters / 4, d_st, d_sink, false);
        run<ADD_U32>(cus, k, iters, d_st, d_sink, false);
        run<RCP>(cus, k, iters, d_st, d_sink, false);
        run<SQRT>(cus, k, iters, d_st, d_sink, false);
        run<FMA64>(cus, k, iters, d_st, d_sink, false);
        run<MUL64>(cus, k, iters, d_st, d_sink, false);
        run<MAD_U64>(cus, k, iters, d_st, d_sink, false);
        run<CND_SGPR>(cus, k, iters / 4, d_st, d_sink, false);
        run<CMP_CND>(cus, k, iters / 4, d_st, d_sink, false);
        run<MAX3>(cus, k, iters, d_st, d_sink, false);
        run<MED3>(cus, k, iters, d_st, d_sink, false);
        run<FMA_RCP>(cus, k, iters, d_st, d_sink, false);
        run<MBCNT>(cus, k, iters, d_st, d_sink, false);
        run<CVT>(cus, k, iters, d_st, d_sink, false);
        run<BPERMUTE>(cus, k, iters / 4, d_st, d_sink, false);
        run<DS_READ>(cus, k, iters / 4, d_st, d_sink, false);
        run<DS_READ128>(cus, k, iters / 4, d_st, d_sink, false);
        run<DS_WRITE>(cus, k, iters / 4, d_st, d_sink, false);
    }
    printf("\n]}\n");
    return 0;
}
