#!/usr/bin/env python3
"""Generates issue_ops.hip: the issue cost of EVERY vector opcode the library's kernels execute, measured one opcode at
a time (no opcode of the roofline's issue model may be priced by a guess).

    python3 gen_issue_ops.py > issue_ops.hip
    hipcc --offload-arch=gfx950 -O2 -o issue_ops issue_ops.hip && ./issue_ops > issue_ops.json

Method = profiles/microbench/valu_peak.hip's: every CU runs k workgroups of 256 threads (k waves per SIMD, residency
forced through the dynamic-LDS request), each wave executes ITERS x 32 independent instructions of one opcode between
two s_memtime / s_memrealtime stamps; reported: SIMD cycles per wave-instruction over the span first stamp .. last
stamp.  Operands: 16 independent 32-bit chains (%0..%15), 8 independent 64-bit chains (%16..%23), two 32-bit vector
inputs (%24, %25), one 64-bit vector input (%26), a 32-bit scalar (%27) and a 64-bit scalar mask (%28); compares and
lane reads write vcc / s[40:47] (clobbered)."""
import sys

A, D = "%{a}", "%{d}"
B, C, BD, S32, S64 = "%24", "%25", "%26", "%27", "%28"


def vop1(op): return "%s %s, %s" % (op, A, A)
def vop2(op): return "%s %s, %s, %s" % (op, A, A, B)
def vop3(op): return "%s %s, %s, %s, %s" % (op, A, A, B, C)
def cmp32(op): return "%s vcc, %s, %s" % (op, A, B)
def cmp64s(op): return "%s s[{p}:{q}], %s, %s" % (op, A, B)


OPS = {}
for op in ("v_add_f32 v_sub_f32 v_subrev_f32 v_mul_f32 v_min_f32 v_max_f32 v_and_b32 v_or_b32 v_xor_b32 v_lshlrev_b32 v_lshrrev_b32 "
           "v_ashrrev_i32 v_add_u32 v_sub_u32 v_subrev_u32 v_min_u32 v_max_u32 v_max_i32 v_min_i32 v_mul_u32_u24 v_mul_lo_u32 v_mul_hi_u32 "
           "v_bcnt_u32_b32 v_add_u16 v_lshrrev_b16 v_lshlrev_b16 v_ldexp_f32").split():
    OPS[op] = vop2(op)
for op in ("v_mov_b32 v_not_b32 v_rcp_f32 v_rcp_iflag_f32 v_rsq_f32 v_sqrt_f32 v_cvt_f32_u32 v_cvt_f32_i32 v_cvt_u32_f32 v_cvt_i32_f32 "
           "v_fract_f32 v_floor_f32 v_trunc_f32 v_rndne_f32 v_bfrev_b32 v_ffbh_u32 v_exp_f32 v_log_f32").split():
    OPS[op] = vop1(op)
for op in ("v_fma_f32 v_min3_f32 v_max3_f32 v_med3_f32 v_add3_u32 v_lshl_add_u32 v_add_lshl_u32 v_lshl_or_b32 v_and_or_b32 v_or3_b32 "
           "v_xad_u32 v_bfe_u32 v_bfi_b32 v_alignbit_b32 v_mad_u32_u24 v_mad_i32_i24 v_div_fixup_f32 v_perm_b32").split():
    OPS[op] = vop3(op)
OPS["v_fmac_f32"] = "v_fmac_f32 %s, %s, %s" % (A, B, C)
# fused multiply-add with a 32-bit literal (round 4: the polynomial constants of the binary32 shared sin / cos)
OPS["v_fmamk_f32"] = "v_fmamk_f32 %s, %s, 0x3e2aaaab, %s" % (A, A, B)
OPS["v_fmaak_f32"] = "v_fmaak_f32 %s, %s, %s, 0x3e2aaaab" % (A, A, B)
OPS["v_cndmask_b32_e32"] = "v_cndmask_b32 %s, %s, %s, vcc" % (A, A, B)
OPS["v_cndmask_b32_e64"] = "v_cndmask_b32_e64 %s, %s, %s, %s" % (A, A, B, S64)
OPS["v_div_scale_f32"] = "v_div_scale_f32 %s, vcc, %s, %s, %s" % (A, B, B, A)
OPS["v_div_fmas_f32"] = "v_div_fmas_f32 %s, %s, %s, %s" % (A, A, B, C)
OPS["v_bitop3_b32"] = "v_bitop3_b32 %s, %s, %s, %s bitop3:0x96" % (A, A, B, C)
OPS["v_bitop3_b16"] = "v_bitop3_b16 %s, %s, %s, %s bitop3:0x96" % (A, A, B, C)
for rel in "lt le gt ge eq neq nlt nle ngt nge o u lg nlg".split():
    OPS["v_cmp_%s_f32_e32" % rel] = cmp32("v_cmp_%s_f32" % rel)
    OPS["v_cmp_%s_f32_e64" % rel] = cmp64s("v_cmp_%s_f32_e64" % rel)
for ty in ("u32", "i32", "u16"):
    for rel in "lt le gt ge eq ne".split():
        OPS["v_cmp_%s_%s_e32" % (rel, ty)] = cmp32("v_cmp_%s_%s" % (rel, ty))
        OPS["v_cmp_%s_%s_e64" % (rel, ty)] = cmp64s("v_cmp_%s_%s_e64" % (rel, ty))
OPS["v_cmp_class_f32_e32"] = cmp32("v_cmp_class_f32")
OPS["v_cmp_class_f32_e64"] = cmp64s("v_cmp_class_f32_e64")
for rel in "lt eq ne gt".split():
    OPS["v_cmp_%s_u64_e32" % rel] = "v_cmp_%s_u64 vcc, %s, %s" % (rel, D, BD)
    OPS["v_cmp_%s_u64_e64" % rel] = "v_cmp_%s_u64_e64 s[{p}:{q}], %s, %s" % (rel, D, BD)
OPS["v_readlane_b32"] = "v_readlane_b32 s{p}, %s, 5" % A
OPS["v_readfirstlane_b32"] = "v_readfirstlane_b32 s{p}, %s" % A
OPS["v_writelane_b32"] = "v_writelane_b32 %s, %s, 7" % (A, S32)
OPS["v_mbcnt_lo_u32_b32"] = "v_mbcnt_lo_u32_b32 %s, %s, %s" % (A, S32, A)
OPS["v_mbcnt_hi_u32_b32"] = "v_mbcnt_hi_u32_b32 %s, %s, %s" % (A, S32, A)
# v_cndmask_b32 with vcc costs 22 cycles when NOTHING in the stream writes vcc (r02 and r03 agree: some stall on a vcc
# that only an s_mov has ever written); in the kernels it always follows the compare that produced its mask: priced as
# 2 x (the pair's cycles per instruction) - the compare's
OPS["pair:v_cmp_lt_f32_e32+v_cndmask_b32_e32"] = "v_cmp_lt_f32 vcc, %s, %s\\n\\tv_cndmask_b32 %s, %s, %s, vcc" % (A, B, A, A, C)
# 64-bit chains
OPS["v_fma_f64"] = "v_fma_f64 %s, %s, %s, %s" % (D, D, BD, BD)
OPS["v_mul_f64"] = "v_mul_f64 %s, %s, %s" % (D, D, BD)
OPS["v_add_f64"] = "v_add_f64 %s, %s, %s" % (D, D, BD)
OPS["v_mov_b64"] = "v_mov_b64 %s, %s" % (D, BD)
OPS["v_lshlrev_b64"] = "v_lshlrev_b64 %s, 1, %s" % (D, D)
OPS["v_lshrrev_b64"] = "v_lshrrev_b64 %s, 1, %s" % (D, D)
OPS["v_lshl_add_u64"] = "v_lshl_add_u64 %s, %s, 2, %s" % (D, D, BD)
OPS["v_mad_u64_u32"] = "v_mad_u64_u32 %s, vcc, %s, %s, %s" % (D, B, C, D)
OPS["v_mad_i64_i32"] = "v_mad_i64_i32 %s, vcc, %s, %s, %s" % (D, B, C, D)
OPS["v_pk_fma_f32"] = "v_pk_fma_f32 %s, %s, %s, %s" % (D, D, BD, BD)
OPS["v_pk_mul_f32"] = "v_pk_mul_f32 %s, %s, %s" % (D, D, BD)
OPS["v_pk_add_f32"] = "v_pk_add_f32 %s, %s, %s" % (D, D, BD)
# packed binary16 (round 5: could the conservative cull test two primitives per instruction?  VERDICT r04 item 3 (i))
for op in "v_pk_fma_f16".split():
    OPS[op] = vop3(op)
for op in "v_pk_min_f16 v_pk_max_f16 v_pk_add_f16 v_pk_mul_f16".split():
    OPS[op] = "%s %s, %s, %s" % (op, A, A, B)
OPS["v_cvt_pkrtz_f16_f32"] = "v_cvt_pkrtz_f16_f32 %s, %s, %s" % (A, A, B)
OPS["v_cmp_gt_f16_e32"] = cmp32("v_cmp_gt_f16")
OPS["v_or_b32_sdwa"] = "v_or_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" % (A, A, B)
OPS["v_cvt_f32_u32_sdwa"] = "v_cvt_f32_u32_sdwa %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (A, A)
OPS["v_fma_f32_abs_neg"] = "v_fma_f32 %s, -%s, |%s|, %s" % (A, A, B, C)
OPS["v_cvt_f32_f64"] = "v_cvt_f32_f64 %s, %s" % (A, D)
OPS["v_cvt_i32_f64"] = "v_cvt_i32_f64 %s, %s" % (A, D)
OPS["v_cvt_u32_f64"] = "v_cvt_u32_f64 %s, %s" % (A, D)
OPS["v_cvt_f64_f32"] = "v_cvt_f64_f32 %s, %s" % (D, A)
OPS["v_cvt_f64_u32"] = "v_cvt_f64_u32 %s, %s" % (D, A)
# scalar instructions (round 4: is the scalar unit a second issue roof?  k_bounce executes 0.73 scalar / branch / message
# instructions per vector instruction); four independent chains in s40 / s42 / s44 / s46
OPS["s_add_u32"] = "s_add_u32 s{p}, s{p}, s{q}"
OPS["s_and_b32"] = "s_and_b32 s{p}, s{p}, %s" % S32
OPS["s_lshl_b32"] = "s_lshl_b32 s{p}, s{p}, 1"
OPS["s_mov_b32"] = "s_mov_b32 s{p}, %s" % S32
OPS["s_and_b64"] = "s_and_b64 s[{p}:{q}], s[{p}:{q}], %s" % S64
OPS["s_bcnt1_i32_b64"] = "s_bcnt1_i32_b64 s{p}, %s" % S64
OPS["s_or_saveexec_b64"] = "s_or_saveexec_b64 s[{p}:{q}], exec"
OPS["s_cmp_lg_u32"] = "s_cmp_lg_u32 s{p}, %s" % S32
OPS["s_cselect_b32"] = "s_cselect_b32 s{p}, s{p}, %s" % S32
OPS["s_nop"] = "s_nop 0"
# a scalar and a vector instruction alternating (both streams independent): do they issue side by side?
OPS["pair:v_fma_f32+s_add_u32"] = "v_fma_f32 %s, %s, %s, %s\\n\\ts_add_u32 s{p}, s{p}, s{q}" % (A, A, B, C)
OPS["pair:v_cmp_lt_f32_e64+s_and_b64"] = "v_cmp_lt_f32_e64 s[{p}:{q}], %s, %s\\n\\ts_and_b64 s[{p}:{q}], s[{p}:{q}], %s" % (A, B, S64)
# memory-side instructions the kernels' inner loops lean on (LDS pipe: per CU, shared by the four SIMDs)
OPS["ds_read_b32"] = "ds_read_b32 %s, %s" % (A, "%29")
OPS["ds_read_b128"] = None      # valu_peak.hip measures the LDS forms (they need their own wait structure)


def body(tmpl):
    out = []
    for i in range(16 if "\\n" in tmpl else 32):
        s = tmpl.replace("{a}", str(i % 16)).replace("{d}", str(16 + i % 8))
        s = s.replace("{p}", str(40 + 2 * (i % 4))).replace("{q}", str(41 + 2 * (i % 4)))
        out.append(s + "\\n\\t")
    return '"' + '"\n                         "'.join(out) + '"'


names = [k for k, v in OPS.items() if v and not k.startswith("ds_")]
print("// generated by gen_issue_ops.py -- do not edit")
print(r'''#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int KIND>
__global__ __launch_bounds__(256) void k_issue(unsigned long long *stamps, float *sink, int iters, float seed) {
    extern __shared__ float lds[];
    float a[16];
    double d[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = (double)a[i];
    const float b = 1.0000001f, c = 1e-9f;
    const double bd = 1.0000001;
    const unsigned int s32 = 0x5555aaaau ^ (unsigned)iters;
    const unsigned long long s64 = 0x5555aaaa3333ccccull ^ (unsigned long long)iters;
    lds[threadIdx.x] = seed;
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_mov_b64 vcc, %0" :: "s"(s64) : "vcc");
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#define OPERANDS : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
                   "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), \
                   "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(b), "v"(c), "v"(bd), "s"(s32), "s"(s64) \
                 : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47"''')
for k, name in enumerate(names):
    print("        %sif (KIND == %d) {\n            asm volatile(%s OPERANDS);\n        }" % ("" if k == 0 else "else ", k, body(OPS[name])))
print(r'''    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)d[i];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { stamps[4 * wave] = t1 - t0; stamps[4 * wave + 1] = r1 - r0; stamps[4 * wave + 2] = r0; stamps[4 * wave + 3] = r1; }
    if (s == 12345.678f) sink[0] = s;
}
static bool g_first = true;
static const char *g_filter = nullptr;            // argv[2]: only the opcodes whose name starts with it
template <int KIND>
static void run(const char *name, int cus, int k, int iters, unsigned long long *d_st, float *d_sink) {
    if (g_filter && strncmp(name, g_filter, strlen(g_filter)) != 0) return;
    const int blocks = cus * k;
    size_t lds_req = ((size_t)(156 * 1024) / (size_t)k) & ~(size_t)511;
    if (lds_req > 64 * 1024) lds_req = 64 * 1024;
    CHK(hipFuncSetAttribute((const void *)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_req));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), lds_req, 0, d_st, d_sink, iters, 1.0f);
        CHK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> st((size_t)blocks * 4 * 4);
    CHK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int w = 0; w < blocks * 4; ++w) {
        cyc.push_back((double)st[4 * w]); clk.push_back((double)st[4 * w] / (double)st[4 * w + 1] * 100e6);
        rmin = std::min(rmin, st[4 * w + 2]); rmax = std::max(rmax, st[4 * w + 3]);
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double med_cyc = cyc[cyc.size() / 2], med_clk = clk[clk.size() / 2];
    const double insts = (double)iters * 32;
    const double span_s = (double)(rmax - rmin) / 100e6;
    const double per_simd_span = (double)blocks * 4 * insts / (span_s * med_clk) / (double)(cus * 4);
    printf("%s{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_wave_inst_per_simd\": %.3f, \"one_wave_cycles_per_inst\": %.3f, \"in_kernel_clock_ghz\": %.3f}",
           g_first ? "" : ",\n  ", name, k, 1.0 / per_simd_span, med_cyc / insts, med_clk / 1e9);
    fflush(stdout);
    g_first = false;
}
int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 1024;
    if (argc > 2) g_filter = argv[2];
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d_st; float *d_sink;
    CHK(hipMalloc(&d_st, (size_t)cus * 8 * 4 * 4 * 8));
    CHK(hipMalloc(&d_sink, 64));
    printf("{\"device\": \"%s\", \"cus\": %d, \"iters\": %d, \"rows\": [\n  ", prop.gcnArchName, cus, iters);
    const int ks[] = {2, 5, 8};
    for (int k : ks) {''')
for k, name in enumerate(names):
    print('        run<%d>("%s", cus, k, iters, d_st, d_sink);' % (k, name))
print(r'''    }
    printf("\n]}\n");
    return 0;
}''')
