// pt_k_common.hpp -- wave-level helpers shared by every kernel: ballots and ranks, the priority rotation, the final-colour store
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

// wave64 ballot straight from the compare (HIP's __ballot() goes through select 0/1 + compare-not-equal)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// number of set bits of the wave mask m below this lane: v_mbcnt_lo + v_mbcnt_hi on the scalar mask (the generic
// popcount(m & ((1 << lane) - 1)) compiles to two ands and two bit counts on per-lane copies of the mask)
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// the lanes of a wave mask as a per-lane predicate, for free (the mask becomes the exec mask of the branch)
__device__ __forceinline__ bool lane_of(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// The instruction arbiter serves the OLDEST wave of a SIMD first.  In a persistent grid whose waves all have the same
// amount of work that is the worst order: measured on k_bounce (per-wave start / end times, 5 waves per SIMD), the wave
// in slot 0 -- the first-dispatched fifth of the workgroups -- ended at 0.55-0.7 of the launch, the one in slot 4 at
// 0.92, and every SIMD spent the last third of every launch with fewer and fewer waves to pick instructions from (mean
// residency 0.71-0.81 of the launch).  Rotating the user priority (s_setprio, which the arbiter ranks above age) with
// the wave's tile counter gives every wave the same share of every level: mean residency 0.84-0.94, C2 +10 %.  (No
// effect in k_mesh -- one 16-wave workgroup per CU, whose waves wait on dependent fetches -- and -2 % in the sorted
// shade kernel, eight short-lived workgroups per CU that wait on memory: not used there.)
// `step`: the wave's loop counter (tiles); `slots`: workgroups per CU of the launch (slot = dispatch order).
#ifndef PT_NO_ROTATE_PRIO
__device__ __forceinline__ void set_priority(uint32_t level) {
    switch (level & 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#else
__device__ __forceinline__ void set_priority(uint32_t) {}
#endif
__device__ __forceinline__ void rotate_priority(uint32_t step, uint32_t slots) {
    set_priority((blockIdx.x * slots) / gridDim.x + step);
}

// Final colour of the path that ends here: one 16-B store into final[pid] = {r, g, b, stamp of this batch} -- and only
// when the colour is not zero.  As three planes (round 1) every ending path dirtied three 32-B sectors to deliver
// 12 B; and four paths in five end with colour 0 (they leave the open box or run out of bounces), which adds nothing to
// the sum (x + 0 = x exactly; the sums are never -0): k_gather takes an entry whose stamp is not this batch's as 0.
// Measured on the sorted C3 pipeline: 58 B of HBM writes per ending path before, the 16-B store and its sector.
__device__ __forceinline__ void put_final(float *fin, uint32_t pid, f3 c, uint32_t stamp) {
    if (!(c.x == 0.0f && c.y == 0.0f && c.z == 0.0f))                       // NaN compares false: written
        reinterpret_cast<float4 *>(fin)[pid] = make_float4(c.x, c.y, c.z, __uint_as_float(stamp));
}
// the stamp of the current batch: a launch argument, or (graph replay: arguments are frozen) Control::keep[0]
__device__ __forceinline__ uint32_t batch_stamp(uint32_t arg, const Control *ctl) { return arg ? arg : ctl->keep[0]; }

}  // namespace
