// pt_k_pool.hpp -- the range-packed path pool: logical index -> slot (find_range, resolve_src) and the last workgroup's scan of the range counts (stable compaction)
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

// ---- reading a range-packed pool -------------------------------------------------------
// Wave-cooperative 64-ary search: largest r in [0, W) with base[r] <= P (P < base[W]).
__device__ __forceinline__ uint32_t find_range(const uint32_t *base, uint32_t W, uint32_t P) {
    const int lane = threadIdx.x & 63;
    uint32_t lo = 0, hi = W;                        // answer in [lo, hi)
    for (int guard = 0; guard < 8 && hi - lo > 1; ++guard) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + (uint32_t)lane * step;
        const uint32_t v = idx < hi ? base[idx] : 0xffffffffu;
        const uint64_t ok = ballot64(v <= P);       // base[] is non-decreasing: a prefix of the lanes
        const uint32_t k = (uint32_t)__popcll((unsigned long long)ok);
        const uint32_t nlo = lo + (k ? k - 1 : 0) * step;
        hi = min(hi, nlo + step);
        lo = nlo;
    }
    return lo;
}

// Source slots of the 64 logical paths p = p0 + lane, starting the search at range `cur`
// (wave-uniform, base[cur] <= p0).  Lane l first holds base[cur + l]; a 6-step binary search
// reads other lanes' values with ds_bpermute.  Returns the slot; `cur` advances to the range of
// the tile's last path so the next tile of the run starts where this one ended.
__device__ __forceinline__ uint32_t resolve_src(const RangeDir &dir, uint32_t span, uint32_t &cur, uint32_t p,
                                                bool active, Control *ctl) {
    const int lane = threadIdx.x & 63;
    const uint32_t *base = dir.base();
#ifndef PT_NO_RESOLVE_FAST
    {
        // A range holds the survivors of a whole run of tiles (thousands of paths), so a tile almost always lies
        // inside the range the previous tile ended in: two wave-uniform loads and one subtraction then replace the
        // windowed search below.
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p) - (uint32_t)__builtin_amdgcn_readfirstlane(lane);
        const uint32_t b0 = base[cur], b1 = base[cur + 1];              // cur < W always (base[] has W + 1 entries)
        if (p0 >= b0 && p0 + 63u < b1) return cur * span + (p - b0);
    }
#endif
    bool resolved = !active;
    uint32_t src = 0, rng = cur;
    uint32_t s = cur;
    // bounded: every window resolves at least the first unresolved lane; every wave reaches the exit
    for (uint32_t guard = 0;; ++guard) {
        if (guard > 66) {
            if (lane == 0) atomicOr(&ctl->error, 2u);
            break;
        }
        const uint32_t t = s + (uint32_t)lane;
        const uint32_t w = t <= dir.nr ? base[t] : 0xffffffffu;
        int lo = 0, hi = 63;                        // w(lane 0) <= p always holds for unresolved lanes
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi + 1) >> 1;
            const uint32_t wm = (uint32_t)__shfl((int)w, mid);
            if (wm <= p) lo = mid; else hi = mid - 1;
        }
        const uint32_t wl = (uint32_t)__shfl((int)w, lo);
        if (!resolved && lo < 63) { resolved = true; rng = s + (uint32_t)lo; src = rng * span + (p - wl); }
        const uint64_t un = ballot64(!resolved);
        if (!un) break;
        // The next window starts at the range that holds the first unresolved path (paths ascend with the lane), found
        // by the 64-ary search -- not 63 ranges further on: between two keys of a sorted pool lie the W ranges of every
        // material nobody survived on (the light: thousands of empty ranges), and a tile that straddles them walked
        // them window by window -- 80 windows of one dependent load each, 60-150 us at the end of every sorted launch.
        const uint32_t pmin = (uint32_t)__builtin_amdgcn_readlane((int)p, __ffsll((unsigned long long)un) - 1);
        const uint32_t nxt = find_range(base, dir.nr, pmin);
        s = nxt > s ? nxt : s + 63;                 // (always ahead: the first unresolved lane lies past this window)
    }
    // the highest active lane holds the tile's last path
    const uint64_t act = ballot64(active);
    if (act) cur = (uint32_t)__builtin_amdgcn_readlane((int)rng, 63 - __builtin_clzll((unsigned long long)act));
    return src;
}


// ---------------------------------------------------------------------------
// stable compaction: range counts -> range bases, by the last workgroup out
// ---------------------------------------------------------------------------
// Hand-off (guide G16): each wave stores its range count with an agent-scope atomic
// (write-through) store and drains it (s_waitcnt vmcnt(0)); after the workgroup's barrier one
// lane adds 1 to done[depth]; the workgroup whose add returns grid-1 is last, acquires once
// (agent scope) and scans the counts.  Nothing spins; nothing depends
// on dispatch order.
__device__ __forceinline__ void scan_range_counts(const RangeDir &dir, uint32_t *n_out,
                                                  uint32_t *lds_scan /* >= 16 words */, uint32_t span_out, bool with_tiles) {
    // ONE pass over all NR entries (W waves, or K * W ranges when survivors are placed by material): thread t owns the
    // `per4` consecutive uint4s from t * per4, sums them (16-B loads, four in flight), the 256 partial sums cross through one
    // wave scan + one LDS exchange, and the thread writes its prefixes.  (Up to round 4 this ran in steps of 8192 entries
    // with the total carried from step to step: five dependent steps, 12 us per launch, for C3's six materials.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t NR = dir.nr;
    const uint4 *count4 = reinterpret_cast<const uint4 *>(dir.count());
    uint4 *base4 = reinterpret_cast<uint4 *>(dir.base());
    uint4 *tbase4 = reinterpret_cast<uint4 *>(dir.tbase());         // with_tiles: the same scan over ceil(count / 64) (RangeDir)
    uint32_t carry = 0, tcarry = 0;
    {
        const uint32_t per4 = ((NR + BLOCK - 1) / BLOCK + 3) / 4;         // uint4s per thread
        const uint32_t first = threadIdx.x * per4 * 4;                    // this thread's first entry
        // (the counts are read twice -- once for the sums, again, from the L2, for the prefixes -- instead of being held in
        // registers across the barrier: this code runs once per launch in one workgroup, but its registers count
        // against the whole kernel's budget, and at six workgroups per CU there are none to spare)
        uint32_t sum = 0, tsum = 0;
        auto tl = [](uint32_t c) { return (c + (uint32_t)TILE - 1u) / (uint32_t)TILE; };
        auto load = [&](uint32_t k) {
            const uint32_t e = first + 4 * k;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k < per4 && e < NR) {
                v = count4[e >> 2];                                          // count[] is padded to a multiple of 4
                if (e + 1 >= NR) v.y = 0;
                if (e + 2 >= NR) v.z = 0;
                if (e + 3 >= NR) v.w = 0;
            }
            return v;
        };
        for (uint32_t k = 0; k < per4; k += 4) {
            const uint4 v0 = load(k), v1 = load(k + 1), v2 = load(k + 2), v3 = load(k + 3);
            sum += (v0.x + v0.y + v0.z + v0.w) + (v1.x + v1.y + v1.z + v1.w) + (v2.x + v2.y + v2.z + v2.w) + (v3.x + v3.y + v3.z + v3.w);
            tsum += tl(v0.x) + tl(v0.y) + tl(v0.z) + tl(v0.w) + tl(v1.x) + tl(v1.y) + tl(v1.z) + tl(v1.w) +
                    tl(v2.x) + tl(v2.y) + tl(v2.z) + tl(v2.w) + tl(v3.x) + tl(v3.y) + tl(v3.z) + tl(v3.w);
        }
        uint32_t incl = sum, tincl = tsum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off), tu = __shfl_up(tincl, off);
            if (lane >= off) { incl += u; tincl += tu; }
        }
        uint32_t *slot = lds_scan;
        if (lane == 63) { slot[wave] = incl; slot[WAVES + wave] = tincl; }
        __syncthreads();
        uint32_t wave_off = 0, total = 0, twave_off = 0, ttotal = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c = slot[w], tc = slot[WAVES + w];
            if (w < wave) { wave_off += c; twave_off += tc; }
            total += c; ttotal += tc;
        }
        uint32_t run = wave_off + incl - sum, trun = twave_off + tincl - tsum;
        auto put = [&](uint32_t k, const uint4 &v) {
            const uint32_t e = first + 4 * k;
            if (k < per4 && e < NR) {
                uint4 b;
                b.x = run; b.y = b.x + v.x; b.z = b.y + v.y; b.w = b.z + v.z;
                base4[e >> 2] = b;                                           // base[] has 4 spare entries
                run = b.w + v.w;
                if (with_tiles) {
                    uint4 t;
                    t.x = trun; t.y = t.x + tl(v.x); t.z = t.y + tl(v.y); t.w = t.z + tl(v.z);
                    tbase4[e >> 2] = t;
                    trun = t.w + tl(v.w);
                }
            }
        };
        for (uint32_t k = 0; k < per4; k += 4) {
            const uint4 v0 = load(k), v1 = load(k + 1), v2 = load(k + 2), v3 = load(k + 3);
            put(k, v0); put(k + 1, v1); put(k + 2, v2); put(k + 3, v3);
        }
        carry = total; tcarry = ttotal;
    }
    if (threadIdx.x == 0) { dir.base()[NR] = carry; *n_out = carry; if (with_tiles) dir.tbase()[NR] = tcarry; *dir.span() = span_out; }
}

}  // namespace
