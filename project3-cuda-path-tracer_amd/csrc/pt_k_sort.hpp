// pt_k_sort.hpp -- material sort, two-kernel form: k_sort_hist, k_shade_sorted, k_shade_sorted_w (INSTRUCTION.md:78-86)
// (one of the kernel-family headers of libptmi355.so, included by pt_kernels.hpp in dependency order; ptmi355.hip is the
// only translation unit)
#pragma once

namespace {

// ---------------------------------------------------------------------------
// material sort (INSTRUCTION.md:78-86; spec 8.0): stable sort of the live paths and their
// intersections by key = materialId (misses last) before shading; the pool order after the
// bounce is the stable partition of that sorted order
// ---------------------------------------------------------------------------
// k_intersect materialises the intersections of the (dense) pool.  Then
//   k_sort_hist   : every WORKGROUP histograms the keys of its contiguous run of 512-path chunks into
//                   table[key][workgroup] -- only the keys whose paths go on (with compaction a key either survives
//                   as a whole or not at all: miss, emissive material and the last bounce end a path, nothing
//                   else does); the last workgroup out scans the table in place: table[key][g] becomes the position
//                   in the OUTPUT pool of workgroup g's first path with that key.
//   k_shade_sorted: every workgroup walks its run again, chunk by chunk: a stable counting sort of the chunk's 512
//                   keys in LDS (per-wave counts -> starts, no data moves), then each wave takes 128 consecutive
//                   SORTED positions, gathers their state and intersection from the chunk's 30 KB of pool rows (every
//                   line the gathers touch is consumed by the same workgroup), shades them -- lanes of a wave run the
//                   same material's code except where two keys meet -- and writes the survivors straight to their
//                   place in the globally sorted, compacted output pool (runs of consecutive slots per key).
// The sort therefore costs one extra read of the keys (8 B per path); nothing is moved to be sorted.  r01 moved
// state + intersection (60 B per path) with fifteen scattered 4-B stores, then read them back: 2.5 TB/s, 43 % of
// the time of a C3 step.
constexpr int SORT_MAX_BINS = 2048;          // one bin per material + misses; per-wave chunk counts live in LDS (32 KiB at the limit)
constexpr int SORT_TPW = 2;                  // 64-path tiles per wave and chunk (1 and 4 measured: -2 % / -5 %)
constexpr int SORT_CHUNK_TILES = SORT_TPW * WAVES;
constexpr int SORT_CHUNK = SORT_CHUNK_TILES * TILE;

__device__ __forceinline__ uint32_t sort_key(const Isect &is, uint32_t i, int nbins) {
    const float t = is.plane(0)[i];
    const int m = is.mat()[i] & 0x7fffffff;
    return t > 0.0f ? (uint32_t)m : (uint32_t)(nbins - 1);
}
// does a path whose intersection has this key go on to the next bounce?  (ptd::shade_scatter's three exits)
__device__ __forceinline__ bool key_survives(const float *__restrict__ mats, uint32_t key, int nbins, bool last_bounce) {
    if (last_bounce || key >= (uint32_t)(nbins - 1)) return false;
    return !(mats[key * ptd::MAT_WORDS + 9] > 0.0f);
}

// in-place exclusive scan of `total` words by one workgroup (1024 words per step); returns the sum
__device__ __forceinline__ uint32_t scan_words_inplace(uint32_t *w, uint32_t total, uint32_t *lds_scan) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (total <= BLOCK * 128u) {
        // small tables (the usual case: 8 keys x 2048 workgroups = 16 K words): every thread owns one contiguous
        // segment, sums it with all its 16-B loads in flight, the 256 sums cross through one wave scan + one LDS
        // exchange, and the segment is read again (L2) and written as prefixes -- one barrier instead of one per
        // 1024 words with a carried dependency (16 steps of ~1.5 us: half of k_sort_hist's 50 us)
        const uint32_t per4 = ((total + BLOCK - 1) / BLOCK + 3) / 4;       // uint4s per thread, <= 32
        const uint32_t first = threadIdx.x * per4 * 4;
        const uint4 *w4 = reinterpret_cast<const uint4 *>(w);
        uint32_t sum = 0;
        for (uint32_t k = 0; k < per4; ++k) {
            const uint32_t e = first + 4 * k;
            if (e < total) {
                const uint4 v = w4[e >> 2];                                  // the table is padded to a multiple of 4 words
                sum += v.x + (e + 1 < total ? v.y : 0u) + (e + 2 < total ? v.z : 0u) + (e + 3 < total ? v.w : 0u);
            }
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        if (lane == 63) lds_scan[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) {
            const uint32_t c = lds_scan[k];
            if (k < wave) wave_off += c;
            tot += c;
        }
        uint32_t run = wave_off + incl - sum;
        for (uint32_t k = 0; k < per4; ++k) {
            const uint32_t e = first + 4 * k;
            if (e < total) {
                const uint4 v = w4[e >> 2];
                uint4 o;
                o.x = run; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
                run = o.w + v.w;
                if (e + 3 < total) reinterpret_cast<uint4 *>(w)[e >> 2] = o;
                else { w[e] = o.x; if (e + 1 < total) w[e + 1] = o.y; if (e + 2 < total) w[e + 2] = o.z; }
            }
        }
        return tot;
    }
    const uint32_t steps = (total + 4 * BLOCK - 1) / (4 * BLOCK);
    uint32_t carry = 0;
    for (uint32_t step = 0; step < steps; ++step) {
        const uint32_t e = (step * BLOCK + threadIdx.x) * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (e + k < total) ? w[e + k] : 0u;
        const uint32_t sum = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        uint32_t *slot = lds_scan + (step & 1) * WAVES;
        if (lane == 63) slot[wave] = incl;
        __syncthreads();
        uint32_t wave_off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) {
            const uint32_t c = slot[k];
            if (k < wave) wave_off += c;
            tot += c;
        }
        uint32_t run = carry + wave_off + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e + k < total) w[e + k] = run;
            run += v[k];
        }
        carry += tot;
    }
    return carry;
}

// chunks of the pool a workgroup owns in the sort kernels: [first, first + count)
__device__ __forceinline__ void sort_run(uint32_t n, uint32_t &first, uint32_t &count) {
    const uint32_t chunks = (n + SORT_CHUNK - 1) / SORT_CHUNK;
    const uint32_t per = (chunks + gridDim.x - 1) / gridDim.x;
    first = min(chunks, blockIdx.x * per);
    count = min(chunks - first, per);
}

// one round per distinct key among the valid lanes: f(key, ballot of the lanes holding it)
template <typename F>
__device__ __forceinline__ void for_each_key(bool valid, uint32_t key, F f) {
    uint64_t rem = ballot64(valid);
    while (rem) {
        const int l = __ffsll((unsigned long long)rem) - 1;
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, l);
        const uint64_t m = ballot64(valid && key == k);
        f(k, m);
        rem &= ~m;
    }
}

template <bool COMPACT>
__global__ __launch_bounds__(BLOCK) void k_sort_hist(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *bins = sctl + LDS_CTL_WORDS;                            // the workgroup's bins
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr bool compact = COMPACT;
    const uint32_t n = (compact && a.depth > 0) ? a.ctl->nlive[a.depth] : a.pool_n;
    uint32_t first, count;
    sort_run(n, first, count);
    for (int b = threadIdx.x; b < a.nbins; b += BLOCK) bins[b] = 0;
    __syncthreads();
    for (uint32_t c = 0; c < count; ++c) {
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t i = ((first + c) * SORT_CHUNK_TILES + wave * SORT_TPW + s) * TILE + lane;
            const bool valid = i < n;
            const uint32_t key = valid ? sort_key(a.isect, i, a.nbins) : 0u;
            for_each_key(valid, key, [&](uint32_t k, uint64_t m) {
                if (lane == 0) atomicAdd(&bins[k], (uint32_t)__popcll((unsigned long long)m));
            });
        }
    }
    __syncthreads();
    // publish table[bin][workgroup] (write-through), then elect the last workgroup to scan it
    const bool last_bounce = a.depth == a.trace_depth - 1;
    for (int b = threadIdx.x; b < a.nbins; b += BLOCK) {
        const uint32_t cnt = (!compact || key_survives(a.scene.mats, (uint32_t)b, a.nbins, last_bounce)) ? bins[b] : 0u;
        __hip_atomic_store(&a.sort_table[(size_t)b * gridDim.x + blockIdx.x], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = elect_last(a.ctl->bucket[a.depth][1], &a.ctl->done_sort[a.depth]);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        sctl[0] = last ? 1u : 0u;
    }
    __syncthreads();
    if (sctl[0]) {
        const uint32_t total = scan_words_inplace(a.sort_table, (uint32_t)a.nbins * gridDim.x, sctl + 2);
        if (threadIdx.x == 0) {
            if (compact) a.ctl->nlive[a.depth + 1] = total;
            if (a.depth == 0) a.ctl->nlive[0] = a.pool_n;
        }
    }
}

template <bool COMPACT>
__global__ __launch_bounds__(BLOCK) void k_shade_sorted(BounceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    const int nb = (a.nbins + 3) & ~3;
    uint32_t *gbase = sctl + LDS_CTL_WORDS;          // [nb] output position of this workgroup's next path per key
    uint32_t *ktot = gbase + nb;                     // [nb] paths per key in the chunk
    uint32_t *kstart = ktot + nb;                    // [nb] first sorted position of the key in the chunk
    uint32_t *wcount = kstart + nb;                  // [WAVES][nb] per-wave counts, then running sorted positions
    uint32_t *order = wcount + WAVES * nb;           // [SORT_CHUNK] sorted position -> element of the chunk
    uint32_t *keyl = order + SORT_CHUNK;             // [SORT_CHUNK] element -> key
    float *mats = reinterpret_cast<float *>(keyl + SORT_CHUNK);       // materials (when they fit: a.nbins <= 64)
    const bool mats_lds = a.nbins <= 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    const uint32_t stamp = batch_stamp(a.fin_stamp, a.ctl);
    const uint32_t n = (COMPACT && a.depth > 0) ? a.ctl->nlive[a.depth] : a.pool_n;
    const bool last_bounce = a.depth == a.trace_depth - 1;
    uint32_t first, count;
    sort_run(n, first, count);
    for (int b = threadIdx.x; b < a.nbins; b += BLOCK) gbase[b] = a.sort_table[(size_t)b * gridDim.x + blockIdx.x];
    if (mats_lds)
        for (int k = threadIdx.x; k < a.scene.nmats * ptd::MAT_WORDS; k += BLOCK) mats[k] = a.scene.mats[k];
    const float *mat_src = mats_lds ? mats : a.scene.mats;
    uint32_t traced = 0;
    for (uint32_t c = 0; c < count; ++c) {
        const uint32_t chunk_base = (first + c) * SORT_CHUNK;
        // ---- A: keys of the wave's two tiles, per-wave counts ----
        for (int b = lane; b < a.nbins; b += 64) wcount[wave * nb + b] = 0;
        uint32_t key2[SORT_TPW];
        bool valid2[SORT_TPW];
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t e = (uint32_t)(wave * SORT_TPW + s) * TILE + lane;
            const uint32_t i = chunk_base + e;
            valid2[s] = i < n;
            key2[s] = valid2[s] ? sort_key(a.isect, i, a.nbins) : 0u;
            keyl[e] = key2[s];
            for_each_key(valid2[s], key2[s], [&](uint32_t k, uint64_t m) {
                if (lane == 0) wcount[wave * nb + k] += (uint32_t)__popcll((unsigned long long)m);
            });
        }
        __syncthreads();
        // ---- B: per key, counts -> starts of each wave's share; chunk totals; starts of the keys ----
        for (int b = threadIdx.x; b < a.nbins; b += BLOCK) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { const uint32_t v = wcount[w * nb + b]; wcount[w * nb + b] = run; run += v; }
            ktot[b] = run;
        }
        __syncthreads();
        if (wave == 0) {
            uint32_t carry = 0;
            for (int base = 0; base < a.nbins; base += 64) {
                const uint32_t v = (base + lane < a.nbins) ? ktot[base + lane] : 0u;
                uint32_t incl = v;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t u = __shfl_up(incl, off);
                    if (lane >= off) incl += u;
                }
                if (base + lane < a.nbins) kstart[base + lane] = carry + incl - v;
                carry += (uint32_t)__shfl((int)incl, 63);
            }
        }
        __syncthreads();
        // ---- C: sorted position of every element (stable: tiles in order, lanes in order) ----
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t e = (uint32_t)(wave * SORT_TPW + s) * TILE + lane;
            for_each_key(valid2[s], key2[s], [&](uint32_t k, uint64_t m) {
                const uint32_t base = kstart[k] + wcount[wave * nb + k];           // same address for the whole wave
                if (valid2[s] && key2[s] == k) order[base + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1)))] = e;
                if (lane == 0) wcount[wave * nb + k] += (uint32_t)__popcll((unsigned long long)m);
            });
        }
        __syncthreads();
        // ---- D: shade 128 consecutive sorted positions per wave ----
        const uint32_t chunk_n = min((uint32_t)SORT_CHUNK, n - chunk_base);
#pragma unroll                       // both tiles' gathers in flight together: +5 % (profiles/r02/variants_sort.log)
        for (int s = 0; s < SORT_TPW; ++s) {
            const uint32_t p = (uint32_t)(wave * SORT_TPW + s) * TILE + lane;
            bool active = p < chunk_n;
            uint32_t key = 0, i = 0, pid = DEAD_PID, dst = 0;
            f3 ro = ptd::mk(0, 0, 0), rd = ptd::mk(0, 0, 1), col = ptd::mk(1, 1, 1);
            if (active) {
                const uint32_t e = order[p];
                key = keyl[e];
                i = chunk_base + e;
                dst = gbase[key] + (p - kstart[key]);
                const SlotPtr q = a.in.slot(i);
                pid = ppid(q);
                ro = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
                rd = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
                col = ptd::mk(pf(q, 6), pf(q, 7), pf(q, 8));
            }
            const bool have = active;
            if (pid == DEAD_PID) active = false;
            bool alive = false;
            ptd::PathState ps;
            ps.o = ro; ps.d = rd; ps.c = col;
            if (active) {
                const float t = at(a.isect.plane(0), i);
                const f3 nrm = ptd::mk(at(a.isect.plane(1), i), at(a.isect.plane(2), i), at(a.isect.plane(3), i));
                const int m = at(a.isect.mat(), i);
                const uint32_t smp = sample_of(a.map, pid);
                const int pixel = local_to_pixel(a.map, (int)(pid - smp * (uint32_t)a.map.tile_pixels));
                alive = ptd::shade_scatter(ps, t, nrm, m & 0x7fffffff, (m < 0) ? 0 : 1, mat_src, iter0 + (int)smp, pixel,
                                           a.depth, last_bounce);
                if (!alive) {
                    put_final(a.fin, pid, ps.c, stamp);
                }
            }
            traced += (uint32_t)__popcll((unsigned long long)ballot64(active));
            if (alive) {
                const SlotPtr q = a.out.slot(dst);
                pf(q, 0) = ps.o.x; pf(q, 1) = ps.o.y; pf(q, 2) = ps.o.z;
                pf(q, 3) = ps.d.x; pf(q, 4) = ps.d.y; pf(q, 5) = ps.d.z;
                pf(q, 6) = ps.c.x; pf(q, 7) = ps.c.y; pf(q, 8) = ps.c.z;
                ppid(q) = pid;
            } else if (!COMPACT && have) {
                a.out.pid(dst) = DEAD_PID;
            }
        }
        __syncthreads();
        // ---- E: this workgroup's output positions move on ----
        for (int b = threadIdx.x; b < a.nbins; b += BLOCK)
            if (!COMPACT || key_survives(mat_src, (uint32_t)b, a.nbins, last_bounce)) gbase[b] += ktot[b];
        __syncthreads();
    }
    if (COMPACT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->alive[a.depth] = n;
    } else {
        if (lane == 0) sctl[8 + wave] = traced;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tb = sctl[8] + sctl[9] + sctl[10] + sctl[11];
            if (tb) atomicAdd(&a.ctl->alive[a.depth], tb);
        }
    }
}

// k_shade_sorted_w: the same result with WAVE-PRIVATE sorting, for up to 64 keys (lane k of a wave holds key k's
// counters in registers).  k_shade_sorted above spends its time between six workgroup barriers per 512-path chunk
// (per-wave counts -> per-key prefix -> key starts -> positions -> shade -> advance), each phase waiting for the
// slowest wave's memory latency: 34 us per chunk and workgroup on C3, of which ~2 us are instructions.  Here a wave
// sorts and shades ITS OWN 128 paths of the chunk (two tiles: stable counting sort through a 128-word LDS strip that
// only this wave touches, so LDS program order replaces the barriers), and the four waves of the workgroup meet once
// per chunk, to exchange their per-key counts: the output position of wave w's first key-k path is
//     gbase[k] + sum over w' < w of count_w'[k],
// the order of the workgroup-wide sort (chunks in order, elements in order), so the global result -- pool order after
// the bounce = stable partition of the stable sort by key -- is unchanged and k_sort_hist's per-workgroup table too.
// The exchange slots alternate by chunk parity: a wave that has passed barrier c cannot still be reading the slots of
// chunk c - 1, so one barrier per chunk is enough.  The gathers of a wave touch only its own two tiles' rows (at most
// four 128-B lines per instruction), so nothing is staged.
constexpr int SORTW_MAX_BINS = 64;
__host__ __device__ constexpr size_t shade_sorted_w_lds_words(int nmats) {
    return (size_t)LDS_CTL_WORDS + 2 * WAVES * 64 + (size_t)((nmats * ptd::MAT_WORDS + 3) & ~3);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off);
        if (lane >= off) v += u;
    }
    return v;
}

template <bool COMPACT, bool GEN = false>
__global__ __launch_bounds__(BLOCK, GEN ? 6 : 8) void k_shade_sorted_w(BounceArgs a) {
    static_assert(SORT_TPW == 2, "a wave handles two tiles per chunk");
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    uint32_t *sctl = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *xch = sctl + LDS_CTL_WORDS;                    // [2][WAVES][64]: per-key counts of each wave, by chunk parity
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mats = reinterpret_cast<float *>(xch + 2 * WAVES * 64);
    const int iter0 = a.iter0 >= 0 ? a.iter0 : (int)a.ctl->iter0;
    const uint32_t stamp = batch_stamp(a.fin_stamp, a.ctl);
    const uint32_t n = (COMPACT && a.depth > 0) ? a.ctl->nlive[a.depth] : a.pool_n;
    const bool last_bounce = a.depth == a.trace_depth - 1;
    uint32_t first, count;
    sort_run(n, first, count);
    // lane k: where this workgroup's next path with key k goes, and whether paths with key k go on at all
    uint32_t gbase = lane < a.nbins ? a.sort_table[(size_t)lane * gridDim.x + blockIdx.x] : 0u;
    const bool key_lives = lane < a.nbins && (!COMPACT || key_survives(a.scene.mats, (uint32_t)lane, a.nbins, last_bounce));
    for (int k = threadIdx.x; k < a.scene.nmats * ptd::MAT_WORDS; k += BLOCK) mats[k] = a.scene.mats[k];
    __syncthreads();
    const uint64_t lt = (1ull << lane) - 1;
    uint32_t traced = 0;
    for (uint32_t c = 0; c < count; ++c) {
        const uint32_t sub_base = (first + c) * SORT_CHUNK + (uint32_t)wave * (SORT_TPW * TILE);
        // ---- the wave's two tiles, whole rows: state + intersection (every load coalesced, all in flight together) ----
        uint32_t idx[SORT_TPW], pid[SORT_TPW], key[SORT_TPW];
        bool valid[SORT_TPW];
        f3 ro[SORT_TPW], rd[SORT_TPW], col[SORT_TPW], nrm[SORT_TPW];
        float th[SORT_TPW];
        int mh[SORT_TPW];
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            idx[s] = sub_base + (uint32_t)s * TILE + lane;
            valid[s] = idx[s] < n;
            pid[s] = DEAD_PID; th[s] = -1.0f; mh[s] = 0;
            ro[s] = ptd::mk(0, 0, 0); rd[s] = ptd::mk(0, 0, 1); col[s] = ptd::mk(1, 1, 1); nrm[s] = ptd::mk(0, 0, 0);
            if (valid[s]) {
                th[s] = at(a.isect.plane(0), idx[s]);
                mh[s] = at(a.isect.mat(), idx[s]);
                nrm[s] = ptd::mk(at(a.isect.plane(1), idx[s]), at(a.isect.plane(2), idx[s]), at(a.isect.plane(3), idx[s]));
                if (GEN) {                                         // bounce 0 of a batch: the ray k_intersect<GEN> generated
                    pid[s] = idx[s];
                    const uint32_t smp = sample_of(a.map, pid[s]);
                    const int pixel = local_to_pixel(a.map, (int)(pid[s] - smp * (uint32_t)a.map.tile_pixels));
                    camera_ray(a.cam, a.lens, a.trace_depth, iter0 + (int)smp, pixel, a.map.W, ro[s], rd[s]);
                } else {
                    const SlotPtr q = a.in.slot(idx[s]);
                    pid[s] = ppid(q);
                    ro[s] = ptd::mk(pf(q, 0), pf(q, 1), pf(q, 2));
                    rd[s] = ptd::mk(pf(q, 3), pf(q, 4), pf(q, 5));
                    col[s] = ptd::mk(pf(q, 6), pf(q, 7), pf(q, 8));
                }
            }
            key[s] = valid[s] ? (th[s] > 0.0f ? (uint32_t)(mh[s] & 0x7fffffff) : (uint32_t)(a.nbins - 1)) : 0u;
        }
        // ---- lane k counts key k over the wave's two tiles ----
        uint32_t cnt0 = 0, cnt1 = 0;
        for_each_key(valid[0], key[0], [&](uint32_t k, uint64_t m) { if ((uint32_t)lane == k) cnt0 = (uint32_t)__popcll((unsigned long long)m); });
        for_each_key(valid[1], key[1], [&](uint32_t k, uint64_t m) { if ((uint32_t)lane == k) cnt1 = (uint32_t)__popcll((unsigned long long)m); });
        uint32_t *slot = xch + (c & 1u) * (WAVES * 64);
        slot[wave * 64 + lane] = cnt0 + cnt1;
        __syncthreads();                                                   // the only barrier of the chunk: the waves' counts
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t v = slot[w * 64 + lane];
            if (w < wave) before += v;
            all += v;
        }
        const uint32_t g0 = gbase + before;                                // lane k: output slot of the wave's first key-k path
        const uint32_t g1 = g0 + cnt0;                                     //         ... of tile 1's first key-k path
        if (key_lives) gbase += all;
        // ---- output slots: stable within a key (tile 0's paths, then tile 1's, lanes in order) ----
        uint32_t dst[SORT_TPW] = {0u, 0u};
        for_each_key(valid[0], key[0], [&](uint32_t k, uint64_t m) {
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)g0, (int)k);
            if (valid[0] && key[0] == k) dst[0] = base + (uint32_t)__popcll((unsigned long long)(m & lt));
        });
        for_each_key(valid[1], key[1], [&](uint32_t k, uint64_t m) {
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)g1, (int)k);
            if (valid[1] && key[1] == k) dst[1] = base + (uint32_t)__popcll((unsigned long long)(m & lt));
        });
        // ---- shade in place (the order of shading is not observable; the output order is) ----
#pragma unroll
        for (int s = 0; s < SORT_TPW; ++s) {
            const bool have = valid[s];
            const bool active = have && pid[s] != DEAD_PID;
            bool alive = false;
            ptd::PathState ps;
            ps.o = ro[s]; ps.d = rd[s]; ps.c = col[s];
            if (active) {
                const uint32_t smp = sample_of(a.map, pid[s]);
                const int pixel = local_to_pixel(a.map, (int)(pid[s] - smp * (uint32_t)a.map.tile_pixels));
                alive = ptd::shade_scatter(ps, th[s], nrm[s], mh[s] & 0x7fffffff, (mh[s] < 0) ? 0 : 1, mats, iter0 + (int)smp, pixel,
                                           a.depth, last_bounce);
                if (!alive) {
                    put_final(a.fin, pid[s], ps.c, stamp);
                }
            }
            traced += (uint32_t)__popcll((unsigned long long)ballot64(active));
            if (alive) {
                const SlotPtr q = a.out.slot(dst[s]);
                pf(q, 0) = ps.o.x; pf(q, 1) = ps.o.y; pf(q, 2) = ps.o.z;
                pf(q, 3) = ps.d.x; pf(q, 4) = ps.d.y; pf(q, 5) = ps.d.z;
                pf(q, 6) = ps.c.x; pf(q, 7) = ps.c.y; pf(q, 8) = ps.c.z;
                ppid(q) = pid[s];
            } else if (!COMPACT && have) {
                a.out.pid(dst[s]) = DEAD_PID;
            }
        }
    }
    if (COMPACT) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl->alive[a.depth] = n;
    } else {
        if (lane == 0) sctl[8 + wave] = traced;
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t tb = sctl[8] + sctl[9] + sctl[10] + sctl[11];
            if (tb) atomicAdd(&a.ctl->alive[a.depth], tb);
        }
    }
}

}  // namespace
