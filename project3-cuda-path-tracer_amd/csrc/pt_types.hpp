// pt_types.hpp -- device-side data structures of libptmi355.so (included by ptmi355.hip only):
// build switches, the path pool / intersection planes / range directory / control block layouts
// and the kernel parameter blocks.  See DESIGN.md sections 2 and 6.
#pragma once

namespace {


#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4                     // waves per SIMD the bounce kernels are register-budgeted for
#endif
#ifndef PT_FUSED_WAVES
#define PT_FUSED_WAVES 6                     // the plain fused k_bounce (no mesh, no material keys): 79 VGPRs when left alone in round 3 -- six
                                             // workgroups per CU -- and 82 after the scatter block changed in round 4; capped at 80 the register
                                             // allocator finds the 79 again without a spill
#endif
#ifndef PT_ISECT_WAVES
#define PT_ISECT_WAVES 4                     // k_intersect (unfused / sorted pipelines)
#endif
#ifndef PT_PRE_WAVES
#define PT_PRE_WAVES 6                       // k_bounce<MESH_PRE>: 87-97 VGPRs unconstrained; capped at 80 (six workgroups per CU, 3-5 registers
                                             // spilled): C4 + hierarchy 17.5 -> 18.05 Grays/s against the cap of 96 (profiles/r03/variants_pre_waves.log)
#endif
#ifndef PT_ITER_WAVES
#define PT_ITER_WAVES PT_MIN_WAVES            // k_iteration: 92 VGPRs unconstrained (five workgroups per CU)
#endif
#ifndef PT_SORT_WAVES
#define PT_SORT_WAVES 6                      // k_bounce with material keys: 84-87 VGPRs unconstrained (five workgroups per CU); capped at 80 (six, 2-9
                                             // registers spilled) C3 sorted 37.6 -> 38.4 Grays/s within one box (profiles/r04/ab_sincos_iter_cap.log:
                                             // `all6`; round 3 had measured the cap neutral)
#endif
#ifndef PT_LOOP_WAVES
#define PT_LOOP_WAVES 3                      // k_bounce<MESH_TILES>: round 6 (stage 1 on the matrix pipe: four B operands, four A operands in flight,
                                             // sixteen accumulators): 170.2 Mrays/s on C4 at 3 waves per SIMD (168 VGPRs, no spill), 165.7 at 4 (35-47
                                             // registers spilled), 158.9 at 2 (profiles/r06/ab_c4_loop_waves.txt); round 3's vector form: 41.9 at 4
#endif
constexpr int BLOCK = 256;                 // 4 waves of 64
constexpr int WAVES = BLOCK / 64;
constexpr int TILE = 64;                   // paths per tile = one wave64
constexpr int TRI_WORDS = 12;              // v0 e1 e2 + 3 pad: three 16-B words per triangle
constexpr int BVH_NODE_WORDS = 16;         // one 64-B record per internal node, pt_bvh.hpp
enum { MESH_NONE = 0, MESH_TILES = 1, MESH_BVH = 2, MESH_PRE = 3 };   // how triangle meshes are intersected (template
                                           // switch): every triangle through LDS tiles, hierarchy walked inline, or results
                                           // of the lane-dense mesh pre-pass (k_mesh) read back
constexpr int MAX_DEPTH = 64;
constexpr uint32_t DEAD_PID = 0xffffffffu;

// ---------------------------------------------------------------------------
// device-side parameter blocks (few pointers: every extra pointer pair costs
// 2 SGPRs per wave for the whole kernel)
// ---------------------------------------------------------------------------
// element `i` of a wave-uniform plane pointer through a 32-bit byte offset: the address is
// SGPR base + zero-extended VGPR offset (one global_load/store, no 64-bit VALU address math).
// pt_init guarantees cap * 4 < 2^32.
template <typename T>
__device__ __forceinline__ T &at(T *plane, uint32_t i) {
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(plane) + (i << 2));
}

// Path pool: SoA *per 64-path tile* -- tile T holds the ten fields (ox oy oz dx dy dz cr cg cb pid) of its 64 paths in
// 2560 consecutive bytes.  A wave reads / writes whole rows (coalesced), and every field of slot s sits at a per-lane
// address plus an immediate: no plane base pointers in SGPRs.  Rounds 1-3: ten 256-B dword rows; now three rows of field
// quads / a pair (below).
#ifndef PT_POOL_QUADS
#define PT_POOL_QUADS 1
#endif
#if PT_POOL_QUADS
// Round 4: the ten fields of a tile sit as THREE rows -- [ox oy oz dx] and [dy dz cr cg], 16 B per lane (1024-B rows), and
// [cb pid], 8 B per lane (512 B): a wave moves a tile with three loads or stores instead of ten dword ones -- same
// bytes, same 2560-B tile, a third of the memory instructions, and where a row store is several interleaved runs
// (survivors placed by material) the runs are four times as long.  (Field pairs -- five 8-B rows -- were the first step:
// C2 +3.0 %, C3 sorted +6 %, profiles/r04/ab_pool_pairs.log.)
struct SlotPtr { char *q, *c; };        // the lane's place in the two 16-B rows / in the 8-B row of its tile
struct Pool {
    float *base;
    uint32_t cap;        // slots, a multiple of 64
    __device__ __forceinline__ SlotPtr slot(uint32_t s) const {
        char *t = reinterpret_cast<char *>(base) + (size_t)(s >> 6) * 2560u;
        return SlotPtr{t + ((s & 63u) << 4), t + 2048u + ((s & 63u) << 3)};
    }
    __device__ __forceinline__ float &f(uint32_t s, int k) const {
        const SlotPtr p = slot(s);
        return *reinterpret_cast<float *>(k < 8 ? p.q + (k >> 2) * 1024 + (k & 3) * 4 : p.c);
    }
    __device__ __forceinline__ uint32_t &pid(uint32_t s) const { return *reinterpret_cast<uint32_t *>(slot(s).c + 4); }
};
__device__ __forceinline__ float &pf(const SlotPtr &p, int k) { return *reinterpret_cast<float *>(k < 8 ? p.q + (k >> 2) * 1024 + (k & 3) * 4 : p.c); }
__device__ __forceinline__ uint32_t &ppid(const SlotPtr &p) { return *reinterpret_cast<uint32_t *>(p.c + 4); }
#else
typedef char *SlotPtr;
struct Pool {
    float *base;
    uint32_t cap;        // slots, a multiple of 64
    // the ten fields of a tile as FIVE 512-B rows of field PAIRS (ox oy | oz dx | dy dz | cr cg | cb pid)
    __device__ __forceinline__ char *slot(uint32_t s) const {
        return reinterpret_cast<char *>(base) + (size_t)(s >> 6) * 2560u + ((s & 63u) << 3);
    }
    __device__ __forceinline__ float &f(uint32_t s, int k) const { return *reinterpret_cast<float *>(slot(s) + (k >> 1) * 512 + (k & 1) * 4); }
    __device__ __forceinline__ uint32_t &pid(uint32_t s) const { return *reinterpret_cast<uint32_t *>(slot(s) + 4 * 512 + 4); }
};
__device__ __forceinline__ float &pf(char *slot, int k) { return *reinterpret_cast<float *>(slot + (k >> 1) * 512 + (k & 1) * 4); }
__device__ __forceinline__ uint32_t &ppid(char *slot) { return *reinterpret_cast<uint32_t *>(slot + 4 * 512 + 4); }
#endif

struct Isect {           // ShadeableIntersection planes t nx ny nz mat (unfused / sort / fake-shader modes)
    float *base;         // mat: bit 31 carries the winning test's !outside
    uint32_t cap;
    __device__ __forceinline__ float *plane(int k) const { return base + (size_t)k * cap; }
    __device__ __forceinline__ int *mat() const { return reinterpret_cast<int *>(base + (size_t)4 * cap); }
};

struct TileMap {         // local pixel index -> global pixelIndex (x + y*W)
    int W, H;
    int tile_index, tile_count, strip_rows;
    int tile_pixels;     // pixels owned by this tile
    uint32_t div_magic;  // pid / tile_pixels without an integer divide (see sample_of)
    uint32_t div_shift;
};

// pid / tile_pixels for every 32-bit pid: round-up magic number, branch-free form
// (q = mulhi(magic, n); ((n - q) >> 1) + q) >> shift), magic/shift chosen by make_div_magic().
__device__ __forceinline__ uint32_t sample_of(const TileMap &m, uint32_t pid) {
    if (m.tile_pixels == 1) return pid;                  // the branch-free form needs a divisor >= 2
    const uint32_t q = __umulhi(m.div_magic, pid);
    return (((pid - q) >> 1) + q) >> m.div_shift;
}

struct Control {         // zeroed from `stamp` on by one hipMemsetAsync per batch (2 KiB)
    uint32_t iter0;                 // first iteration of the batch when the launches come from a replayed graph
    uint32_t keep[15];              //   (kernel arguments are frozen at capture time); survives the per-batch clear.
                                    //   keep[0]: the batch's final-colour stamp under graph replay (-DPT_MESH_STATS builds count
                                    //   walks there: do not combine them with PTMI355_GRAPH=1)
    unsigned long long stamp[16];   // -DPT_STAMPS: s_memrealtime at the phases of wave 0 / the last workgroup
    uint32_t nlive[MAX_DEPTH + 1];  // nlive[d] = paths entering bounce d (compaction on)
    uint32_t alive[MAX_DEPTH + 1];  // paths actually traced at bounce d
    uint32_t done[MAX_DEPTH];       // election buckets that finished bounce d (last-one-out election, top level)
    uint32_t done_sort[MAX_DEPTH];  // same for the material-sort histogram of bounce d
    uint32_t error;
    uint32_t scan_ticks[MAX_DEPTH]; // 100 MHz ticks the last workgroup spent scanning (diagnostic)
    uint32_t pad[512 - 16 - 32 - 2 * (MAX_DEPTH + 1) - 3 * MAX_DEPTH - 1];
    // first-level election counters: 32 buckets per bounce, one 64-B line apart
    uint32_t bucket[MAX_DEPTH][2][32 * 16];       // [bounce][bounce kernel | sort histogram][bucket * 16]
    // NOT part of the per-batch clear: k_iteration's own last-workgroup election (elect_last_self_clearing) -- 32 buckets
    // one 64-B line apart, then the top counter; the last arriver of each puts it back to zero, so that a batch that runs
    // as one launch needs no hipMemsetAsync in front of it
    uint32_t ticket[33 * 16];
};
constexpr int ELECT_BUCKETS = 32;
static_assert(offsetof(Control, ticket) == 2048 + 2 * MAX_DEPTH * 32 * 16 * 4, "Control up to `ticket` is one memset block");
static_assert(sizeof(Control) % 16 == 0, "memset block is a multiple of 16 B");

// Last-workgroup-out election without hammering one address: a same-address atomic costs ~12 ns at
// the memory side, so 1-2 thousand workgroups finishing together would serialise for tens of
// microseconds.  Workgroup b adds to bucket b % 32 (own cache line); the last arriver of a bucket
// adds to the top counter; the last of those is the last workgroup of the launch.  Call from ONE
// thread, after the workgroup's stores have drained and its barrier.
__device__ __forceinline__ bool elect_last(uint32_t *buckets /* [32*16] */, uint32_t *top) {
    const uint32_t G = gridDim.x;
    const uint32_t k = blockIdx.x % ELECT_BUCKETS;
    const uint32_t members = (G - k + ELECT_BUCKETS - 1) / ELECT_BUCKETS;       // workgroups with b % 32 == k
    const uint32_t used = G < ELECT_BUCKETS ? G : ELECT_BUCKETS;                 // buckets that have members
    const uint32_t old = __hip_atomic_fetch_add(&buckets[k * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old != members - 1) return false;
    const uint32_t t = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t == used - 1;
}

// The same election on counters nobody clears (Control::ticket): the last arriver of a bucket puts it back to zero, the
// last of the launch the top counter too.  Valid because the launches that share a control block are stream-ordered:
// the next launch's first workgroup arrives after this launch has ended.
__device__ __forceinline__ bool elect_last_self_clearing(uint32_t *ticket /* [33*16] */) {
    const uint32_t G = gridDim.x;
    const uint32_t k = blockIdx.x % ELECT_BUCKETS;
    const uint32_t members = (G - k + ELECT_BUCKETS - 1) / ELECT_BUCKETS;
    const uint32_t used = G < ELECT_BUCKETS ? G : ELECT_BUCKETS;
    const uint32_t old = __hip_atomic_fetch_add(&ticket[k * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old != members - 1) return false;
    __hip_atomic_store(&ticket[k * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t t = __hip_atomic_fetch_add(&ticket[32 * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != used - 1) return false;
    __hip_atomic_store(&ticket[32 * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Range directory of one bounce's OUTPUT pool.  Wave w of the persistent grid owns the
// contiguous run of `R` logical tiles [wR, (w+1)R) (R = ceil(tiles / W)) and packs every
// survivor of that run, in order, to the front of the run's own span of R*64 slots; count[w]
// is how many it packed, base[] the exclusive scan of count[] (W+1 entries).  Logical path i
// of the output therefore lives in slot r*R*64 + (i - base[r]) for the range r with
// base[r] <= i < base[r+1] -- the stable partition's order, with no cross-wave communication
// inside the launch and only W (<= 8192) words to scan at its end.
//
// Material sort folded into the compaction (PT_SORT_MATERIAL, fused form): the pool order the completion spec asks for
// after a sorted bounce is "survivors, stably sorted by the material they hit".  With K materials the directory
// simply has K * W ranges, KEY-MAJOR: range k * W + w holds wave w's survivors whose hit had material k, packed at the
// front of a span of their own (same stride R*64: the pool is K times as large -- 288 GB of HBM are there to be
// used), and the exclusive scan of count[] in that order is the sorted order.  Readers are unchanged: logical path i
// lives in slot r*R*64 + (i - base[r]).  Nothing is moved to be sorted, and nothing extra is read or written.
//
// Round 4 -- tiles ALIGNED to the ranges.  Cutting the logical sequence into tiles of 64 wherever they fall makes almost
// every tile straddle two physical tiles of the source pool (logical i sits at offset i - base[r] in its range): each of
// its ten rows is then 64-B sectors of two 256-B rows, the sectors at the seam are fetched twice -- by this tile and by
// its neighbour -- and they have often left the L2 in between (measured: 1.34 x the necessary reads; the microbenchmark
// profiles/microbench/hbm_patterns.hip reproduces 1.24 x).  So the readers of an unsorted packed pool walk it in tiles
// that start at the range starts: range r holds ceil(count[r] / 64) tiles, tbase[] is their exclusive scan (made by the
// same last-workgroup scan as base[]), tile T lies in the range r with tbase[r] <= T < tbase[r + 1] and reads the slots
// r * span + 64 (T - tbase[r]) + lane: whole rows, except in the last tile of a range.  The order is the logical order
// (ranges in order, tiles in order), so the stable partition is untouched; the price is up to one partly filled tile per
// range (W of ~260 000 tiles per launch on C2).  span (slots per range) is what the PRODUCER used: it is written into the
// directory (it no longer follows from the live count alone).
#ifndef PT_ALIGNED_TILES
#define PT_ALIGNED_TILES 1
#endif
struct RangeDir {
    uint32_t *mem;       // count[nrp] | base[nrp+4] | tbase[nrp+4] | span  (nrp = nr rounded up to 4); nullptr = dense pool
    uint32_t W;          // waves in the persistent grid
    uint32_t nr;         // ranges = W, or K * W when survivors are placed by material
    __device__ __forceinline__ uint32_t *count() const { return mem; }
    __device__ __forceinline__ uint32_t *base() const { return mem + ((nr + 3u) & ~3u); }
    __device__ __forceinline__ uint32_t *tbase() const { return mem + 2u * ((nr + 3u) & ~3u) + 4u; }   // nr + 1 entries
    __device__ __forceinline__ uint32_t *span() const { return mem + 3u * ((nr + 3u) & ~3u) + 8u; }     // slots per range, written by the producer
};
__host__ __device__ constexpr size_t range_dir_words(size_t nr) { return 3u * ((nr + 3u) & ~(size_t)3u) + 12u; }

// Which run of tiles a wave owns: wave j of workgroup b takes run j*G + b, so the first G runs go
// to G different workgroups.  When a bounce has fewer runs than waves (small pools, late
// bounces) the busy waves are then spread over every CU instead of filling the first workgroups
// the dispatcher happens to co-locate (measured at 800x800, 1 spp, bounce 7: 42 -> 2x shorter).
__device__ __forceinline__ uint32_t run_id() {
    return __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * gridDim.x + blockIdx.x);
}

// tiles per range for a pool of n paths split over W waves
__host__ __device__ __forceinline__ uint32_t range_tiles(uint32_t n, uint32_t W) {
    const uint32_t tiles = (n + 63u) / 64u;
    return (tiles + W - 1) / W;
}

struct Persist {         // survives the per-batch memset
    unsigned long long rays;        // sum over bounces of paths traced since pt_init
    unsigned long long iterations;
    unsigned long long first_rays;  // paths traced at bounce 0 (rays - first_rays = compaction survivors)
};

// What pt_trace / pt_trace_batch hand back as pt_stats, written by the last workgroup of k_iteration straight into
// page-locked host memory (device-mapped): the synchronous call then needs no device-to-host copy of the control block
// after its stream synchronisation -- at one iteration per call that copy was a tenth of the call.
struct HostStats {
    uint32_t alive[MAX_DEPTH + 1];
    uint32_t error;
    uint32_t serial;                // the batch stamp of the launch that wrote this (collect_stats checks it)
};

// camera extensions of the completion spec (the TODO at pathtrace.cu:134)
struct Lens {
    int aa;                  // jitter the sample position inside its pixel
    float radius, focal;     // thin lens; radius <= 0: pinhole
};

struct SceneDev {
    const float *geoms;  int ngeoms;       // GEOM_WORDS dwords each (mesh ranges / grids; scalar loads)
    const float *cull;                     // CULL_WORDS per geom: padded world box + type (scalar loads)
    const float *grec;                     // GREC_WORDS per geom: the three matrices (per-lane gathers)
    const uint32_t *ginfo;                 // per geom: materialid | type << 28
    float rmax;                            // |origin|_1 bound the cull boxes were derived for
    const float *mats;   int nmats;        // MAT_WORDS dwords each
    const float *tris;   int ntris;        // v0, e1, e2 + pad (12 dwords each)
    const void *tri_rec;                   // per triangle the 32 binary16 K-slots (64 B) of the every-triangle loop's first stage
                                           //   (pt_h_scene.hpp: make_tri_records; per mesh padded to a multiple of 64 records, offset in
                                           //   word G_INV + 6 of its geom record, the mesh's frame {g, 1 / Rm} in words G_INV + 7 .. + 10)
    const float *bvh_nodes;                // PT_MESH_BVH: all meshes' trees, BVH_NODE_WORDS per node
    const float *bvh_tris;                 //   leaf-ordered triangle records, word 9 = original index
    float bvh_prune;  int bvh_guard;       //   prune margin; upper bound on nodes visited per walk
    const int4 *bvh_meshes; int bvh_nmesh; //   per mesh, in geom order: {geom, root record, triangles, top offset | top count << 16}
    const float *bvh_top;  int bvh_top_n;  //   the meshes' first records (the tops of their trees) back to back: k_mesh keeps them in LDS
};

struct BounceArgs {
    // FIRST member (kernarg offset 0): where an INSTRUMENTED build of a kernel adds its per-basic-block execution
    // counts when it ends (profiles/tools/isa_count.py rewrites the device assembly; the library's own code never reads
    // this field).  nullptr unless PTMI355_DBG_COUNTS is set.
    unsigned int *dbg_counts;
    Pool in, out;
    Lens lens;
    Isect isect;
    SceneDev scene;
    TileMap map;
    Control *ctl;
    RangeDir dir_in;       // directory of the pool being read (mem == nullptr: dense)
    RangeDir dir_out;      // directory this launch produces
    float *fin;            // final colours, float4[cap] = {r, g, b, batch stamp}, index = pid; written for non-zero colours only
    uint32_t fin_stamp;    // this batch's stamp; 0: read Control::keep[0] (graph replay)
    pt_camera cam;         // used when gen_rays != 0
    int depth, trace_depth, iter0;   // iter0 < 0: read Control::iter0 (graph replay)
    uint32_t pool_n;       // paths in the pool when compaction is off / at bounce 0
    int gen_rays;          // bounce 0 generates the camera ray instead of loading it
    // mesh pre-pass (k_mesh -> k_bounce<MESH_PRE>): nearest mesh hit per source slot {t, geom, triangle, -}
    // and, per logical 64-path tile, the lanes that have one
    float4 *mesh_hit;
    // one flag per POOL SLOT (64 per word): "mesh_hit[slot] holds this path's mesh result (possibly: no hit)".
    // flags_in describes the pool this bounce reads, flags_out the one it writes: k_bounce marks the survivors whose
    // new ray reaches a mesh's root boxes, so that the next bounce's k_mesh only touches those (mesh_scan = 0)
    unsigned long long *mesh_flags_in, *mesh_flags_out;
    int mesh_scan;         // k_mesh must find the candidates itself (bounce 0, or the previous bounce did not mark)
    // bounce 0 of a pinhole camera: one bit per 64 local pixels, set when a camera ray of those pixels can reach the
    // grid of some mesh (pt_init projects the grids onto the image); nullptr: no such knowledge, test every ray
    const unsigned long long *cam_mask;
    // bounce 0 of a pinhole camera without jitter: per 64 local pixels, the primitives (bit g) some camera ray of
    // those pixels is a cull candidate of (k_cull0_mask); nullptr: test every primitive
    const unsigned long long *cull0;
    uint32_t cull0_tiles;  // words in cull0 = tile_pixels / 64
    // k_iteration at 1 spp, launches that do not overlap others: every wave gathers its own pixels into epi_image (the
    // device's running sum: finalGather inside the launch, no k_gather) and, with a host image, writes the new sums to
    // epi_host (the caller's page-locked image, device-mapped); epi_image == nullptr: k_gather does it
    float *epi_image, *epi_host;
    // ... and epi_direct: no final-colour buffer in between -- the path that ends with a colour adds it to its pixel of
    // epi_image itself (one path per pixel in such a launch) and, with a host image, writes that pixel's new sum to
    // epi_host; the pixels of paths that end with colour 0 are not touched: their sums are what the host already has
    int epi_direct;
    // k_iteration: per-workgroup traced counts [bounce][workgroup] (plain stores, nothing to clear), folded by the
    // launch's last workgroup into Control::alive, Persist and (synchronous calls) the host's pt_stats block
    uint32_t *iter_counts;
    Persist *persist;
    HostStats *host_stats;
    // material sort, two-kernel form: table[key][workgroup] of k_sort_hist / k_shade_sorted; keys = materials + 1 (misses)
    uint32_t *sort_table;
    int nbins;
};

// what k_intersect needs to generate bounce 0's camera rays itself (sorted batches: no k_raygen, no pool to read)
struct RayGen {
    pt_camera cam;
    Lens lens;
    TileMap map;
    int trace_depth, iter0;    // iter0 < 0: read Control::iter0 (graph replay)
};

__device__ __forceinline__ int local_to_pixel(const TileMap &m, int j) {
    if (m.tile_count == 1) return j;
    int ly = j / m.W;
    int x = j - ly * m.W;
    int ls = ly / m.strip_rows;
    int y = (ls * m.tile_count + m.tile_index) * m.strip_rows + (ly - ls * m.strip_rows);
    return x + y * m.W;
}

// generateRayFromCamera (pathtrace.cu:122-143) for one pixel of iteration `iter`; with no jitter
// and no lens nothing random is drawn and the ray is the reference's.  `pix` is the global pixel
// index (it keys the random engine together with the depth slot `trace_depth`, which no bounce uses).
__device__ __forceinline__ void camera_ray(const pt_camera &cam, const Lens &lens, int trace_depth, int iter,
                                           int pix, int W, f3 &ro, f3 &rd) {
    const int y = pix / W;
    const int x = pix - y * W;
    f3 view = ptd::mk(cam.view.x, cam.view.y, cam.view.z);
    f3 right = ptd::mk(cam.right.x, cam.right.y, cam.right.z);
    f3 up = ptd::mk(cam.up.x, cam.up.y, cam.up.z);
    f3 pos = ptd::mk(cam.position.x, cam.position.y, cam.position.z);
    float fx = (float)x, fy = (float)y;
    uint32_t rng = 0;
    if (lens.aa || lens.radius > 0.0f) rng = ptd::seeded_engine(iter, pix, trace_depth);
    if (lens.aa) {
        fx = fx + (ptd::u01(rng) - 0.5f);
        fy = fy + (ptd::u01(rng) - 0.5f);
    }
    f3 a = ptd::scale(ptd::scale(right, cam.pixelLength[0]), (fx - (float)cam.resolution[0] * 0.5f));
    f3 b = ptd::scale(ptd::scale(up, cam.pixelLength[1]), (fy - (float)cam.resolution[1] * 0.5f));
    ro = pos;
    rd = ptd::normalize(ptd::sub(ptd::sub(view, a), b));
    if (lens.radius > 0.0f) {
        const float ft = lens.focal / ptd::dot(rd, view);
        const f3 focus = ptd::add(pos, ptd::scale(rd, ft));
        const float r = lens.radius * __builtin_sqrtf(ptd::u01(rng));
        const float theta = ptd::u01(rng) * 6.2831853071795864769252867665590057683943f;
        float sa, ca;
        ptd::sincos_shared(theta, sa, ca);
        ro = ptd::add(ptd::add(pos, ptd::scale(right, r * ca)), ptd::scale(up, r * sa));
        rd = ptd::normalize(ptd::sub(focus, ro));
    }
}

}  // namespace
