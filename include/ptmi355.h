/*
 * ptmi355.h -- C-ABI of libptmi355.so, the MI355X-native (gfx950, HIP) wavefront
 * path tracer that replaces the hot path of CIS565 Project3-CUDA-Path-Tracer.
 *
 * The reference exposes the path as three C++ free functions
 *     void pathtraceInit(Scene *scene);                         src/pathtrace.h:6
 *     void pathtraceFree();                                     src/pathtrace.h:7
 *     void pathtrace(uchar4 *pbo, int frame, int iteration);    src/pathtrace.h:8
 * called only from runCuda() (src/main.cpp:126-127,137,143).  `Scene` holds
 * std::vector / std::string / ifstream (src/scene.h:13-26) and cannot cross a C
 * boundary, so the C-ABI takes the same information as plain pointers + sizes;
 * project3-cuda-path-tracer_amd/host/pathtrace_shim.cpp provides the three
 * reference signatures on top of it.
 *
 * Struct layouts are byte-identical to src/sceneStructs.h:15-76 as compiled for
 * x86-64 (sizes/offsets in SURVEY.md 8b; asserted in csrc/ptmi355.hip).
 *
 * Contract (same as the reference, src/pathtrace.cu:70-75): one renderer
 * instance per process, not re-entrant, calls are synchronous unless stated.
 * One instance may drive several GPUs of the node (pt_scene_desc::devices):
 * that stays invisible to the caller, who still sees one frame.
 * Every int-returning entry point returns PT_OK (0) or a negative pt_status;
 * pt_last_error() then describes the failure.  The library never falls back
 * to a CPU path: without a HIP device every call fails with PT_ERR_DEVICE.
 *
 * Environment.  The shipped library reads exactly these ten variables, at
 * pt_init (PTMI355_RCCL_LIB: when RCCL is first needed).  NONE of them changes a
 * result: images, per-bounce statistics and path order are bit-identical under
 * every setting (each is exercised against the default by a `-m gpu` test, named
 * in brackets); they select launch plans, transports and memory budgets.
 *   PTMI355_DEVICES="0,1,.."|"all"  tile the frame over these GPUs for a host that knows one device
 *                                   (pt_scene_desc::devices below)          [test_devices_from_the_environment_and_the_reference_host]
 *   PTMI355_XCHG=rccl|peer          transport of the in-library tile exchange (default: RCCL when every
 *                                   context has its own device, peer copies otherwise)  [test_rccl_calls_with_a_communicator_of_one]
 *   PTMI355_RCCL_LIB=/path/librccl.so.1  the RCCL library to dlopen before the default names
 *                                                                            [test_rccl_library_named_by_the_environment]
 *   PTMI355_WHOLE_MAX=<paths>       largest batch (paths) traced as ONE launch (k_iteration; default 6 000 000);
 *                                   0 = a kernel per bounce for every batch   [the launch_plan fixture: most parity tests run under both]
 *   PTMI355_WHOLE_MAX_HOST=<paths>  the same limit for a pt_trace call that hands over a host image (default
 *                                   16 000 000)                              [test_4k_one_iteration_per_call_into_the_host_image]
 *   PTMI355_OVERLAP=0|1|n           consecutive asynchronous batches overlap on n lanes (1 = default lane count,
 *                                   0 = strictly one after the other)        [test_overlapped_small_batches]
 *   PTMI355_OVERLAP_GB=<GB>         memory the lanes' extra path pools may take (they are dropped when it does
 *                                   not suffice)                             [test_overlapped_small_batches]
 *   PTMI355_GRAPH=1                 batches captured once and replayed with hipGraphLaunch  [test_graph_replay_equals_direct_launches]
 *   PTMI355_CULL0=0|1               the per-camera bounce-0 candidate masks (k_cull0_mask) off / on  [test_bounce0_candidate_masks]
 *   PTMI355_SCENE_LDS=0             scene records read through the vector cache instead of staged in LDS (what
 *                                   scenes too large for LDS get anyway)     [test_scene_gathers_from_global_memory]
 * Everything else rounds 1-4 switched through the environment (occupancy, stream-layout and transport experiments,
 * test hooks) exists only in a build with -DPT_EXPERIMENTS (profiles/tools/build_variant.sh); pt_version() of such a
 * build ends in "+experiments".
 */
#ifndef PTMI355_H
#define PTMI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- POD mirrors of src/sceneStructs.h ---------------------------------- */
typedef struct pt_vec3 { float x, y, z; } pt_vec3;                 /* glm::vec3, 12 B */
typedef struct pt_mat4 { float m[4][4]; } pt_mat4;                 /* glm::mat4, m[col][row] */

enum pt_geom_type { PT_SPHERE = 0, PT_CUBE = 1,                    /* sceneStructs.h:10-13 */
                    PT_TRIANGLE_MESH = 2 };                        /* extension (INSTRUCTION.md:123-128) */

typedef struct pt_ray { pt_vec3 origin, direction; } pt_ray;       /* sceneStructs.h:15-18, 24 B */

typedef struct pt_geom {                                           /* sceneStructs.h:20-29, 236 B */
    int32_t type;
    int32_t materialid;
    pt_vec3 translation, rotation, scale;
    pt_mat4 transform, inverseTransform, invTranspose;
} pt_geom;

typedef struct pt_material {                                       /* sceneStructs.h:31-41, 44 B */
    pt_vec3 color;
    struct { float exponent; pt_vec3 color; } specular;
    float hasReflective, hasRefractive, indexOfRefraction, emittance;
} pt_material;

typedef struct pt_camera {                                         /* sceneStructs.h:43-52, 84 B */
    int32_t resolution[2];
    pt_vec3 position, lookAt, view, up, right;
    float fov[2];
    float pixelLength[2];
} pt_camera;

typedef struct pt_path_segment {                                   /* sceneStructs.h:62-67, 44 B */
    pt_ray ray;
    pt_vec3 color;
    int32_t pixelIndex;
    int32_t remainingBounces;
} pt_path_segment;

typedef struct pt_shadeable_intersection {                         /* sceneStructs.h:72-76, 20 B */
    float t;
    pt_vec3 surfaceNormal;
    int32_t materialId;
} pt_shadeable_intersection;

/* world-space triangle soup for PT_TRIANGLE_MESH geoms (no reference counterpart) */
typedef struct pt_triangle { pt_vec3 v0, v1, v2; } pt_triangle;    /* 36 B */
typedef struct pt_mesh { int32_t geom_index, first_triangle, triangle_count; } pt_mesh;

/* ---- status codes ---------------------------------------------------------- */
enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,      /* bad argument / called in the wrong state */
    PT_ERR_DEVICE = -2,       /* HIP runtime error (pt_last_error has hipGetErrorString) */
    PT_ERR_NOMEM = -3,
    PT_ERR_INTERNAL = -4      /* kernel-side watchdog tripped (look-back spin bound) */
};

/* ---- run-time toggles (the compile-time #defines the assignment asks for,
 *      INSTRUCTION.md:77-89, as flags) ------------------------------------- */
enum pt_flags {
    PT_COMPACT       = 1u << 0,  /* stable live-path compaction after every bounce */
    PT_SORT_MATERIAL = 1u << 1,  /* live paths stably sorted by the materialId they hit (INSTRUCTION.md:78-86): the pool
                                    order after every bounce is the stable partition of that sorted order.  With
                                    compaction and up to 64 materials the survivors are PLACED by material as the fused
                                    kernel writes them (csrc/pt_types.hpp: RangeDir); PT_UNFUSED | PT_SORT_MATERIAL, or
                                    no compaction, runs the separate intersect -> key histogram -> sorted-shade kernels */
    PT_FAKE_SHADER   = 1u << 2,  /* the reference as shipped: one bounce + shadeFakeMaterial
                                    (pathtrace.cu:224-266,339-377) */
    PT_CACHE_FIRST   = 1u << 3,  /* cache the bounce-0 intersections (INSTRUCTION.md:87-89) */
    PT_UNFUSED       = 1u << 4,  /* debug: separate intersect / shade kernels with the
                                    ShadeableIntersection planes materialised in HBM */
    PT_MESH_BVH      = 1u << 5,  /* cull triangle tests with a bounding-volume hierarchy built at
                                    pt_init (INSTRUCTION.md:129-139,218-240); same winner as the
                                    loop over every triangle */
    PT_AA_JITTER     = 1u << 6,  /* stochastic antialiasing: jitter each camera ray inside its
                                    pixel (the TODO at pathtrace.cu:134; INSTRUCTION.md:110).
                                    Excludes PT_CACHE_FIRST (INSTRUCTION.md:113). */
    PT_PIN_IMAGE     = 1u << 8,  /* opt-in: the host image handed to pt_trace / pt_trace_batch is ONE buffer that stays
                                    allocated at its address until pt_free (the reference's scene->state.image is: sized at
                                    load, scene.cpp:145-147).  The library then page-locks it on first use and, when an
                                    iteration runs as one launch, lets the kernel write the running sums into it over PCIe
                                    while it is still tracing -- EVERY pixel, every call, like the reference's cudaMemcpy
                                    (pathtrace.cu:389-390): whatever the host did to the buffer between two calls is
                                    overwritten.  Without the flag every call copies into whatever buffer it is given
                                    (pageable path): buffers may be freed or reallocated between calls. */
    PT_HOST_SPARSE   = 1u << 10, /* opt-in, with PT_PIN_IMAGE / PT_ASYNC_IMAGE: the host promises to only READ the image
                                    between calls (the reference's host does: main.cpp:78-99 reads it in saveImage, nothing
                                    writes it).  From the second consecutive pt_trace on the launch then writes only the
                                    pixels whose sum changed (a path that ends with colour 0 adds nothing: four in five at
                                    800x800 Cornell): 0.133 ms per call against 0.236.  The library tracks everything IT
                                    does to the accumulation buffer (batches, pt_clear_image, pt_set_image, another host
                                    buffer) and writes every pixel again after such a change; a write by the HOST into the
                                    buffer is not seen -- such pixels stay as the host left them until their sum changes. */
    PT_SHARED_IMAGE  = 1u << 9,  /* tiled sessions (tile_count > 1), pt_trace: host_image_sum is ONE frame shared by all the
                                    ranks that tile it (every process maps the same memory, e.g. POSIX shared memory) under the
                                    rules of PT_PIN_IMAGE and PT_HOST_SPARSE, which it implies.  A rank's launch writes the pixels of its OWN tile
                                    into it and nothing else, so the ranks assemble the frame in host memory with no exchange
                                    between them: it holds the sum after iteration i once every rank's call for i has returned.
                                    Without the flag a tiled session copies its whole accumulation buffer (zeros outside its
                                    tile).  Needs iterations that run as one launch and a mappable buffer: PT_ERR_INVALID if not. */
    PT_LOOKAHEAD     = 1u << 11, /* opt-in: pt_trace TRACES AHEAD of its caller.  The reference's host calls ONE pathtrace() per
                                    iteration (main.cpp:130-140) and a path's whole life is a function of (iteration, pixelIndex,
                                    depth) alone, so the iterations to come can be traced before they are asked for.  pt_trace(iter)
                                    then traces a WINDOW [iter, iter + n) as one path pool (n grows 4, 16, .. up to max_batch; up to three
                                    further windows are traced ahead, beside the one being consumed) and
                                    keeps every sample's final colours; the calls for iter + 1 .. iter + n - 1 only run finalGather
                                    for their own sample -- image[pixel] += colour, the same single addition per pixel and
                                    iteration, in iteration order -- write the pixels whose sum changed into the host image
                                    (PT_PIN_IMAGE | PT_HOST_SPARSE; every pixel otherwise) and tonemap the PBO, while the NEXT
                                    window is already being traced beside them.  state.image, the PBO and the device's
                                    accumulation buffer are complete when each call returns, bit for bit what the same calls
                                    produce without the flag.  A window is discarded (and the iteration traced afresh) when
                                    `iter` is not the next consecutive one, when camera, traceDepth or lens differ from what it
                                    was traced with, and by pt_clear_image, pt_set_image, batches and the stepping interface.
                                    What differs: pt_get_stats reports a window's counts ONCE, with the call that starts
                                    consuming it (the calls served from it afterwards report rays = 0: the sums over a run of
                                    calls are exact), and pt_total_rays / pt_get_counters count a window when it is traced,
                                    ahead of its calls (iterations traced ahead and then discarded stay counted).  With a
                                    page-locked host image (PT_PIN_IMAGE | PT_HOST_SPARSE) on a 256-CU device the windows are
                                    traced on 232 compute units and the calls' gathers write the image from the other 24
                                    (CU-masked streams of the library's own; results unchanged).  Single-device sessions that own the whole frame (tile_count <= 1) on the fused
                                    pipelines; ignored elsewhere (PT_UNFUSED, PT_FAKE_SHADER, PT_CACHE_FIRST, two-kernel sort,
                                    PT_ASYNC_IMAGE, max_batch < 2). */
    PT_ASYNC_IMAGE   = 1u << 7   /* opt-in: pt_trace / pt_trace_batch return without waiting; the copy of the
                                    running sum into host_image_sum overlaps the NEXT call's tracing and is
                                    complete when the next pt_trace / pt_trace_batch returns, or after
                                    pt_synchronize, pt_get_image or pt_free (pt_get_stats needs pt_synchronize).  (pt_trace: the launch
                                    writes the page-locked buffer itself, as under PT_PIN_IMAGE; with PT_HOST_SPARSE only the
                                    pixels that changed.)  Off = the reference's synchronous pathtrace()
                                    (pathtrace.cu:389-392).  Implies the lifetime rule of PT_PIN_IMAGE for the buffers
                                    handed over (they are read by a copy that is still running when the call returns). */
};

typedef struct pt_scene_desc {
    const pt_geom *geoms;          int32_t num_geoms;       /* scene->geoms  (scene.h:23) */
    const pt_material *materials;  int32_t num_materials;   /* scene->materials (scene.h:24) */
    const pt_triangle *triangles;  int32_t num_triangles;   /* optional */
    const pt_mesh *meshes;         int32_t num_meshes;      /* optional */
    pt_camera camera;                                       /* scene->state.camera */
    int32_t trace_depth;                                    /* scene->state.traceDepth */
    uint32_t flags;                                         /* pt_flags */
    int32_t device;                /* HIP device ordinal */
    void *stream;                  /* hipStream_t to launch on, NULL = a private stream */
    /* frame tiling for multi-GPU: this instance owns the rows y with
     * (y / strip_rows) % tile_count == tile_index.  {0,1,*} = whole frame. */
    int32_t tile_index, tile_count, strip_rows;
    int32_t max_batch;             /* iterations in flight in pt_trace_batch (>=1) */
    float *device_image;           /* optional caller-owned device buffer, W*H*3 floats,
                                      used as the accumulation buffer (e.g. a torch tensor
                                      handed to RCCL); NULL = library-owned */
    /* thin-lens depth of field (INSTRUCTION.md:111): rays start on a disc of this radius
     * around camera.position and meet the pinhole ray on the plane focal_distance along
     * camera.view.  0 (a zeroed descriptor) = the reference's pinhole camera. */
    float lens_radius, focal_distance;
    /* Several GPUs of one node behind this one session (SURVEY 8b "Threading", 8e): the frame is tiled over
     * devices[0..num_devices) in interleaved strips of strip_rows rows (0 = 8), every device traces its tile on its
     * own host thread and stream, and after every pt_trace / batch the tiles' running sums travel to devices[0] over
     * RCCL (one grouped ncclSend / ncclRecv exchange on a communicator from ncclCommInitAll; peer copies where RCCL
     * cannot be used), which assembles the frame pt_trace hands back -- bit-identical to the single-device image
     * (global pixelIndex as RNG key).  num_devices == 0 (a zeroed descriptor): the single device `device`, unless
     * the environment names several (PTMI355_DEVICES="0,1,2,3" | "all": the reference's host, which knows one
     * device -- preview.cpp:107 -- then needs no change).  With several devices: `device`, tile_index / tile_count
     * are ignored (tile_count must be 0 or 1), `stream` and `device_image` belong to devices[0], PT_ASYNC_IMAGE is
     * ignored (calls that hand over a host image are synchronous) and the stepping interface is unavailable. */
    const int32_t *devices;
    int32_t num_devices;
} pt_scene_desc;

typedef struct pt_stats {
    int32_t  bounces;              /* bounces executed in the last iteration / batch */
    int64_t  rays;                 /* sum over bounces of live paths traced (the metric's numerator) */
    int32_t  live[64];             /* live[d] = paths traced at bounce d, last iteration / batch */
    int64_t  total_rays;           /* since pt_init */
    int64_t  total_iterations;
} pt_stats;

/* pathtraceInit (pathtrace.cu:79-98): copies geoms/materials/triangles to the
 * device, allocates the path pool, zeroes the accumulation buffer. */
int pt_init(const pt_scene_desc *desc);

/* pathtraceFree (pathtrace.cu:100-112): idempotent, safe before the first init
 * (main.cpp:126 calls it that way). */
void pt_free(void);

/* The reference re-reads camera + traceDepth from the Scene on every
 * pathtrace() (pathtrace.cu:285-286); the shim forwards them here each call.
 * Resolution must not change between pt_init and pt_free. */
int pt_set_camera(const pt_camera *camera, int trace_depth);
int pt_set_lens(float lens_radius, float focal_distance);   /* see pt_scene_desc */

/* pathtrace (pathtrace.cu:284-393): one iteration `iter` (1-based; RNG key and
 * tonemap divisor).  pbo_rgba: optional DEVICE pointer to W*H RGBA8 (the mapped
 * GL PBO in the reference), may be NULL.  host_image_sum: optional HOST buffer
 * of W*H*3 floats that receives the running sum (scene->state.image,
 * pathtrace.cu:389-390), may be NULL.  Synchronous: the buffer is complete
 * when the call returns, and belongs to the caller again (it may be freed).
 * With PT_PIN_IMAGE (see there) buffers of 1 MiB and more are page-locked on
 * first use and written by the kernel itself. */
int pt_trace(uint8_t *pbo_rgba, int frame, int iter, float *host_image_sum);

/* `count` consecutive iterations iter0..iter0+count-1 traced as one path pool
 * (count <= max_batch); bit-identical image to `count` pt_trace calls. */
int pt_trace_batch(int iter0, int count, float *host_image_sum);

/* Asynchronous form: enqueue only; pt_synchronize() waits.  Consecutive calls may OVERLAP on the device (each batch on
 * one of up to four launch streams with its own path pools; their memory is allocated on first use, within
 * PTMI355_OVERLAP_GB; PTMI355_OVERLAP=0 switches this off): the image is still summed in iteration order, bit for bit,
 * and everything enqueued afterwards on the session's stream -- pt_trace, pt_get_image, pt_tonemap, pt_synchronize --
 * comes after every batch enqueued before it. */
int pt_trace_batch_async(int iter0, int count);
int pt_synchronize(void);

/* ---- stepping interface (parity tests drive the loop bounce by bounce) ----- */
int pt_trace_begin(int iter0, int count);            /* generateRayFromCamera, pathtrace.cu:329 */
int pt_trace_bounce(int depth, int *n_live_after);   /* one pass of the loop body, :340-377 + 8.0 */
int pt_trace_end(void);                              /* finalGather, :380-381 */
/* current path pool as the reference's AoS (live prefix first); returns count */
int pt_export_paths(pt_path_segment *host_paths, int capacity, int *n_live);
/* ShadeableIntersection records of the last bounce (PT_UNFUSED / sort / fake-shader modes) */
int pt_export_intersections(pt_shadeable_intersection *host_isects, uint8_t *host_outside,
                            int capacity);
/* computeIntersections (pathtrace.cu:149-213) on caller-supplied rays: host AoS in,
 * host AoS out; runs the production intersect kernel. */
int pt_intersect_once(const pt_path_segment *host_paths, int n,
                      pt_shadeable_intersection *host_isects, uint8_t *host_outside);

/* ---- results ----------------------------------------------------------------- */
int pt_get_image(float *host_image_sum);             /* W*H*3 floats, running sum */
int pt_tonemap(uint8_t *host_rgba, int iter);        /* sendImageToPBO (pathtrace.cu:48-68) to host */
int pt_clear_image(void);
/* Resume an accumulation: the running sum becomes `host_image_sum` (W*H*3 floats, e.g. what pt_get_image / the host's
 * PFM dump returned after iteration k); tracing iterations k+1.. then yields, bit for bit, the image of the
 * uninterrupted run -- the running sum is the whole state the reference carries from one iteration to the next
 * (dev_image, pathtrace.cu:71,84,389; the iteration number is the caller's).  Synchronises the session first. */
int pt_set_image(const float *host_image_sum);
float *pt_device_image(void);                        /* device pointer of the accumulation buffer */
int pt_get_stats(pt_stats *stats);
/* rays traced since pt_init, read from the device-side counter (includes
 * asynchronous batches); synchronises the stream.  Negative = pt_status. */
long long pt_total_rays(void);
/* the same counter plus the paths traced at bounce 0 (rays - first = paths that survived a
 * compaction) and the iterations traced */
int pt_get_counters(int64_t *rays, int64_t *first_bounce_rays, int64_t *iterations);

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's
 * roofline leg).  Off by default; when on, every launch of the per-bounce
 * kernels is bracketed by two events from a preallocated pool. */
enum pt_stage { PT_STAGE_RAYGEN = 0, PT_STAGE_BOUNCE = 1, PT_STAGE_INTERSECT = 2,
                PT_STAGE_SORT = 3, PT_STAGE_GATHER = 4, PT_STAGE_MESH = 5, PT_STAGE_COUNT = 6 };
typedef struct pt_profile {
    double  ms[PT_STAGE_COUNT];        /* summed event-elapsed time per stage */
    int64_t launches[PT_STAGE_COUNT];  /* launches measured */
} pt_profile;
int pt_set_profiling(int enable);      /* also clears the accumulated profile */
int pt_get_profile(pt_profile *out);   /* synchronises the stream, drains pending events */
/* PT_MESH_BVH: what pt_init built (all meshes together) */
typedef struct pt_bvh_info {
    int32_t nodes, triangles, depth;
    float   pad;                       /* box padding, world units (= 2 x the pad of the spec's hit-point test) */
    float   prune;                     /* additive slack of the distance prune: 0 (none is needed, csrc/pt_bvh.hpp) */
} pt_bvh_info;
int pt_get_bvh_info(pt_bvh_info *out);
/* host-only (no GPU): build the hierarchy of `count` triangles; returns the node count, or the
 * required count when `node_capacity` is too small (nothing written then).  nodes: 16 dwords
 * each (layout in csrc/pt_bvh.hpp); order: leaf slot -> triangle index; grid (8 floats,
 * optional): origin xyz, step xyz of the 16-bit box grid, box padding, prune margin. */
int pt_bvh_build(const pt_triangle *triangles, int count, float *nodes, int node_capacity,
                 int32_t *order, float *grid);
/* host-only (no GPU): the world-space cull boxes pt_init derives for `count` primitives seen from `eye` (3 floats,
 * may be NULL): boxes = count x {lo.xyz, hi.xyz}; *origin_bound = the |origin|_1 up to which they hold.  A ray
 * outside a primitive's box is never handed to the exact test (csrc/pt_cull.hpp has the error bound that makes
 * this identical to the reference's loop over every primitive, pathtrace.cu:176-199).  reject (optional) = count x
 * {mode (0..2 diagonal row k, 4 general row, 3 none), m_k0, m_k1, m_k2, m_k3}: the row of the inverseTransform the
 * kernel's exact one-axis early miss evaluates for cubes (same file). */
int pt_cull_boxes(const pt_geom *geoms, int count, const float *eye, float *boxes, float *origin_bound, float *reject);
/* host-only (no GPU): the per-triangle spheres of the every-triangle loop's first stage for ONE mesh of `count` triangles and
 * ray origins with |origin|_1 <= origin_bound: bounds = ((count + 3) & ~3) x {centre xyz, Rs^2} (padding entries: Rs^2 = -1).
 * A ray whose line passes a centre at more than Rs is never accepted for that triangle by the completion spec
 * (glm::intersectRayTriangle + the hit-point test; csrc/pt_h_scene.hpp: make_tri_bounds has the derivation), so the kernel
 * does not run the exact test for the pair.  Returns the number of entries written. */
int pt_tri_bounds(const pt_triangle *triangles, int count, float origin_bound, float *bounds);
/* host-only (no GPU): the same spheres as the every-triangle loop's first stage reads them -- it evaluates "line passes the centre
 * at more than Rs" on the matrix pipe (v_mfma_f32_16x16x32_f16) as ONE bilinear form per (ray, triangle) pair, in the mesh's own
 * frame (centre g, scale 1 / Rm: every sphere in the unit ball): records = ((count + 63) & ~63) x 32 binary16 K-slots (the triangle's
 * side of the form: csrc/pt_h_scene.hpp: make_tri_records; padding records reach nobody), frame = {gx, gy, gz, 1 / Rm}.  The ray's
 * side and the error budget: csrc/pt_k_trisweep.hpp.  Returns the number of records written. */
int pt_tri_records(const pt_triangle *triangles, int count, float origin_bound, uint16_t *records, float frame[4]);
/* ---- known-answer probes: the DEVICE's own arithmetic on caller data (no session needed, any HIP device) -------
 * pt_probe_rng: thrust::default_random_engine (minstd_rand) as makeSeededRandomEngine constructs it
 * (pathtrace.cu:41-45 -> engine(seed)): for each of the n seeds, seed the engine and draw `draws` times through
 * uniform_real_distribution<float>(0,1) (csrc/pt_device.hpp: lcg_seed, u01); state[i] = the engine's state after the
 * last draw, u[i] = the last draw's value (both optional).  Known answer (C++ [rand.predef]): seed 1, 10 000 draws ->
 * state 399268537.
 * pt_probe_sincos: the shared sin / cos of calculateRandomDirectionInHemisphere (interactions.h:40-41; DESIGN.md
 * section 4) for n arguments, or -- x == NULL -- for the `n` consecutive binary32 values that start at bit pattern
 * `first_bits`, reduced to sum[0] = sum over k of bits(sin) * (2k + 1), sum[1] = the same for cos (mod 2^64): the
 * oracle computes the same two sums on the CPU, so every float of [0, 2 pi] can be compared without moving 9 GB.
 * pt_probe_hemisphere: calculateRandomDirectionInHemisphere (interactions.h:10-42) for n (normal, engine seed)
 * pairs; dirs = n x 3 floats.
 * pt_probe_sqrt: the kernels' sqrt and 1 / sqrt (glm::length / glm::normalize, func_geometric.inl:94-100,153-159:
 * Newton's iteration on v_rsq_f32, csrc/pt_device.hpp: sqrt_newton) against the correctly rounded sqrtf and divide, for
 * the `n` consecutive binary32 values from bit pattern `first_bits` on: mismatch[0] = arguments whose root differs,
 * mismatch[1] = whose reciprocal of the root differs.  Zero for every x in [2^-102, 2^128).  For x in [1 - 2^-12, 1 + 2^-12]
 * the reciprocal is the kernels' four-addition form for vectors that are unit vectors up to rounding (rsqrt_near_one). */
/* pt_probe_clock: the shader clock (GHz) the device holds WHILE whatever is enqueued runs -- one wave counts its cycle counter
 * against the constant 100-MHz counter for `microseconds` on a stream of its own, beside the session's launches (bench.py's
 * `sustained` object: the issue roof is priced at the 2.4 GHz peak, the chip holds 2.1-2.2 under this library's kernels). */
int pt_probe_clock(int microseconds, double *ghz);
int pt_probe_rng(const uint32_t *seeds, int n, int draws, uint32_t *state, float *u);
int pt_probe_sincos(const float *x, uint32_t first_bits, uint32_t n, float *s, float *c, uint64_t sum[2]);
int pt_probe_hemisphere(const float *normals, const uint32_t *seeds, int n, float *dirs);
int pt_probe_sqrt(uint32_t first_bits, uint32_t n, uint64_t mismatch[2]);
/* devices of the current session (0: not initialised) and how their tiles reach devices[0]: "rccl", "peer"
 * (hipMemcpyPeerAsync) or "none" (one device) */
int pt_num_devices(void);
const char *pt_exchange_transport(void);
const char *pt_last_error(void);
const char *pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTMI355_H */
