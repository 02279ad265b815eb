// ptbench.cpp -- headless host: the reference's main.cpp / runCuda() loop without GLFW
// (src/main.cpp:33-76,101-147): load the scene, recompute the camera, pathtraceFree();
// pathtraceInit(); pathtrace() x ITERATIONS; saveImage(); pathtraceFree().
//
//   ptbench SCENEFILE.txt [--iters N] [--batch B] [--out BASENAME] [--sort] [--no-compact]
//           [--cache-first] [--bvh] [--aa] [--lens RADIUS FOCAL] [--pfm] [--device D] [--tile R/K] [--strip-rows S]
//           [--gpus K | --devices D0,D1,...] [--save-sum] [--resume SUMFILE.pfm [--start N]]
//
// --batch B: iterations per launch sequence (default: up to 64 and ~40 M paths, as the shim sizes its windows; 1 = one
// pathtrace() per iteration); every batch but the last is enqueued without waiting, so consecutive batches overlap on the
// device.  The image is the same bit for bit whatever B.
//
// --save-sum / --resume: a render across several runs (C5's 5000 spp across GPU leases).  The running sum is the whole
// state the reference carries from one iteration to the next (dev_image, pathtrace.cu:71,84,389): --save-sum writes it
// raw (BASE.<N>samp.sum.pfm, exact), --resume loads it (pt_set_image) and continues with iteration N + 1 (N from
// --start or from the "<N>samp" of the file name) up to --iters: bit for bit the image of the uninterrupted run.
//
// --gpus K / --devices LIST: ONE process renders the frame on several GPUs: the library tiles it over the devices
// (interleaved strips of S rows), traces every tile on its own host thread and gathers the tiles' sums onto the first
// device over RCCL after every call (include/ptmi355.h: pt_scene_desc::devices); the image is the single-GPU image.
//
// --tile R/K: this process renders only tile R of K (the rows y with (y / S) % K == R, global pixelIndex and RNG
// keys unchanged), so K processes -- one per GPU, --device each -- render one frame between them; the other rows
// of its image stay zero and the K raw sums (--pfm) add up, exactly, to the single-process image.
//
// Links libptmi355.so (the HIP library) and host/pthost.cpp.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pthost.h"

int main(int argc, char **argv) {
    if (argc < 2) {
        printf("Usage: %s SCENEFILE.txt [--iters N] [--batch B] [--out BASE] [--sort] [--no-compact] [--cache-first] "
               "[--bvh] [--aa] [--lens RADIUS FOCAL] [--pfm] [--device D] [--tile R/K] [--strip-rows S] "
               "[--gpus K | --devices D0,D1,...] [--save-sum] [--resume SUMFILE.pfm [--start N]]\n", argv[0]);
        return 1;
    }
    int iters = -1, batch = 0, device = 0, tile_index = 0, tile_count = 1, strip_rows = 8;
    unsigned flags = PT_COMPACT | PT_PIN_IMAGE | PT_HOST_SPARSE;      // `image` below lives until pt_free and is only read here
    bool pfm = false, save_sum = false;
    std::string resume;
    int start = -1;
    float lens_radius = 0.0f, focal_distance = 0.0f;
    std::string out;
    std::vector<int32_t> devices;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
        else if (a == "--batch" && i + 1 < argc) batch = atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--sort") flags |= PT_SORT_MATERIAL;
        else if (a == "--no-compact") flags &= ~PT_COMPACT;
        else if (a == "--cache-first") flags |= PT_CACHE_FIRST;
        else if (a == "--bvh") flags |= PT_MESH_BVH;
        else if (a == "--aa") flags |= PT_AA_JITTER;
        else if (a == "--lens" && i + 2 < argc) { lens_radius = (float)atof(argv[++i]); focal_distance = (float)atof(argv[++i]); }
        else if (a == "--pfm") pfm = true;
        else if (a == "--save-sum") save_sum = true;
        else if (a == "--resume" && i + 1 < argc) resume = argv[++i];
        else if (a == "--start" && i + 1 < argc) start = atoi(argv[++i]);
        else if (a == "--tile" && i + 1 < argc) {
            if (sscanf(argv[++i], "%d/%d", &tile_index, &tile_count) != 2 || tile_count < 1 || tile_index < 0 || tile_index >= tile_count) {
                fprintf(stderr, "--tile wants R/K with 0 <= R < K\n");
                return 1;
            }
        }
        else if (a == "--strip-rows" && i + 1 < argc) strip_rows = atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) { devices.clear(); for (int k = 0, n = atoi(argv[++i]); k < n; ++k) devices.push_back(k); }
        else if (a == "--devices" && i + 1 < argc) {
            devices.clear();
            for (const char *p = argv[++i]; *p;) { devices.push_back((int32_t)strtol(p, (char **)&p, 10)); if (*p == ',') ++p; else if (*p) { fprintf(stderr, "--devices wants D0,D1,...\n"); return 1; } }
        }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
    }
    pth_scene *sc = pth_load_scene(argv[1]);
    if (!sc) { fprintf(stderr, "%s\n", pth_last_error()); return 1; }
    if (iters < 0) iters = sc->iterations;
    if (batch < 1) {                                      // not given: as the shim sizes its windows -- up to 64 iterations and ~40 M paths per batch
        const long long pixels = (long long)sc->camera.resolution[0] * sc->camera.resolution[1];
        const long long k = 41000000LL / (pixels > 0 ? pixels : 1);
        batch = (int)(k < 4 ? 4 : (k > 64 ? 64 : k));
    }
    const int W = sc->camera.resolution[0], H = sc->camera.resolution[1];
    printf("scene %s: %d geoms, %d materials, %d triangles, %dx%d, depth %d, %d iterations\n", argv[1],
           sc->num_geoms, sc->num_materials, sc->num_triangles, W, H, sc->trace_depth, iters);

    pt_scene_desc d;
    memset(&d, 0, sizeof d);
    d.geoms = sc->geoms; d.num_geoms = sc->num_geoms;
    d.materials = sc->materials; d.num_materials = sc->num_materials;
    d.triangles = sc->triangles; d.num_triangles = sc->num_triangles;
    d.meshes = sc->meshes; d.num_meshes = sc->num_meshes;
    d.camera = sc->camera; d.trace_depth = sc->trace_depth; d.flags = flags; d.device = device;
    d.tile_index = tile_index; d.tile_count = tile_count; d.strip_rows = strip_rows; d.max_batch = batch;
    d.lens_radius = lens_radius; d.focal_distance = focal_distance;
    if (!devices.empty()) { d.devices = devices.data(); d.num_devices = (int32_t)devices.size(); }
    pt_free();                                            // main.cpp:126
    if (pt_init(&d) != PT_OK) { fprintf(stderr, "pathtraceInit: %s\n", pt_last_error()); return 1; }

    std::vector<float> image((size_t)W * H * 3, 0.0f);    // scene->state.image
    int iteration = 0;
    if (!resume.empty()) {
        if (start < 0) {                                   // BASE.<N>samp.sum.pfm
            const size_t e = resume.rfind("samp");
            size_t b = e;
            while (b != std::string::npos && b > 0 && resume[b - 1] >= '0' && resume[b - 1] <= '9') --b;
            if (e == std::string::npos || b == e) { fprintf(stderr, "--resume: no <N>samp in %s, say --start N\n", resume.c_str()); return 1; }
            start = atoi(resume.substr(b, e - b).c_str());
        }
        if (start < 0 || start > iters) { fprintf(stderr, "--resume: %d iterations done, %d wanted\n", start, iters); return 1; }
        if (pth_read_pfm(resume.c_str(), image.data(), W, H) != 0) { fprintf(stderr, "%s\n", pth_last_error()); return 1; }
        if (pt_set_image(image.data()) != PT_OK) { fprintf(stderr, "pt_set_image: %s\n", pt_last_error()); return 1; }
        iteration = start;
        printf("resumed %s: %d iterations done\n", resume.c_str(), start);
    }
    const int first_iteration = iteration;
    const auto t0 = std::chrono::steady_clock::now();
    while (iteration < iters) {                            // runCuda: iteration++ ; pathtrace(pbo, 0, iteration)
        const int n = (iters - iteration < batch) ? iters - iteration : batch;
        const int last = (iteration + n == iters);
        // every batch but the last is only enqueued: consecutive batches overlap on the device; the last one
        // waits for all of them and brings the image
        int rc = (n == 1) ? pt_trace(NULL, 0, iteration + 1, last ? image.data() : NULL)
                 : last   ? pt_trace_batch(iteration + 1, n, image.data())
                          : pt_trace_batch_async(iteration + 1, n);
        if (rc != PT_OK) { fprintf(stderr, "pathtrace: %s\n", pt_last_error()); return 1; }
        iteration += n;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const long long rays = pt_total_rays();
    printf("%d iterations, %lld rays, %.3f s, %.1f Mrays/s on %d device(s), tile exchange: %s\n", iteration - first_iteration, rays, sec, rays / sec / 1e6,
           pt_num_devices(), pt_exchange_transport());
    if (iteration == first_iteration && pt_get_image(image.data()) != PT_OK) { fprintf(stderr, "%s\n", pt_last_error()); return 1; }

    if (out.empty()) out = std::string(sc->image_name[0] ? sc->image_name : "render");
    char name[512];
    snprintf(name, sizeof name, "%s.%dsamp.png", out.c_str(), iteration);   // saveImage(): <FILE>.<time>.<N>samp
    std::vector<uint8_t> rgb((size_t)W * H * 3);
    pth_image_to_rgb8(image.data(), W, H, (float)iteration, rgb.data());
    if (pth_write_png(name, rgb.data(), W, H) != 0) { fprintf(stderr, "%s\n", pth_last_error()); return 1; }
    printf("Saved %s.\n", name);
    if (pfm) {
        snprintf(name, sizeof name, "%s.%dsamp.pfm", out.c_str(), iteration);
        pth_write_pfm(name, image.data(), W, H, (float)iteration);
        printf("Saved %s.\n", name);
    }
    if (save_sum) {
        snprintf(name, sizeof name, "%s.%dsamp.sum.pfm", out.c_str(), iteration);
        if (pth_write_pfm(name, image.data(), W, H, 1.0f) != 0) { fprintf(stderr, "%s\n", pth_last_error()); return 1; }
        printf("Saved %s.\n", name);
    }
    pt_free();
    pth_free_scene(sc);
    return 0;
}
