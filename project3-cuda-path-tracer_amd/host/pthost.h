/*
 * pthost.h -- headless host side of the path tracer: the counterpart of the
 * reference's scene loader (src/scene.cpp, src/utilities.cpp:65-72), of the camera set-up that
 * runCuda performs before the first pathtrace() (src/main.cpp:53-67,102-120) and of saveImage /
 * image::savePNG (src/main.cpp:78-99, src/image.cpp:22-39).  Plain C ABI so that tests can drive
 * it through ctypes; structs are the C-ABI mirrors of include/ptmi355.h.
 */
#ifndef PTHOST_H
#define PTHOST_H

#include "../../include/ptmi355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene {
    pt_geom *geoms;          int32_t num_geoms;
    pt_material *materials;  int32_t num_materials;
    pt_triangle *triangles;  int32_t num_triangles;   /* "mesh" objects (format extension) */
    pt_mesh *meshes;         int32_t num_meshes;
    pt_camera camera_loaded;   /* exactly as Scene::loadCamera leaves it (right = NaN, scene.cpp:138) */
    pt_camera camera;          /* after the runCuda orbit recompute: what the kernels must see */
    int32_t iterations;        /* ITERATIONS */
    int32_t trace_depth;       /* DEPTH */
    char image_name[256];      /* FILE */
} pth_scene;

/* Scene::Scene(filename) (scene.cpp:7-33).  Returns NULL and sets pth_last_error() on failure. */
pth_scene *pth_load_scene(const char *path);
void pth_free_scene(pth_scene *s);
const char *pth_last_error(void);

/* utilityCore::buildTransformationMatrix (utilities.cpp:65-72) + glm::inverse + glm::inverseTranspose
 * (scene.cpp:82-85), GLM 0.9.6.3 operation order */
void pth_build_geom_matrices(pt_geom *g);

/* saveImage + image::savePNG: pixel = sum / samples, x-flip, clamp [0,1], * 255.f, truncate.
 * rgb_out: W*H*3 bytes in file order. */
void pth_image_to_rgb8(const float *image_sum, int w, int h, float samples, uint8_t *rgb_out);
int pth_write_png(const char *path, const uint8_t *rgb, int w, int h);          /* stored-deflate PNG */
int pth_write_pfm(const char *path, const float *image_sum, int w, int h, float samples);
/* the floats of a little-endian colour PFM of exactly w x h pixels (times |scale|), rows top to bottom as in memory:
 * with samples = 1 on the writing side this is the running sum itself (the reference keeps nothing else between
 * iterations, pathtrace.cu:71,84,389), exact */
int pth_read_pfm(const char *path, float *image_sum, int w, int h);

#ifdef __cplusplus
}
#endif
#endif
