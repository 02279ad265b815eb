// ref_wrapper.cpp -- ORACLE INFRASTRUCTURE (this container only; NOT product code).
//
// Thin extern "C" driver around the REFERENCE's own headers, included by path
// from /root/reference (never copied).  It is compiled by oracle/Makefile into
// oracle/_ref/*.so (git-ignored) in two flavours:
//
//   REF_TU_A  g++  + the genuine CUDA toolkit headers that ship in this image
//             (triton/backends/nvidia/include/cuda_runtime.h) + vendored GLM:
//             src/intersections.h (utilhash, getPointOnRay, multiplyMV, box /
//             sphere tests), src/scene.cpp + src/utilities.cpp (the loader).
//             No stand-in header of any kind.
//   REF_TU_B  hipcc --offload-host-only + rocThrust 2.8.5 (the installed
//             thrust; arithmetic identical to CUDA thrust) +
//             a one-line forwarding <cuda_runtime.h> -> <hip/hip_runtime.h>
//             (rocThrust and NVIDIA's vector_types.h cannot share a TU):
//             src/interactions.h (calculateRandomDirectionInHemisphere), the
//             thrust RNG, and the whole-iteration driver loop.
//             TU_B also re-exports the intersection tests so tests can assert
//             TU_A == TU_B bit-for-bit (i.e. the forwarding header changes
//             nothing).
//
// The reference calls unqualified cos/sin/sqrt/abs/min/max (interactions.h:
// 15-16,25-27,40-41; intersections.h:119,128,131).  CUDA binds them to float
// overloads; a host pass must be told to.  REF_TRIG_SHARED binds
// cos/sin to the oracle's pto_sincos instead of libm -- the variant that pins
// the GPU bit-for-bit.
#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>
#include <string>
#include <cstdint>

using std::sqrt; using std::abs; using std::min; using std::max;
#if defined(REF_TRIG_SHARED)
extern "C" void pto_sincos(float x, float *s, float *c);
static inline float cos(float x) { float s, c; pto_sincos(x, &s, &c); return c; }
static inline float sin(float x) { float s, c; pto_sincos(x, &s, &c); return s; }
#else
using std::cos; using std::sin;
#endif

#if defined(REF_TU_B)
#include <thrust/random.h>
#endif

#include "sceneStructs.h"
#include "intersections.h"
#if defined(REF_TU_B)
#include "interactions.h"
#endif
#if defined(REF_TU_A)
#include "scene.h"
#include "image.h"
#endif

static_assert(sizeof(Geom) == 236 && sizeof(Material) == 44 && sizeof(Camera) == 84 &&
              sizeof(PathSegment) == 44 && sizeof(ShadeableIntersection) == 20 &&
              sizeof(Ray) == 24, "reference struct ABI (SURVEY 8b)");

extern "C" {

int ref_abi(int *out) {   // sizes + offsets for the ABI test
    int i = 0;
    out[i++] = sizeof(Geom); out[i++] = offsetof(Geom, transform);
    out[i++] = offsetof(Geom, inverseTransform); out[i++] = offsetof(Geom, invTranspose);
    out[i++] = sizeof(Material); out[i++] = offsetof(Material, specular);
    out[i++] = offsetof(Material, hasReflective); out[i++] = offsetof(Material, emittance);
    out[i++] = sizeof(Camera); out[i++] = offsetof(Camera, position);
    out[i++] = offsetof(Camera, view); out[i++] = offsetof(Camera, fov);
    out[i++] = offsetof(Camera, pixelLength);
    out[i++] = sizeof(PathSegment); out[i++] = offsetof(PathSegment, color);
    out[i++] = offsetof(PathSegment, pixelIndex); out[i++] = offsetof(PathSegment, remainingBounces);
    out[i++] = sizeof(ShadeableIntersection); out[i++] = offsetof(ShadeableIntersection, materialId);
    out[i++] = (int)SPHERE; out[i++] = (int)CUBE;
    return i;
}

unsigned ref_utilhash(unsigned a) { return utilhash(a); }

void ref_get_point_on_ray(const Ray *r, float t, float *out3) {
    glm::vec3 p = getPointOnRay(*r, t);
    out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
}

void ref_multiply_mv(const float *m16, const float *v4, float *out3) {
    glm::mat4 m; std::memcpy(&m, m16, 64);
    glm::vec3 p = multiplyMV(m, glm::vec4(v4[0], v4[1], v4[2], v4[3]));
    out3[0] = p.x; out3[1] = p.y; out3[2] = p.z;
}

// out: t, point[3], normal[3], outside  (8 floats; outside as 0/1; untouched
// outputs keep the sentinel the caller put there)
void ref_box(const Geom *g, const Ray *rays, int n, float *out8) {
    for (int i = 0; i < n; ++i) {
        glm::vec3 p(out8[8 * i + 1], out8[8 * i + 2], out8[8 * i + 3]);
        glm::vec3 nn(out8[8 * i + 4], out8[8 * i + 5], out8[8 * i + 6]);
        bool outside = out8[8 * i + 7] != 0.0f;
        float t = boxIntersectionTest(*g, rays[i], p, nn, outside);
        float *o = out8 + 8 * i;
        o[0] = t; o[1] = p.x; o[2] = p.y; o[3] = p.z; o[4] = nn.x; o[5] = nn.y; o[6] = nn.z;
        o[7] = outside ? 1.0f : 0.0f;
    }
}

void ref_sphere(const Geom *g, const Ray *rays, int n, float *out8) {
    for (int i = 0; i < n; ++i) {
        glm::vec3 p(out8[8 * i + 1], out8[8 * i + 2], out8[8 * i + 3]);
        glm::vec3 nn(out8[8 * i + 4], out8[8 * i + 5], out8[8 * i + 6]);
        bool outside = out8[8 * i + 7] != 0.0f;
        float t = sphereIntersectionTest(*g, rays[i], p, nn, outside);
        float *o = out8 + 8 * i;
        o[0] = t; o[1] = p.x; o[2] = p.y; o[3] = p.z; o[4] = nn.x; o[5] = nn.y; o[6] = nn.z;
        o[7] = outside ? 1.0f : 0.0f;
    }
}

// GLM functions the completion spec leans on (vendored GLM 0.9.6.3)
void ref_glm_reflect(const float *I, const float *N, float *out3) {
    glm::vec3 r = glm::reflect(glm::vec3(I[0], I[1], I[2]), glm::vec3(N[0], N[1], N[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void ref_glm_refract(const float *I, const float *N, float eta, float *out3) {
    glm::vec3 r = glm::refract(glm::vec3(I[0], I[1], I[2]), glm::vec3(N[0], N[1], N[2]), eta);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
int ref_glm_ray_triangle(const float *o, const float *d, const float *v9, float *bary3) {
    glm::vec3 b(bary3[0], bary3[1], bary3[2]);
    bool hit = glm::intersectRayTriangle(glm::vec3(o[0], o[1], o[2]), glm::vec3(d[0], d[1], d[2]),
                                         glm::vec3(v9[0], v9[1], v9[2]), glm::vec3(v9[3], v9[4], v9[5]),
                                         glm::vec3(v9[6], v9[7], v9[8]), b);
    bary3[0] = b.x; bary3[1] = b.y; bary3[2] = b.z;
    return hit ? 1 : 0;
}

// computeIntersections (pathtrace.cu:149-213) driven through the reference's
// tests: the loop body is the wrapper's, the arithmetic is the reference's.
void ref_compute_intersections(int n, const PathSegment *paths, const Geom *geoms, int ngeoms,
                               ShadeableIntersection *isects, unsigned char *outside_out) {
    for (int p = 0; p < n; ++p) {
        PathSegment ps = paths[p];
        float t, tmin = FLT_MAX; int hit = -1; bool outside = true, hit_outside = true;
        glm::vec3 ti, tn, nrm;
        for (int g = 0; g < ngeoms; ++g) {
            const Geom &geom = geoms[g];
            if (geom.type == CUBE) t = boxIntersectionTest(geom, ps.ray, ti, tn, outside);
            else if (geom.type == SPHERE) t = sphereIntersectionTest(geom, ps.ray, ti, tn, outside);
            if (t > 0.0f && tmin > t) { tmin = t; hit = g; nrm = tn; hit_outside = outside; }
        }
        if (hit == -1) isects[p].t = -1.0f;
        else { isects[p].t = tmin; isects[p].materialId = geoms[hit].materialid; isects[p].surfaceNormal = nrm; }
        if (outside_out) outside_out[p] = hit_outside ? 1 : 0;
    }
}

#if defined(REF_TU_A)
// ---- the reference's loader (scene.cpp, utilities.cpp), compiled as is ----
// Scene::~Scene is declared but never defined (scene.h:21): only ever `new`.
// returns 0 on success; copies up to the given capacities.
int ref_load_scene(const char *path, Geom *geoms, int cap_g, int *ng, Material *mats, int cap_m,
                   int *nm, Camera *cam, int *iterations, int *traceDepth, char *name64) {
    FILE *old = stdout; (void)old;
    Scene *s = new Scene(std::string(path));
    *ng = (int)s->geoms.size(); *nm = (int)s->materials.size();
    for (int i = 0; i < *ng && i < cap_g; ++i) geoms[i] = s->geoms[i];
    for (int i = 0; i < *nm && i < cap_m; ++i) mats[i] = s->materials[i];
    *cam = s->state.camera;
    *iterations = (int)s->state.iterations;
    *traceDepth = s->state.traceDepth;
    std::strncpy(name64, s->state.imageName.c_str(), 63); name64[63] = 0;
    return 0;
}

// main.cpp:53-67 (derive phi/theta/zoom) + main.cpp:102-120 (runCuda's
// recompute), same GLM calls in the same order; main.cpp itself needs the
// GLFW/GL stack and cannot be built headless.
void ref_camera_orbit(Camera *camp) {
    Camera &cam = *camp;
    glm::vec3 view = cam.view;
    glm::vec3 up = cam.up;
    glm::vec3 right = glm::cross(view, up);
    up = glm::cross(right, view);
    glm::vec3 cameraPosition = cam.position;
    glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
    glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
    float phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
    float theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
    glm::vec3 ogLookAt = cam.lookAt;
    float zoom = glm::length(cam.position - ogLookAt);
    // runCuda, camchanged == true
    cameraPosition.x = zoom * sin(phi) * sin(theta);
    cameraPosition.y = zoom * cos(theta);
    cameraPosition.z = zoom * cos(phi) * sin(theta);
    cam.view = -glm::normalize(cameraPosition);
    glm::vec3 v = cam.view;
    glm::vec3 u = glm::vec3(0, 1, 0);
    glm::vec3 r = glm::cross(v, u);
    cam.up = glm::cross(r, v);
    cam.right = r;
    cam.position = cameraPosition;
    cameraPosition += cam.lookAt;
    cam.position = cameraPosition;
}

// saveImage (main.cpp:78-99) through the reference's image class + stb_image_write (image.cpp,
// stb.cpp compiled as they are): writes <base>.png
void ref_save_image(const float *image_sum, int width, int height, float samples, const char *base) {
    image img(width, height);
    for (int x = 0; x < width; x++) {
        for (int y = 0; y < height; y++) {
            int index = x + (y * width);
            glm::vec3 pix(image_sum[3 * index], image_sum[3 * index + 1], image_sum[3 * index + 2]);
            img.setPixel(width - 1 - x, y, glm::vec3(pix) / samples);
        }
    }
    img.savePNG(std::string(base));
}
#endif  // REF_TU_A

#if defined(REF_TU_B)
// thrust::default_random_engine seeded with s; emits n raw states and n u01 draws
void ref_rng_sequence(unsigned seed, int n, unsigned *raw_out, float *u01_out) {
    if (raw_out) { thrust::default_random_engine e(seed); for (int i = 0; i < n; ++i) raw_out[i] = e(); }
    if (u01_out) {
        thrust::default_random_engine e(seed);
        thrust::uniform_real_distribution<float> u01(0, 1);
        for (int i = 0; i < n; ++i) u01_out[i] = u01(e);
    }
}

// makeSeededRandomEngine (pathtrace.cu:41-45) -- same expression, reference's
// utilhash + thrust engine
static thrust::default_random_engine mk(int iter, int index, int depth) {
    int h = utilhash((1 << 31) | (depth << 22) | iter) ^ utilhash(index);
    return thrust::default_random_engine(h);
}
unsigned ref_seeded_first_raw(int iter, int index, int depth) {
    thrust::default_random_engine e = mk(iter, index, depth);
    return e();
}

void ref_hemisphere(const float *normals3, const unsigned *seeds, int n, float *out3) {
    for (int i = 0; i < n; ++i) {
        thrust::default_random_engine rng(seeds[i]);
        glm::vec3 d = calculateRandomDirectionInHemisphere(
            glm::vec3(normals3[3 * i], normals3[3 * i + 1], normals3[3 * i + 2]), rng);
        out3[3 * i] = d.x; out3[3 * i + 1] = d.y; out3[3 * i + 2] = d.z;
    }
}

// shadeFakeMaterial (pathtrace.cu:224-266) cannot be compiled (it lives in the
// .cu); its arithmetic is GLM operators + thrust u01, replayed here verbatim
// in structure so the oracle's restatement can be checked.
void ref_shade_fake(int iter, int n, const ShadeableIntersection *isx, PathSegment *paths,
                    const Material *materials) {
    for (int idx = 0; idx < n; ++idx) {
        ShadeableIntersection intersection = isx[idx];
        if (intersection.t > 0.0f) {
            thrust::default_random_engine rng = mk(iter, idx, 0);
            thrust::uniform_real_distribution<float> u01(0, 1);
            Material material = materials[intersection.materialId];
            glm::vec3 materialColor = material.color;
            if (material.emittance > 0.0f) {
                paths[idx].color *= (materialColor * material.emittance);
            } else {
                float lightTerm = glm::dot(intersection.surfaceNormal, glm::vec3(0.0f, 1.0f, 0.0f));
                paths[idx].color *= (materialColor * lightTerm) * 0.3f +
                                    ((1.0f - intersection.t * 0.02f) * materialColor) * 0.7f;
                paths[idx].color *= u01(rng);
            }
        } else {
            paths[idx].color = glm::vec3(0.0f);
        }
    }
}

void ref_generate_rays(const Camera *camp, int traceDepth, PathSegment *paths) {
    const Camera &cam = *camp;
    for (int y = 0; y < cam.resolution.y; ++y)
        for (int x = 0; x < cam.resolution.x; ++x) {
            int index = x + (y * cam.resolution.x);
            PathSegment &segment = paths[index];
            segment.ray.origin = cam.position;
            segment.color = glm::vec3(1.0f, 1.0f, 1.0f);
            segment.ray.direction = glm::normalize(cam.view
                - cam.right * cam.pixelLength.x * ((float)x - (float)cam.resolution.x * 0.5f)
                - cam.up * cam.pixelLength.y * ((float)y - (float)cam.resolution.y * 0.5f));
            segment.pixelIndex = index;
            segment.remainingBounces = traceDepth;
        }
}

void ref_send_image_to_pbo(unsigned char *pbo, int w, int h, int iter, const glm::vec3 *image) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int index = x + (y * w);
            glm::vec3 pix = image[index];
            glm::ivec3 color;
            color.x = glm::clamp((int)(pix.x / iter * 255.0), 0, 255);
            color.y = glm::clamp((int)(pix.y / iter * 255.0), 0, 255);
            color.z = glm::clamp((int)(pix.z / iter * 255.0), 0, 255);
            pbo[4 * index + 3] = 0;
            pbo[4 * index + 0] = color.x; pbo[4 * index + 1] = color.y; pbo[4 * index + 2] = color.z;
        }
}

// One iteration of the SURVEY 8.0 completion spec driven through the
// reference's headers + GLM (diffuse / mirror / dielectric, stable
// compaction).  live[d] = paths traced at bounce d; returns rays traced.
// `order_out` (optional, traceDepth*N ints) gets the pixelIndex sequence of the
// live prefix after each bounce, -1 terminated.
long long ref_trace_iteration(const Geom *geoms, int ngeoms, const Material *materials,
                              const Camera *camp, int traceDepth, int iter, int compact,
                              glm::vec3 *image, int *live, int *order_out) {
    const Camera &cam = *camp;
    const int N = cam.resolution.x * cam.resolution.y;
    std::vector<PathSegment> paths(N);
    std::vector<ShadeableIntersection> isx(N);
    std::vector<unsigned char> outs(N);
    ref_generate_rays(camp, traceDepth, paths.data());
    int n = N; long long rays = 0;
    for (int depth = 0; depth < traceDepth && n > 0; ++depth) {
        int alive = 0;
        for (int p = 0; p < n; ++p) alive += paths[p].remainingBounces > 0;
        if (alive == 0) break;
        live[depth] = alive; rays += alive;
        std::memset(isx.data(), 0, sizeof(ShadeableIntersection) * N);
        ref_compute_intersections(n, paths.data(), geoms, ngeoms, isx.data(), outs.data());
        for (int p = 0; p < n; ++p) {
            PathSegment &s = paths[p]; ShadeableIntersection &x = isx[p];
            if (s.remainingBounces <= 0) continue;
            if (x.t > 0.0f) {
                const Material &m = materials[x.materialId];
                if (m.emittance > 0.0f) { s.color *= (m.color * m.emittance); s.remainingBounces = 0; }
                else {
                    thrust::default_random_engine rng = mk(iter, s.pixelIndex, depth);
                    thrust::uniform_real_distribution<float> u01(0, 1);
                    glm::vec3 P = getPointOnRay(s.ray, x.t);
                    glm::vec3 I = s.ray.direction;
                    if (m.hasReflective > 0.0f) {
                        s.ray.direction = glm::reflect(I, x.surfaceNormal); s.ray.origin = P;
                        s.color *= m.specular.color;
                    } else if (m.hasRefractive > 0.0f) {
                        glm::vec3 nn = glm::dot(I, x.surfaceNormal) > 0.0f ? -x.surfaceNormal : x.surfaceNormal;
                        float ior = m.indexOfRefraction;
                        float eta = outs[p] ? (1.0f / ior) : ior;
                        float dv = glm::dot(nn, I);
                        float k = 1.0f - eta * eta * (1.0f - dv * dv);
                        bool refl;
                        if (k < 0.0f) refl = true;
                        else {
                            float r0 = (1.0f - ior) / (1.0f + ior); r0 = r0 * r0;
                            float cm = 1.0f - (-dv);
                            float c5 = (((cm * cm) * cm) * cm) * cm;
                            float R = r0 + (1.0f - r0) * c5;
                            float u = u01(rng);
                            refl = u < R;
                            if (!refl) { s.ray.direction = glm::refract(I, nn, eta); s.ray.origin = P + I * 0.0002f; }
                        }
                        if (refl) { s.ray.direction = glm::reflect(I, nn); s.ray.origin = P; }
                        s.color *= m.specular.color;
                    } else {
                        s.ray.direction = calculateRandomDirectionInHemisphere(x.surfaceNormal, rng);
                        s.ray.origin = P; s.color *= m.color;
                    }
                    if (--s.remainingBounces == 0) s.color = glm::vec3(0);
                }
            } else { s.color = glm::vec3(0); s.remainingBounces = 0; }
        }
        if (compact)
            n = (int)(std::stable_partition(paths.begin(), paths.begin() + n,
                      [](const PathSegment &s) { return s.remainingBounces > 0; }) - paths.begin());
        if (order_out) {
            int *o = order_out + (size_t)depth * N;
            for (int p = 0; p < N; ++p) o[p] = p < n ? paths[p].pixelIndex : -1;
        }
    }
    for (int p = 0; p < N; ++p) image[paths[p].pixelIndex] += paths[p].color;
    return rays;
}
#endif  // REF_TU_B

}  // extern "C"
