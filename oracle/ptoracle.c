/*
 * ptoracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See ptoracle.h for scope, parity status and who may call this.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * No FMA contraction, IEEE-754 binary32 arithmetic in exactly the operation
 * order of the reference / GLM 0.9.6.3.  Every
 * function cites the reference file:line it restates; paths are relative to
 * the reference repository root.
 */
#define _GNU_SOURCE
#include "ptoracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

_Static_assert(sizeof(pto_vec3) == 12, "vec3");
_Static_assert(sizeof(pto_mat4) == 64, "mat4");
_Static_assert(sizeof(pto_ray) == 24, "Ray");
_Static_assert(sizeof(pto_geom) == 236, "Geom");
_Static_assert(__builtin_offsetof(pto_geom, transform) == 44, "Geom.transform");
_Static_assert(__builtin_offsetof(pto_geom, inverseTransform) == 108, "Geom.inverseTransform");
_Static_assert(__builtin_offsetof(pto_geom, invTranspose) == 172, "Geom.invTranspose");
_Static_assert(sizeof(pto_material) == 44, "Material");
_Static_assert(__builtin_offsetof(pto_material, hasReflective) == 28, "Material.hasReflective");
_Static_assert(__builtin_offsetof(pto_material, emittance) == 40, "Material.emittance");
_Static_assert(sizeof(pto_camera) == 84, "Camera");
_Static_assert(__builtin_offsetof(pto_camera, pixelLength) == 76, "Camera.pixelLength");
_Static_assert(sizeof(pto_path) == 44, "PathSegment");
_Static_assert(sizeof(pto_isect) == 20, "ShadeableIntersection");
_Static_assert(sizeof(pto_tri) == 36, "tri");

/* src/utilities.h:12-15 */
#define PTO_PI                3.1415926535897932384626422832795028841971f
#define PTO_TWO_PI            6.2831853071795864769252867665590057683943f
#define PTO_SQRT_OF_ONE_THIRD 0.5773502691896257645091487805019574556476f

/* ------------------------------------------------------------------------ */
/* GLM 0.9.6.3 vector algebra, in GLM's operation order                      */
/* ------------------------------------------------------------------------ */
static inline pto_vec3 v3(float x, float y, float z) { pto_vec3 r = {x, y, z}; return r; }
static inline pto_vec4 v4(pto_vec3 v, float w) { pto_vec4 r = {v.x, v.y, v.z, w}; return r; }
static inline pto_vec3 add3(pto_vec3 a, pto_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline pto_vec3 sub3(pto_vec3 a, pto_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline pto_vec3 mul3(pto_vec3 a, pto_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline pto_vec3 muls(pto_vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline pto_vec3 neg3(pto_vec3 a) { return v3(-a.x, -a.y, -a.z); }
/* glm/detail/func_geometric.inl:64-72: tmp = x*y; tmp.x + tmp.y + tmp.z */
static inline float dot3(pto_vec3 a, pto_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* func_geometric.inl:94-100 */
static inline float length3(pto_vec3 a) { return sqrtf(dot3(a, a)); }
/* func_geometric.inl:153-159 + func_exponential.inl:149-153: x * (1 / sqrt(dot)) */
static inline pto_vec3 normalize3(pto_vec3 a) { return muls(a, 1.0f / sqrtf(dot3(a, a))); }
/* func_geometric.inl:133-142 */
static inline pto_vec3 cross3(pto_vec3 x, pto_vec3 y) {
    return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm/detail/func_common.inl:409-414, 430-435 */
static inline float glm_min(float x, float y) { return x < y ? x : y; }
static inline float glm_max(float x, float y) { return x > y ? x : y; }
/* std::min / std::max, which the unqualified min/max of intersections.h:128,131
 * bind to in a host compilation */
static inline float std_min(float a, float b) { return (b < a) ? b : a; }
static inline float std_max(float a, float b) { return (a < b) ? b : a; }

/* func_geometric.inl:175-179: I - N * dot(N, I) * 2 */
pto_vec3 pto_reflect(pto_vec3 I, pto_vec3 N) {
    return sub3(I, muls(muls(N, dot3(N, I)), 2.0f));
}

/* ------------------------------------------------------------------------ */
/* integer hash + thrust::minstd_rand + uniform_real_distribution<float>     */
/* ------------------------------------------------------------------------ */

/* src/intersections.h:12-20 */
uint32_t pto_utilhash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

/* thrust/random/detail/linear_congruential_engine.inl:43-50 (rocThrust 2.8.5;
 * minstd_rand: a=48271, c=0, m=2^31-1, linear_congruential_engine.h:274) */
uint32_t pto_lcg_seed(uint32_t s) {
    const uint32_t m = 2147483647u;
    uint32_t r = s % m;
    return r == 0 ? 1u : r;
}

/* linear_congruential_engine.inl:54-60 + detail/mod.h:32-51 (Schrage) --
 * identical to the 64-bit product a*x mod m */
uint32_t pto_lcg_next(uint32_t *state) {
    uint64_t p = (uint64_t)48271u * (uint64_t)(*state);
    *state = (uint32_t)(p % 2147483647ull);
    return *state;
}

/* thrust/random/detail/uniform_real_distribution.inl:70-79 with (a,b)=(0,1):
 * float(x - min) / (1.0f + float(max - min)) = float(x-1) / 2147483648.0f;
 * then * (1-0) + 0, both exact. */
float pto_u01(uint32_t *state) {
    uint32_t x = pto_lcg_next(state);
    float result = (float)(x - 1u);
    result /= (1.0f + (float)(2147483646u - 1u));
    return (result * (1.0f - 0.0f)) + 0.0f;
}

/* src/pathtrace.cu:41-45.  `int h` -> engine(result_type=uint32) */
uint32_t pto_make_seeded_engine(int iter, int index, int depth) {
    uint32_t k = (1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter;
    uint32_t h = pto_utilhash(k) ^ pto_utilhash((uint32_t)index);
    return pto_lcg_seed(h);
}

/* ------------------------------------------------------------------------ */
/* shared sin/cos (build-defined; DESIGN.md "shared trig")                   */
/* ------------------------------------------------------------------------ */
/* sin/cos of a float argument in binary32 with explicit fused multiply-adds -- fmaf() here, v_fma_f32 in the HIP
 * kernels (csrc/pt_device.hpp: sincos_shared), the same sequence operation for operation, so the result is
 * bit-identical on host and device (checked on EVERY float of [0, 2 pi]: tests/test_gpu_pins.py).
 *   k  = x * 2/pi rounded to nearest (1.5 * 2^23 trick);   r + rl = x - k pi/2 in two floats (three-constant Cody-Waite,
 *        every step one fma: exact to 2^-36);
 *   sin(r) = r + (r z (S1 + z (S2 + z (S3 + z S4))) + rl (1 - z/2)),          z = r r, S = minimax fit on [-pi/4, pi/4]
 *   cos(r) = w + (z z (C1 + z (C2 + z C3)) + ((1 - w) - z/2 - ze/2 - rl sin r)),  w = fl(1 - z/2), ze = r r - z exactly
 * and the quadrant k mod 4 picks and signs the two.  Within 1 ulp of the correctly rounded value on every float of
 * [0, 2 pi] and correctly rounded on 98.6 % of a uniform grid over it (oracle test_shared_sincos_accuracy) -- what the
 * libms the reference can bind to deliver (CUDA's sinf / cosf: 1 ulp; glibc's: < 1 ulp).  Rounds 1-3 evaluated fdlibm's
 * binary64 polynomials instead: ~40 binary64 instructions at twice the issue cost inside the kernels' scatter block
 * (9 % of k_bounce's issue cycles).  Valid for |x| < 2^22 (the path tracer needs [0, 2 pi]). */
void pto_sincos(float x, float *s, float *c) {
    static const float TWO_OVER_PI = 0.636619772367581343f;      /* 0x3f22f983 */
    static const float MAGIC = 12582912.0f;                       /* 1.5 * 2^23 */
    static const float P1 = 1.57079625129699707031f;             /* 0x3fc90fda: pi/2 truncated to 24 bits */
    static const float P2 = 7.54978941586159635335e-08f;          /* 0x33a22168: fl(pi/2 - P1) */
    static const float P3 = 5.39030285815811905290e-15f;          /* 0x27c234c4: fl(pi/2 - P1 - P2) */
    static const float S1 = -1.66666671633720398e-01f, S2 = 8.33333190530538559e-03f,      /* 0xbe2aaaab 0x3c088887 */
                       S3 = -1.98401714442297816e-04f, S4 = 2.72681563728838228e-06f;      /* 0xb9500a0e 0x3636fe56 */
    static const float C1 = 4.16666530072689056e-02f, C2 = -1.38876168057322502e-03f,      /* 0x3d2aaaa7 0xbab6071c */
                       C3 = 2.44678121816832572e-05f;                                       /* 0x37cd403a */
    float kf = fmaf(x, TWO_OVER_PI, MAGIC) - MAGIC;   /* nearest integer, ties-to-even */
    float r1 = fmaf(-kf, P1, x);
    float r = fmaf(-kf, P2, r1);
    float rl = fmaf(-kf, P3, fmaf(-kf, P2, r1 - r));
    float z = r * r;
    float ze = fmaf(r, r, -z);
    float ps = fmaf(z, fmaf(z, fmaf(z, S4, S3), S2), S1);
    float pc = fmaf(z, fmaf(z, C3, C2), C1);
    float hz = 0.5f * z;
    float w = 1.0f - hz;
    float e = (1.0f - w) - hz;
    e = fmaf(-0.5f, ze, e);
    float u = (r * z) * ps;
    float sn = r + fmaf(rl, w, u);
    float cs = w + fmaf(z * z, pc, fmaf(-rl, sn, e));
    int q = (int)kf & 3;
    float so, co;
    switch (q) {
    case 0: so = sn;  co = cs;  break;
    case 1: so = cs;  co = -sn; break;
    case 2: so = -sn; co = -cs; break;
    default: so = -cs; co = sn; break;
    }
    *s = so;
    *c = co;
}

/* The two checksums libptmi355.so's pt_probe_sincos forms on the device (include/ptmi355.h), here on the CPU: over the
 * n consecutive binary32 values from bit pattern first_bits on, sum[0] = sum of bits(sin) * (2k + 1), sum[1] = the
 * same for cos (mod 2^64) -- every argument calculateRandomDirectionInHemisphere can form (u01 * TWO_PI, [0, 2 pi])
 * is compared without storing a value. */
void pto_sincos_sums(uint32_t first_bits, uint32_t n, uint64_t sum[2]) {
    uint64_t as = 0, ac = 0;
    for (uint32_t k = 0; k < n; ++k) {
        union { uint32_t u; float f; } x, sv, cv;
        x.u = first_bits + k;
        pto_sincos(x.f, &sv.f, &cv.f);
        as += (uint64_t)sv.u * (2ull * k + 1ull);
        ac += (uint64_t)cv.u * (2ull * k + 1ull);
    }
    sum[0] = as; sum[1] = ac;
}

/* ------------------------------------------------------------------------ */
/* geometry helpers                                                          */
/* ------------------------------------------------------------------------ */

/* src/intersections.h:27-29 */
pto_vec3 pto_get_point_on_ray(pto_ray r, float t) {
    return add3(r.origin, muls(normalize3(r.direction), (t - .0001f)));
}

/* src/intersections.h:34-36; glm/detail/type_mat4x4.inl:617-628:
 * (m[0]*v0 + m[1]*v1) + (m[2]*v2 + m[3]*v3) */
pto_vec3 pto_multiply_mv(const pto_mat4 *m, pto_vec4 v) {
    float out[3];
    for (int r = 0; r < 3; ++r) {
        float mul0 = m->m[0][r] * v.x;
        float mul1 = m->m[1][r] * v.y;
        float add0 = mul0 + mul1;
        float mul2 = m->m[2][r] * v.z;
        float mul3 = m->m[3][r] * v.w;
        float add1 = mul2 + mul3;
        out[r] = add0 + add1;
    }
    return v3(out[0], out[1], out[2]);
}

/* src/intersections.h:48-90 */
float pto_box_test(const pto_geom *box, pto_ray r, pto_vec3 *point, pto_vec3 *normal,
                   int *outside) {
    pto_ray q;
    q.origin = pto_multiply_mv(&box->inverseTransform, v4(r.origin, 1.0f));
    q.direction = normalize3(pto_multiply_mv(&box->inverseTransform, v4(r.direction, 0.0f)));

    float tmin = -1e38f;
    float tmax = 1e38f;
    pto_vec3 tmin_n = v3(0, 0, 0);   /* glm::vec3 default ctor zero-inits (type_vec3.inl:39-43) */
    pto_vec3 tmax_n = v3(0, 0, 0);
    const float qo[3] = {q.origin.x, q.origin.y, q.origin.z};
    const float qd[3] = {q.direction.x, q.direction.y, q.direction.z};
    for (int xyz = 0; xyz < 3; ++xyz) {
        float qdxyz = qd[xyz];
        /* no zero-guard: intersections.h:60 is commented out */
        float t1 = (-0.5f - qo[xyz]) / qdxyz;
        float t2 = (+0.5f - qo[xyz]) / qdxyz;
        float ta = glm_min(t1, t2);
        float tb = glm_max(t1, t2);
        float n[3] = {0, 0, 0};
        n[xyz] = t2 < t1 ? +1 : -1;
        if (ta > 0 && ta > tmin) {
            tmin = ta;
            tmin_n = v3(n[0], n[1], n[2]);
        }
        if (tb < tmax) {
            tmax = tb;
            tmax_n = v3(n[0], n[1], n[2]);
        }
    }

    if (tmax >= tmin && tmax > 0) {
        *outside = 1;
        if (tmin <= 0) {
            tmin = tmax;
            tmin_n = tmax_n;
            *outside = 0;
        }
        *point = pto_multiply_mv(&box->transform, v4(pto_get_point_on_ray(q, tmin), 1.0f));
        *normal = normalize3(pto_multiply_mv(&box->transform, v4(tmin_n, 0.0f)));
        return length3(sub3(r.origin, *point));
    }
    return -1;
}

/* src/intersections.h:102-144 */
float pto_sphere_test(const pto_geom *sphere, pto_ray r, pto_vec3 *point, pto_vec3 *normal,
                      int *outside) {
    float radius = .5;

    pto_vec3 ro = pto_multiply_mv(&sphere->inverseTransform, v4(r.origin, 1.0f));
    pto_vec3 rd = normalize3(pto_multiply_mv(&sphere->inverseTransform, v4(r.direction, 0.0f)));

    pto_ray rt;
    rt.origin = ro;
    rt.direction = rd;

    float vDotDirection = dot3(rt.origin, rt.direction);
    /* powf(.5f, 2) == 0.25f exactly */
    float radicand = vDotDirection * vDotDirection - (dot3(rt.origin, rt.origin) - (radius * radius));
    if (radicand < 0) {
        return -1;
    }

    float squareRoot = sqrtf(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;

    float t = 0;
    if (t1 < 0 && t2 < 0) {
        return -1;
    } else if (t1 > 0 && t2 > 0) {
        t = std_min(t1, t2);
        *outside = 1;
    } else {
        t = std_max(t1, t2);
        *outside = 0;
    }

    pto_vec3 objspaceIntersection = pto_get_point_on_ray(rt, t);

    *point = pto_multiply_mv(&sphere->transform, v4(objspaceIntersection, 1.f));
    *normal = normalize3(pto_multiply_mv(&sphere->invTranspose, v4(objspaceIntersection, 0.f)));
    if (!*outside) {
        *normal = neg3(*normal);
    }

    return length3(sub3(r.origin, *point));
}

/* external/include/glm/gtx/intersect.inl:37-74 (glm::intersectRayTriangle) */
int pto_ray_triangle(pto_vec3 orig, pto_vec3 dir, pto_vec3 v0, pto_vec3 v1, pto_vec3 v2,
                     pto_vec3 *bary) {
    pto_vec3 e1 = sub3(v1, v0);
    pto_vec3 e2 = sub3(v2, v0);
    pto_vec3 p = cross3(dir, e2);
    float a = dot3(e1, p);
    float Epsilon = FLT_EPSILON;
    if (a < Epsilon) return 0;
    float f = 1.0f / a;
    pto_vec3 s = sub3(orig, v0);
    bary->x = f * dot3(s, p);
    if (bary->x < 0.0f) return 0;
    if (bary->x > 1.0f) return 0;
    pto_vec3 q = cross3(s, e1);
    bary->y = f * dot3(dir, q);
    if (bary->y < 0.0f) return 0;
    if (bary->y + bary->x > 1.0f) return 0;
    bary->z = f * dot3(e2, q);
    return bary->z >= 0.0f;
}

/* Completion spec 8.0 "Triangles": naive loop over the mesh's world-space
 * triangles; nearest = strictly smallest parametric bary.z (first triangle
 * wins ties); the mesh reports, like the cube/sphere tests, the world point
 * P = o + d*tz, the geometric normal normalize(cross(e1,e2)) and
 * t = length(o - P).  Back-face culled (a < eps -> miss), so `outside` is
 * always true.
 *
 * Hit-point test (spec 8.0 "Triangles", second paragraph).  In single precision
 * glm::intersectRayTriangle accepts rays that run almost inside a large
 * triangle's plane with barycentrics that are rounding noise: the point
 * o + d*tz it reports can lie metres away from the triangle.  The spec
 * therefore counts a triangle hit only when that point -- evaluated as
 * fl(o_k + fl(d_k * tz)) -- lies inside the triangle's bounding box
 * [min, max](v0_k, fl(v0_k + e1_k), fl(v0_k + e2_k)), e = the float edge
 * vectors of the test, widened by the mesh's pad = 2^-14 * max(1, largest
 * finite |vertex coordinate| of the mesh).  Every geometrically meaningful
 * hit passes; what it removes is garbage, and it is what makes a spatial
 * hierarchy over the triangles provably equal to this loop (pt_bvh.hpp). */
float pto_mesh_pad(const pto_tri *tris, int first, int count) {
    float amax = 0.0f;
    for (int i = first; i < first + count; ++i) {
        const float *c = (const float *)&tris[i];
        for (int k = 0; k < 9; ++k) {
            float m = fabsf(c[k]);
            if (m <= FLT_MAX && m > amax) amax = m;
        }
    }
    return ldexpf(amax > 1.0f ? amax : 1.0f, -14);
}

int pto_tri_point_ok(pto_vec3 o, pto_vec3 d, float tz, const pto_tri *t, float pad) {
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    const float a[3] = {t->v0.x, t->v0.y, t->v0.z};
    const float b[3] = {t->v1.x, t->v1.y, t->v1.z}, c[3] = {t->v2.x, t->v2.y, t->v2.z};
    for (int k = 0; k < 3; ++k) {
        float p = oo[k] + dd[k] * tz;
        float e1 = b[k] - a[k], e2 = c[k] - a[k];
        float x1 = a[k] + e1, x2 = a[k] + e2;
        float lo = fminf(a[k], fminf(x1, x2)), hi = fmaxf(a[k], fmaxf(x1, x2));
        if (!(p >= lo - pad && p <= hi + pad)) return 0;
    }
    return 1;
}

int pto_mesh_winner(const pto_tri *tris, int first, int count, pto_ray r, float pad, float *tz) {
    float best = FLT_MAX;
    int hit = -1;
    for (int i = first; i < first + count; ++i) {
        pto_vec3 b;
        if (pto_ray_triangle(r.origin, r.direction, tris[i].v0, tris[i].v1, tris[i].v2, &b)) {
            if (b.z > 0.0f && best > b.z && pto_tri_point_ok(r.origin, r.direction, b.z, &tris[i], pad)) {
                best = b.z;
                hit = i;
            }
        }
    }
    *tz = best;
    return hit;
}

void pto_mesh_winners(const pto_tri *tris, int first, int count, const pto_path *paths, int n,
                      int32_t *index, float *tz) {
    const float pad = pto_mesh_pad(tris, first, count);
    for (int k = 0; k < n; ++k) index[k] = pto_mesh_winner(tris, first, count, paths[k].ray, pad, &tz[k]);
}

/* every (ray, triangle) pair the spec counts, not only the winner: accepted[k * count + i] (tests of the kernels' cull stages) */
void pto_mesh_accepted(const pto_tri *tris, int first, int count, const pto_path *paths, int n, uint8_t *accepted) {
    const float pad = pto_mesh_pad(tris, first, count);
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < count; ++i) {
            pto_vec3 b;
            const pto_tri *t = &tris[first + i];
            accepted[(size_t)k * count + i] =
                pto_ray_triangle(paths[k].ray.origin, paths[k].ray.direction, t->v0, t->v1, t->v2, &b) && b.z > 0.0f &&
                pto_tri_point_ok(paths[k].ray.origin, paths[k].ray.direction, b.z, t, pad);
        }
}

float pto_mesh_test(const pto_tri *tris, int first, int count, pto_ray r, float pad, pto_vec3 *point,
                    pto_vec3 *normal, int *outside) {
    float best;
    int hit = pto_mesh_winner(tris, first, count, r, pad, &best);
    if (hit < 0) return -1;
    *outside = 1;
    *point = add3(r.origin, muls(r.direction, best));
    *normal = normalize3(cross3(sub3(tris[hit].v1, tris[hit].v0), sub3(tris[hit].v2, tris[hit].v0)));
    return length3(sub3(r.origin, *point));
}

/* src/interactions.h:10-42 */
pto_vec3 pto_hemisphere(pto_vec3 normal, uint32_t *rng, int trig) {
    float up = sqrtf(pto_u01(rng));       /* cos(theta) */
    float over = sqrtf(1 - up * up);      /* sin(theta) */
    float around = pto_u01(rng) * PTO_TWO_PI;

    pto_vec3 directionNotNormal;
    if (fabsf(normal.x) < PTO_SQRT_OF_ONE_THIRD) {
        directionNotNormal = v3(1, 0, 0);
    } else if (fabsf(normal.y) < PTO_SQRT_OF_ONE_THIRD) {
        directionNotNormal = v3(0, 1, 0);
    } else {
        directionNotNormal = v3(0, 0, 1);
    }

    pto_vec3 perpendicularDirection1 = normalize3(cross3(normal, directionNotNormal));
    pto_vec3 perpendicularDirection2 = normalize3(cross3(normal, perpendicularDirection1));

    float ca, sa;
    if (trig == PTO_TRIG_LIBM) {
        ca = cosf(around);
        sa = sinf(around);
    } else {
        pto_sincos(around, &sa, &ca);
    }
    /* up * normal + cos(around) * over * p1 + sin(around) * over * p2
     * = ((up*normal) + ((cos*over)*p1)) + ((sin*over)*p2) */
    pto_vec3 a = muls(normal, up);
    pto_vec3 b = muls(perpendicularDirection1, ca * over);
    pto_vec3 c = muls(perpendicularDirection2, sa * over);
    return add3(add3(a, b), c);
}

/* ------------------------------------------------------------------------ */
/* kernels restated as loops                                                 */
/* ------------------------------------------------------------------------ */

/* src/pathtrace.cu:122-143, with the two camera extensions the reference leaves as a TODO at :134
 * ("implement antialiasing by jittering the ray"; INSTRUCTION.md:110-113) -- completion spec 8.0/DESIGN 3:
 *   engine = makeSeededRandomEngine(iter, index, traceDepth)      a depth slot no bounce uses
 *   jitter : x += u01 - 0.5, y += u01 - 0.5                       (drawn in this order, before the lens)
 *   lens   : focus = position + dir * (focalDistance / dot(dir, view));
 *            r = lensRadius * sqrt(u01), theta = u01 * 2pi;
 *            origin = position + right * (r cos theta) + up * (r sin theta);
 *            direction = normalize(focus - origin)
 * With neither enabled no random number is drawn and the result is the reference's, bit for bit. */
static void generate_ray(const pto_camera *cam, int traceDepth, int iter, int aa, float lensRadius,
                         float focalDistance, int trig, int x, int y, pto_path *segment) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    int index = x + (y * W);
    uint32_t rng = 0;
    if (aa || lensRadius > 0.0f) rng = pto_make_seeded_engine(iter, index, traceDepth);
    float fx = (float)x, fy = (float)y;
    if (aa) {
        fx = fx + (pto_u01(&rng) - 0.5f);
        fy = fy + (pto_u01(&rng) - 0.5f);
    }
    segment->ray.origin = cam->position;
    segment->color = v3(1.0f, 1.0f, 1.0f);
    /* view - right*plx*(x - W*0.5) - up*ply*(y - H*0.5), left-assoc */
    pto_vec3 a = muls(muls(cam->right, cam->pixelLength[0]), (fx - (float)W * 0.5f));
    pto_vec3 b = muls(muls(cam->up, cam->pixelLength[1]), (fy - (float)H * 0.5f));
    segment->ray.direction = normalize3(sub3(sub3(cam->view, a), b));
    if (lensRadius > 0.0f) {
        pto_vec3 dir = segment->ray.direction;
        float ft = focalDistance / dot3(dir, cam->view);
        pto_vec3 focus = add3(cam->position, muls(dir, ft));
        float r = lensRadius * sqrtf(pto_u01(&rng));
        float theta = pto_u01(&rng) * PTO_TWO_PI;
        float ca, sa;
        if (trig == PTO_TRIG_LIBM) {
            ca = cosf(theta);
            sa = sinf(theta);
        } else {
            pto_sincos(theta, &sa, &ca);
        }
        pto_vec3 origin = add3(add3(cam->position, muls(cam->right, r * ca)), muls(cam->up, r * sa));
        segment->ray.origin = origin;
        segment->ray.direction = normalize3(sub3(focus, origin));
    }
    segment->pixelIndex = index;
    segment->remainingBounces = traceDepth;
}

void pto_generate_rays(const pto_camera *cam, int traceDepth, pto_path *paths) {
    const int W = cam->resolution[0], H = cam->resolution[1];
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) generate_ray(cam, traceDepth, 0, 0, 0.0f, 0.0f, PTO_TRIG_SHARED, x, y, &paths[x + y * W]);
}

void pto_generate_rays_ex(const pto_scene *sc, int iter, pto_path *paths) {
    const int W = sc->camera.resolution[0], H = sc->camera.resolution[1];
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            generate_ray(&sc->camera, sc->traceDepth, iter, (sc->flags & PTO_F_AA) != 0, sc->lensRadius,
                         sc->focalDistance, sc->trig, x, y, &paths[x + y * W]);
}

/* src/pathtrace.cu:149-213.  isects[] must be pre-zeroed by the caller exactly
 * as pathtrace.cu:343 does (a miss only writes t). `outside_or_null` records the
 * winning test's `outside` flag, which the completion spec's dielectric needs. */
static void compute_intersections_range(int begin, int end, const pto_path *paths,
                                        const pto_geom *geoms, int ngeoms, const pto_tri *tris,
                                        const pto_mesh *meshes, int nmeshes, pto_isect *isects,
                                        uint8_t *outside_out) {
    float *pads = (float *)malloc(sizeof(float) * (size_t)(nmeshes > 0 ? nmeshes : 1));
    for (int k = 0; k < nmeshes; ++k) pads[k] = pto_mesh_pad(tris, meshes[k].first_tri, meshes[k].tri_count);
    for (int path_index = begin; path_index < end; ++path_index) {
        pto_path pathSegment = paths[path_index];
        float t = 0.0f;
        pto_vec3 normal = v3(0, 0, 0);
        float t_min = FLT_MAX;
        int hit_geom_index = -1;
        int outside = 1;
        int hit_outside = 1;
        pto_vec3 tmp_intersect = v3(0, 0, 0);
        pto_vec3 tmp_normal = v3(0, 0, 0);

        for (int i = 0; i < ngeoms; i++) {
            const pto_geom *geom = &geoms[i];
            if (geom->type == PTO_CUBE) {
                t = pto_box_test(geom, pathSegment.ray, &tmp_intersect, &tmp_normal, &outside);
            } else if (geom->type == PTO_SPHERE) {
                t = pto_sphere_test(geom, pathSegment.ray, &tmp_intersect, &tmp_normal, &outside);
            } else if (geom->type == PTO_TRIMESH) {
                t = -1;
                for (int k = 0; k < nmeshes; ++k) {
                    if (meshes[k].geom_index == i) {
                        t = pto_mesh_test(tris, meshes[k].first_tri, meshes[k].tri_count,
                                          pathSegment.ray, pads[k], &tmp_intersect, &tmp_normal, &outside);
                        break;
                    }
                }
            }
            if (t > 0.0f && t_min > t) {
                t_min = t;
                hit_geom_index = i;
                normal = tmp_normal;
                hit_outside = outside;
            }
        }

        if (hit_geom_index == -1) {
            isects[path_index].t = -1.0f;
        } else {
            isects[path_index].t = t_min;
            isects[path_index].materialId = geoms[hit_geom_index].materialid;
            isects[path_index].surfaceNormal = normal;
        }
        if (outside_out) outside_out[path_index] = (uint8_t)hit_outside;
    }
    free(pads);
}

void pto_compute_intersections(int n, const pto_path *paths, const pto_geom *geoms, int ngeoms,
                               const pto_tri *tris, const pto_mesh *meshes, int nmeshes,
                               pto_isect *isects, uint8_t *outside_or_null) {
    compute_intersections_range(0, n, paths, geoms, ngeoms, tris, meshes, nmeshes, isects,
                                outside_or_null);
}

/* src/pathtrace.cu:224-266 */
void pto_shade_fake(int iter, int n, const pto_isect *isects, pto_path *paths,
                    const pto_material *materials) {
    for (int idx = 0; idx < n; ++idx) {
        pto_isect intersection = isects[idx];
        if (intersection.t > 0.0f) {
            uint32_t rng = pto_make_seeded_engine(iter, idx, 0);
            pto_material material = materials[intersection.materialId];
            pto_vec3 materialColor = material.color;
            if (material.emittance > 0.0f) {
                paths[idx].color = mul3(paths[idx].color, muls(materialColor, material.emittance));
            } else {
                float lightTerm = dot3(intersection.surfaceNormal, v3(0.0f, 1.0f, 0.0f));
                pto_vec3 a = muls(muls(materialColor, lightTerm), 0.3f);
                /* (1.0f - t*0.02f) * materialColor : scalar * vec */
                pto_vec3 b = muls(muls(materialColor, (1.0f - intersection.t * 0.02f)), 0.7f);
                paths[idx].color = mul3(paths[idx].color, add3(a, b));
                paths[idx].color = muls(paths[idx].color, pto_u01(&rng));
            }
        } else {
            paths[idx].color = v3(0.0f, 0.0f, 0.0f);
        }
    }
}

/* src/pathtrace.cu:269-278 */
void pto_final_gather(int n, pto_vec3 *image, const pto_path *paths) {
    for (int index = 0; index < n; ++index) {
        pto_path iterationPath = paths[index];
        image[iterationPath.pixelIndex] = add3(image[iterationPath.pixelIndex], iterationPath.color);
    }
}

/* src/pathtrace.cu:48-68: (int)(pix / iter * 255.0) -- float divide, then a
 * DOUBLE multiply, truncation, clamp to [0,255]; w = 0. */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
void pto_send_image_to_pbo(uint8_t *pbo, int w, int h, int iter, const pto_vec3 *image) {
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int index = x + (y * w);
            pto_vec3 pix = image[index];
            double cx = (double)(pix.x / (float)iter) * 255.0;
            double cy = (double)(pix.y / (float)iter) * 255.0;
            double cz = (double)(pix.z / (float)iter) * 255.0;
            /* out-of-range double->int is UB in C; the reference relies on the
             * x86/CUDA saturating behaviour only for huge values.  Clamp in
             * double first: identical for every in-range value. */
            int ix = cx >= 2147483647.0 ? 2147483647 : (cx <= -2147483648.0 ? (-2147483647 - 1) : (int)cx);
            int iy = cy >= 2147483647.0 ? 2147483647 : (cy <= -2147483648.0 ? (-2147483647 - 1) : (int)cy);
            int iz = cz >= 2147483647.0 ? 2147483647 : (cz <= -2147483648.0 ? (-2147483647 - 1) : (int)cz);
            if (cx != cx) ix = 0;
            if (cy != cy) iy = 0;
            if (cz != cz) iz = 0;
            pbo[4 * index + 3] = 0;
            pbo[4 * index + 0] = (uint8_t)clampi(ix, 0, 255);
            pbo[4 * index + 1] = (uint8_t)clampi(iy, 0, 255);
            pbo[4 * index + 2] = (uint8_t)clampi(iz, 0, 255);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* completion spec                                           */
/* ------------------------------------------------------------------------ */

/* The body the reference leaves empty at src/interactions.h:69-79.
 * Precedence: mirror (hasReflective > 0), dielectric (hasRefractive > 0),
 * else diffuse.  `intersect` is getPointOnRay(ray, t).  Does not touch
 * remainingBounces (the caller does, as in pathtrace.cu's shader). */
void pto_scatter_ray(pto_path *path, pto_vec3 intersect, pto_vec3 normal, int outside,
                     const pto_material *m, uint32_t *rng, int trig) {
    pto_vec3 I = path->ray.direction;
    if (m->hasReflective > 0.0f) {
        path->ray.direction = pto_reflect(I, normal);
        path->ray.origin = intersect;
        path->color = mul3(path->color, m->specular.color);
    } else if (m->hasRefractive > 0.0f) {
        /* face-forward normal: oppose the incoming direction (the sphere test
         * already flips it when inside, intersections.h:139-141; the cube test
         * returns the outward exit-face normal, intersections.h:80-86) */
        float d0 = dot3(I, normal);
        pto_vec3 nn = d0 > 0.0f ? neg3(normal) : normal;
        float ior = m->indexOfRefraction;
        float eta = outside ? (1.0f / ior) : ior;
        /* glm::refract (func_geometric.inl:192-200), k tested before the sqrt */
        float dotValue = dot3(nn, I);
        float k = 1.0f - eta * eta * (1.0f - dotValue * dotValue);
        int do_reflect;
        if (k < 0.0f) {
            do_reflect = 1;                 /* total internal reflection */
        } else {
            /* Schlick: R0 + (1-R0)(1-cos)^5 with cos = -dot(nn, I) */
            float r0 = (1.0f - ior) / (1.0f + ior);
            r0 = r0 * r0;
            float cm = 1.0f - (-dotValue);
            float c5 = (((cm * cm) * cm) * cm) * cm;
            float R = r0 + (1.0f - r0) * c5;
            float u = pto_u01(rng);         /* Fresnel choice draw comes first */
            do_reflect = u < R;
            if (!do_reflect) {
                /* eta * I - (eta * dotValue + sqrt(k)) * N */
                pto_vec3 a = muls(I, eta);
                pto_vec3 b = muls(nn, (eta * dotValue + sqrtf(k)));
                path->ray.direction = sub3(a, b);
                /* step through the surface: P is 1e-4 short of it */
                path->ray.origin = add3(intersect, muls(I, 0.0002f));
            }
        }
        if (do_reflect) {
            path->ray.direction = pto_reflect(I, nn);
            path->ray.origin = intersect;
        }
        path->color = mul3(path->color, m->specular.color);
    } else {
        path->ray.direction = pto_hemisphere(normal, rng, trig);
        path->ray.origin = intersect;
        path->color = mul3(path->color, m->color);
    }
}

static void shade_scatter_range(int iter, int depth, int begin, int end, const pto_isect *isects,
                                const uint8_t *outside, pto_path *paths,
                                const pto_material *materials, int trig) {
    for (int idx = begin; idx < end; ++idx) {
        pto_path *seg = &paths[idx];
        if (seg->remainingBounces <= 0) continue;          /* compaction-off mode */
        pto_isect x = isects[idx];
        if (x.t > 0.0f) {
            const pto_material *material = &materials[x.materialId];
            if (material->emittance > 0.0f) {
                /* pathtrace.cu:247-249, then terminate */
                seg->color = mul3(seg->color, muls(material->color, material->emittance));
                seg->remainingBounces = 0;
            } else {
                /* RNG key (iter, pixelIndex, depth): 8.0 "RNG key" */
                uint32_t rng = pto_make_seeded_engine(iter, seg->pixelIndex, depth);
                pto_vec3 P = pto_get_point_on_ray(seg->ray, x.t);
                pto_scatter_ray(seg, P, x.surfaceNormal, outside ? outside[idx] : 1, material,
                                &rng, trig);
                seg->remainingBounces -= 1;
                if (seg->remainingBounces == 0) seg->color = v3(0.0f, 0.0f, 0.0f);
            }
        } else {
            /* pathtrace.cu:262-264, then terminate */
            seg->color = v3(0.0f, 0.0f, 0.0f);
            seg->remainingBounces = 0;
        }
    }
}

void pto_shade_scatter(int iter, int depth, int n, const pto_isect *isects, const uint8_t *outside,
                       pto_path *paths, const pto_material *materials, int trig) {
    shade_scatter_range(iter, depth, 0, n, isects, outside, paths, materials, trig);
}

/* stable partition by remainingBounces > 0 over [0,n); dead paths keep their
 * relative order in the tail [n_live, n). */
int pto_compact(int n, pto_path *paths, pto_path *scratch) {
    int live = 0, dead = 0;
    for (int i = 0; i < n; ++i)
        if (paths[i].remainingBounces > 0) live++;
    int li = 0;
    dead = live;
    for (int i = 0; i < n; ++i) {
        if (paths[i].remainingBounces > 0) scratch[li++] = paths[i];
        else scratch[dead++] = paths[i];
    }
    memcpy(paths, scratch, (size_t)n * sizeof(pto_path));
    return live;
}

/* stable sort of (path, isect, outside) by materialId ascending; misses
 * (t <= 0) carry key = INT_MAX-ish so they sort last. Insertion via counting
 * of distinct keys is overkill here: use a stable merge on an index array. */
typedef struct { int key; int idx; } sort_item;
static int sort_cmp(const void *a, const void *b) {
    const sort_item *x = (const sort_item *)a, *y = (const sort_item *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);   /* stability */
}
void pto_sort_by_material(int n, pto_path *paths, pto_isect *isects, uint8_t *outside,
                          void *scratch) {
    sort_item *items = (sort_item *)malloc((size_t)n * sizeof(sort_item));
    for (int i = 0; i < n; ++i) {
        items[i].key = isects[i].t > 0.0f ? isects[i].materialId : 0x7fffffff;
        items[i].idx = i;
    }
    qsort(items, (size_t)n, sizeof(sort_item), sort_cmp);
    pto_path *ps = (pto_path *)scratch;
    pto_isect *is = (pto_isect *)(ps + n);
    uint8_t *os = (uint8_t *)(is + n);
    for (int i = 0; i < n; ++i) {
        ps[i] = paths[items[i].idx];
        is[i] = isects[items[i].idx];
        if (outside) os[i] = outside[items[i].idx];
    }
    memcpy(paths, ps, (size_t)n * sizeof(pto_path));
    memcpy(isects, is, (size_t)n * sizeof(pto_isect));
    if (outside) memcpy(outside, os, (size_t)n);
    free(items);
}

/* ---- iteration-parallel driver: the CPU baseline of bench.py ------------------------------------------
 * Iterations are independent (the RNG is keyed by the iteration number), so the natural way to use every host
 * core is one whole iteration per thread -- what the GPU's batch mode does with samples.  Each iteration goes to
 * its own image and the images are added in iteration order, so the sum equals `count` sequential calls of
 * pto_trace_iteration bit for bit. */
typedef struct {
    const pto_scene *sc;
    int iter0, count, N;
    pto_vec3 *images;              /* count * N */
    int next;                      /* next iteration to hand out */
    int64_t rays;
    pthread_mutex_t lock;
} par_job;

static void *par_worker(void *arg) {
    par_job *j = (par_job *)arg;
    pto_path *paths = (pto_path *)malloc((size_t)j->N * sizeof(pto_path));
    pto_isect *isects = (pto_isect *)malloc((size_t)j->N * sizeof(pto_isect));
    for (;;) {
        pthread_mutex_lock(&j->lock);
        const int k = j->next < j->count ? j->next++ : -1;
        pthread_mutex_unlock(&j->lock);
        if (k < 0) break;
        pto_stats st;
        pto_trace_iteration(j->sc, j->iter0 + k, j->images + (size_t)k * j->N, paths, isects, &st, NULL, NULL);
        pthread_mutex_lock(&j->lock);
        j->rays += st.rays;
        pthread_mutex_unlock(&j->lock);
    }
    free(paths);
    free(isects);
    return NULL;
}

int64_t pto_trace_iterations_parallel(const pto_scene *sc, int iter0, int count, pto_vec3 *image_sum, int nthreads) {
    par_job j;
    j.sc = sc; j.iter0 = iter0; j.count = count; j.next = 0; j.rays = 0;
    j.N = sc->camera.resolution[0] * sc->camera.resolution[1];
    j.images = (pto_vec3 *)calloc((size_t)count * j.N, sizeof(pto_vec3));
    if (!j.images) return -1;
    pthread_mutex_init(&j.lock, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > count) nthreads = count;
    if (nthreads > 1024) nthreads = 1024;
    pthread_t *th = (pthread_t *)malloc((size_t)nthreads * sizeof(pthread_t));
    int started = 0;
    for (int t = 0; t < nthreads; ++t)
        if (pthread_create(&th[started], NULL, par_worker, &j) == 0) ++started;
    if (started == 0) par_worker(&j);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    for (int k = 0; k < count; ++k) {                       /* iteration order: the running sum of the reference */
        const pto_vec3 *im = j.images + (size_t)k * j.N;
        for (int p = 0; p < j.N; ++p) {
            image_sum[p].x += im[p].x; image_sum[p].y += im[p].y; image_sum[p].z += im[p].z;
        }
    }
    free(th);
    free(j.images);
    pthread_mutex_destroy(&j.lock);
    return j.rays;
}

uint64_t pto_fnv1a_i32(const int32_t *v, int stride_bytes, int n) {
    uint64_t h = 1469598103934665603ull;
    const uint8_t *p = (const uint8_t *)v;
    for (int i = 0; i < n; ++i) {
        uint32_t w = *(const uint32_t *)(p + (size_t)i * (size_t)stride_bytes);
        for (int b = 0; b < 4; ++b) {
            h ^= (w >> (8 * b)) & 0xffu;
            h *= 1099511628211ull;
        }
    }
    return h;
}

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* pathtrace.cu:284-393 with the 8.0 bounce loop in place of :339-377 */
void pto_trace_iteration(const pto_scene *sc, int iter, pto_vec3 *image, pto_path *paths,
                         pto_isect *isects, pto_stats *stats, pto_bounce_cb cb, void *user) {
    const int N = sc->camera.resolution[0] * sc->camera.resolution[1];
    pto_stats st;
    memset(&st, 0, sizeof st);
    double t0 = now_sec();
    pto_generate_rays_ex(sc, iter, paths);
    st.sec_other += now_sec() - t0;

    uint8_t *outside = (uint8_t *)malloc((size_t)N);
    void *scratch = malloc((size_t)N * (sizeof(pto_path) + sizeof(pto_isect) + 1));
    int n = N;

    if (sc->flags & PTO_F_FAKESHADE) {
        /* the reference as shipped: one bounce, fake shader (pathtrace.cu:339-377) */
        memset(isects, 0, (size_t)N * sizeof(pto_isect));
        pto_compute_intersections(n, paths, sc->geoms, sc->ngeoms, sc->tris, sc->meshes,
                                  sc->nmeshes, isects, outside);
        pto_shade_fake(iter, n, isects, paths, sc->materials);
        st.live[0] = n; st.rays = n; st.bounces = 1;
        if (cb) cb(user, 0, n, n, paths, isects);
    } else {
        for (int depth = 0; depth < sc->traceDepth && n > 0; ++depth) {
            double a = now_sec();
            memset(isects, 0, (size_t)N * sizeof(pto_isect));       /* pathtrace.cu:343 */
            int n_alive = n;
            if (!(sc->flags & PTO_F_COMPACT)) {
                /* compaction off: dead paths stay in place and are skipped */
                n_alive = 0;
                for (int i = 0; i < n; ++i) n_alive += paths[i].remainingBounces > 0;
                if (n_alive == 0) break;
            }
            pto_compute_intersections(n, paths, sc->geoms, sc->ngeoms, sc->tris, sc->meshes,
                                      sc->nmeshes, isects, outside);
            double b = now_sec();
            if (sc->flags & PTO_F_SORT)
                pto_sort_by_material(n, paths, isects, outside, scratch);
            double c = now_sec();
            pto_shade_scatter(iter, depth, n, isects, outside, paths, sc->materials, sc->trig);
            double d = now_sec();
            int n_before = n;
            if (sc->flags & PTO_F_COMPACT) n = pto_compact(n, paths, (pto_path *)scratch);
            double e = now_sec();
            st.sec_intersect += b - a;
            st.sec_shade += d - c;
            st.sec_other += (c - b) + (e - d);
            if (depth < 64) {
                st.live[depth] = n_alive;
                st.seq_hash[depth] = pto_fnv1a_i32(&paths[0].pixelIndex, (int)sizeof(pto_path), n);
            }
            st.rays += n_alive;
            st.bounces = depth + 1;
            if (cb) cb(user, depth, n_before, n, paths, isects);
        }
    }
    double g = now_sec();
    pto_final_gather(N, image, paths);                               /* pathtrace.cu:380-381 */
    st.sec_other += now_sec() - g;
    free(outside);
    free(scratch);
    if (stats) *stats = st;
}

/* ------------------------------------------------------------------------ */
/* multi-threaded iteration for the CPU baseline                             */
/* ------------------------------------------------------------------------ */
typedef struct {
    const pto_scene *sc; int iter, depth, begin, end;
    pto_path *paths; pto_isect *isects; uint8_t *outside;
} mt_job;

static void *mt_bounce(void *arg) {
    mt_job *j = (mt_job *)arg;
    memset(j->isects + j->begin, 0, (size_t)(j->end - j->begin) * sizeof(pto_isect));
    compute_intersections_range(j->begin, j->end, j->paths, j->sc->geoms, j->sc->ngeoms,
                                j->sc->tris, j->sc->meshes, j->sc->nmeshes, j->isects, j->outside);
    shade_scatter_range(j->iter, j->depth, j->begin, j->end, j->isects, j->outside, j->paths,
                        j->sc->materials, j->sc->trig);
    return NULL;
}

void pto_trace_iteration_mt(const pto_scene *sc, int iter, pto_vec3 *image, pto_path *paths,
                            pto_isect *isects, pto_stats *stats, int nthreads) {
    const int N = sc->camera.resolution[0] * sc->camera.resolution[1];
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pto_stats st;
    memset(&st, 0, sizeof st);
    pto_generate_rays_ex(sc, iter, paths);
    uint8_t *outside = (uint8_t *)malloc((size_t)N);
    pto_path *scratch = (pto_path *)malloc((size_t)N * sizeof(pto_path));
    pthread_t th[256];
    mt_job jobs[256];
    int n = N;
    for (int depth = 0; depth < sc->traceDepth && n > 0; ++depth) {
        for (int t = 0; t < nthreads; ++t) {
            mt_job *j = &jobs[t];
            j->sc = sc; j->iter = iter; j->depth = depth;
            j->begin = (int)((int64_t)n * t / nthreads);
            j->end = (int)((int64_t)n * (t + 1) / nthreads);
            j->paths = paths; j->isects = isects; j->outside = outside;
            pthread_create(&th[t], NULL, mt_bounce, j);
        }
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        if (depth < 64) st.live[depth] = n;
        st.rays += n;
        st.bounces = depth + 1;
        n = pto_compact(n, paths, scratch);
        if (depth < 64)
            st.seq_hash[depth] = pto_fnv1a_i32(&paths[0].pixelIndex, (int)sizeof(pto_path), n);
    }
    pto_final_gather(N, image, paths);
    free(outside);
    free(scratch);
    if (stats) *stats = st;
}

/* The rows [y0, y1) of one iteration: exactly the paths pto_trace_iteration traces for those pixels (a path's
 * whole history is keyed by (iter, global pixelIndex, depth): SURVEY 8.0), nothing else.  Lets the parity tests
 * hold a frame tile of a scene that is too heavy for a whole-frame oracle run (C4: 100 032 triangles per ray)
 * against the oracle.  image (N vec3, whole frame) accumulates the rows' pixels only. */
void pto_trace_rows_mt(const pto_scene *sc, int iter, pto_vec3 *image, int y0, int y1, pto_stats *stats, int nthreads) {
    const int W = sc->camera.resolution[0], H = sc->camera.resolution[1];
    const int N = W * H;
    if (y0 < 0) y0 = 0;
    if (y1 > H) y1 = H;
    pto_stats st;
    memset(&st, 0, sizeof st);
    if (stats) *stats = st;
    if (y1 <= y0) return;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pto_path *all = (pto_path *)malloc((size_t)N * sizeof(pto_path));
    pto_generate_rays_ex(sc, iter, all);                     /* generateRayFromCamera for every pixel; keep the rows */
    int n = (y1 - y0) * W;
    pto_path *paths = (pto_path *)malloc((size_t)n * sizeof(pto_path));
    memcpy(paths, all + (size_t)y0 * W, (size_t)n * sizeof(pto_path));
    free(all);
    pto_isect *isects = (pto_isect *)malloc((size_t)n * sizeof(pto_isect));
    uint8_t *outside = (uint8_t *)malloc((size_t)n);
    pto_path *scratch = (pto_path *)malloc((size_t)n * sizeof(pto_path));
    const int n0 = n;
    pthread_t th[256];
    mt_job jobs[256];
    for (int depth = 0; depth < sc->traceDepth && n > 0; ++depth) {
        for (int t = 0; t < nthreads; ++t) {
            mt_job *j = &jobs[t];
            j->sc = sc; j->iter = iter; j->depth = depth;
            j->begin = (int)((int64_t)n * t / nthreads);
            j->end = (int)((int64_t)n * (t + 1) / nthreads);
            j->paths = paths; j->isects = isects; j->outside = outside;
            pthread_create(&th[t], NULL, mt_bounce, j);
        }
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        if (depth < 64) st.live[depth] = n;
        st.rays += n;
        st.bounces = depth + 1;
        n = pto_compact(n, paths, scratch);
        if (depth < 64)
            st.seq_hash[depth] = pto_fnv1a_i32(&paths[0].pixelIndex, (int)sizeof(pto_path), n);
    }
    pto_final_gather(n0, image, paths);
    free(scratch); free(outside); free(isects); free(paths);
    if (stats) *stats = st;
}
