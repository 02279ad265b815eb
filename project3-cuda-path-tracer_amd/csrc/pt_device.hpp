// pt_device.hpp -- gfx950 device-side arithmetic of the path tracer's hot path.
//
// Everything here is written for one wave64 lane = one path.  The arithmetic
// reproduces, operation for operation and in binary32 without FMA contraction
// (the library is built with -ffp-contract=off), what the reference computes
// through GLM 0.9.6.3 in
//   src/intersections.h:12-144   utilhash, getPointOnRay, multiplyMV, box / sphere tests
//   src/interactions.h:10-42     calculateRandomDirectionInHemisphere
//   src/pathtrace.cu:41-45       makeSeededRandomEngine (+ thrust minstd_rand / u01)
// and the build-defined completion of src/interactions.h:69-79 (scatterRay;
// DESIGN.md section 3).  The data layout is NOT the reference's: geometry lives in
// LDS as 40-dword records, path state in SoA planes, normals are evaluated
// once for the winning primitive instead of once per primitive.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct f3 { float x, y, z; };

#define PTD __device__ __forceinline__

PTD f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PTD f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PTD f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PTD f3 mul(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PTD f3 scale(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
PTD f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
// glm dot(vec3): (x*x' + y*y') + z*z'   (func_geometric.inl:64-72)
PTD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// correctly rounded sqrt / divide (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), through
// the wave-gated rescale-free sequences defined further down when every lane is in their range
PTD float length_gated(f3 a);
PTD f3 normalize_gated(f3 a);
PTD float sqrt_gated(float x);
PTD float length(f3 a) { return length_gated(a); }
// glm normalize: x * (1 / sqrt(dot(x,x)))   (func_geometric.inl:153-159)
PTD f3 normalize(f3 a) { return normalize_gated(a); }
// ... of a vector that already IS a unit vector up to rounding (the reference normalises those again: see normalize_unit)
PTD f3 normalize_unit(f3 a);
PTD f3 cross(f3 x, f3 y) {
    return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// cross(x, y) for a y whose components are 0 or 1 (the sampler's axis, interactions.h:23-31): every product is exact, so
// each component's multiply-multiply-subtract rounds once, like an fma on one exact product (same zero signs, same NaNs)
PTD f3 cross_axis(f3 x, f3 y) {
    return mk(__builtin_fmaf(x.y, y.z, -(y.y * x.z)), __builtin_fmaf(x.z, y.x, -(y.z * x.x)), __builtin_fmaf(x.x, y.y, -(y.x * x.y)));
}
// glm reflect: I - N * dot(N, I) * 2   (func_geometric.inl:175-179).  The doubling is exact (also of a subnormal), so
// I - fl(2 p) = fl(I - 2 p) is ONE fma per component on the rounded product p = fl(N dot): three instructions less
PTD f3 reflect(f3 I, f3 N) {
    const f3 p = scale(N, dot(N, I));
    return mk(__builtin_fmaf(-2.0f, p.x, I.x), __builtin_fmaf(-2.0f, p.y, I.y), __builtin_fmaf(-2.0f, p.z, I.z));
}

// ---------------------------------------------------------------------------
// RNG: utilhash (intersections.h:12-20), minstd_rand, uniform_real<float>(0,1)
// ---------------------------------------------------------------------------
PTD uint32_t utilhash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

// x mod (2^31-1) by Mersenne folding; valid for x < 2^47
PTD uint32_t mod_m31(uint64_t x) {
    uint32_t r = (uint32_t)(x & 0x7fffffffull) + (uint32_t)(x >> 31);
    return r >= 0x7fffffffu ? r - 0x7fffffffu : r;
}

// thrust::default_random_engine(h): seed = h % m, 0 -> 1
PTD uint32_t lcg_seed(uint32_t s) {
    uint32_t r = mod_m31((uint64_t)s);
    return r == 0u ? 1u : r;
}

// pathtrace.cu:41-45
PTD uint32_t seeded_engine(int iter, int index, int depth) {
    uint32_t k = (1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter;
    return lcg_seed(utilhash(k) ^ utilhash((uint32_t)index));
}

// one minstd step (a = 48271) followed by thrust's u01 map: float(x-1) / 2^31
PTD float u01(uint32_t &state) {
    state = mod_m31(48271ull * (uint64_t)state);
    return (float)(state - 1u) / 2147483648.0f;
}

// ---------------------------------------------------------------------------
// shared sin/cos: binary32, explicit fused multiply-adds, the oracle's sequence operation for operation
// (oracle/ptoracle.c: pto_sincos has the derivation; DESIGN.md "shared trig").  Within 1 ulp of the correctly rounded
// value on every float of [0, 2 pi].  27 f32 instructions + the quadrant selects; rounds 1-3 evaluated fdlibm's
// binary64 polynomials here (~40 binary64 instructions at twice the issue cost: 9 % of k_bounce's issue cycles).
// ---------------------------------------------------------------------------
PTD void sincos_shared(float x, float &s, float &c) {
    const float TWO_OVER_PI = 0.636619772367581343f;
    const float MAGIC = 12582912.0f;
    const float P1 = 1.57079625129699707031f, P2 = 7.54978941586159635335e-08f, P3 = 5.39030285815811905290e-15f;
    const float S1 = -1.66666671633720398e-01f, S2 = 8.33333190530538559e-03f,
                S3 = -1.98401714442297816e-04f, S4 = 2.72681563728838228e-06f;
    const float C1 = 4.16666530072689056e-02f, C2 = -1.38876168057322502e-03f, C3 = 2.44678121816832572e-05f;
    const float kf = __builtin_fmaf(x, TWO_OVER_PI, MAGIC) - MAGIC;
    const float r1 = __builtin_fmaf(-kf, P1, x);
    const float r = __builtin_fmaf(-kf, P2, r1);
    const float rl = __builtin_fmaf(-kf, P3, __builtin_fmaf(-kf, P2, r1 - r));
    const float z = r * r;
    const float ze = __builtin_fmaf(r, r, -z);
    const float ps = __builtin_fmaf(z, __builtin_fmaf(z, __builtin_fmaf(z, S4, S3), S2), S1);
    const float pc = __builtin_fmaf(z, __builtin_fmaf(z, C3, C2), C1);
    const float hz = 0.5f * z;
    const float w = 1.0f - hz;
    float e = (1.0f - w) - hz;
    e = __builtin_fmaf(-0.5f, ze, e);
    const float u = (r * z) * ps;
    const float sn = r + __builtin_fmaf(rl, w, u);
    const float cs = w + __builtin_fmaf(z * z, pc, __builtin_fmaf(-rl, sn, e));
    const int q = (int)kf & 3;
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    so = (q & 2) ? -so : so;
    co = ((q + 1) & 2) ? -co : co;
    s = so;
    c = co;
}

// interactions.h:10-42
PTD f3 hemisphere(f3 normal, uint32_t &rng) {
    const float TWO_PI = 6.2831853071795864769252867665590057683943f;
    const float SQRT_OF_ONE_THIRD = 0.5773502691896257645091487805019574556476f;
    float up = sqrt_gated(u01(rng));
    float over = sqrt_gated(1 - up * up);
    float around = u01(rng) * TWO_PI;
    f3 notN;
    if (__builtin_fabsf(normal.x) < SQRT_OF_ONE_THIRD) notN = mk(1, 0, 0);
    else if (__builtin_fabsf(normal.y) < SQRT_OF_ONE_THIRD) notN = mk(0, 1, 0);
    else notN = mk(0, 0, 1);
    f3 p1 = normalize(cross_axis(normal, notN));
    f3 p2 = normalize_unit(cross(normal, p1));           // two perpendicular unit vectors
    float sa, ca;
    sincos_shared(around, sa, ca);
    return add(add(scale(normal, up), scale(p1, ca * over)), scale(p2, sa * over));
}

// ---------------------------------------------------------------------------
// scene records in LDS
// ---------------------------------------------------------------------------
// geom record: 40 dwords.  [0] type [1] materialid [2] first_tri [3] tri_count
// [4..15] inverseTransform cols 0..3 x rows 0..2   [16..27] transform   [28..39] invTranspose
constexpr int GEOM_WORDS = 40;
constexpr int G_INV = 4, G_FWD = 16, G_INVT = 28;
// material record: 12 dwords. color[3] spec_color[3] hasReflective hasRefractive ior emittance pad pad
constexpr int MAT_WORDS = 12;

// vec3(m * vec4(v, 1)): (m0*v.x + m1*v.y) + (m2*v.z + m3*1)   (type_mat4x4.inl:617-628)
template <typename P> PTD f3 mv_point(P m, f3 v) {
    f3 r;
    r.x = (m[0] * v.x + m[3] * v.y) + (m[6] * v.z + m[9]);
    r.y = (m[1] * v.x + m[4] * v.y) + (m[7] * v.z + m[10]);
    r.z = (m[2] * v.x + m[5] * v.y) + (m[8] * v.z + m[11]);
    return r;
}
// vec3(m * vec4(v, 0)): the m3 * 0.0f product is kept (it is +-0, or NaN for a non-finite matrix) -- as the addend's
// multiplier of one fma: the product is EXACT, so fl(fl(m2 v.z) + m3 * 0) has the same single rounding, the same zero
// signs and the same NaNs as the reference's multiply-then-add (three instructions less per call)
template <typename P> PTD f3 mv_dir(P m, f3 v) {
    f3 r;
    r.x = (m[0] * v.x + m[3] * v.y) + __builtin_fmaf(m[9], 0.0f, m[6] * v.z);
    r.y = (m[1] * v.x + m[4] * v.y) + __builtin_fmaf(m[10], 0.0f, m[7] * v.z);
    r.z = (m[2] * v.x + m[5] * v.y) + __builtin_fmaf(m[11], 0.0f, m[8] * v.z);
    return r;
}

// getPointOnRay (intersections.h:27-29): o + (t - .0001f) * normalize(d)
PTD f3 point_on_ray(f3 o, f3 d, float t) { return add(o, scale(normalize_unit(d), (t - .0001f))); }

struct Hit {
    float t;        // world distance, FLT_MAX while nothing is hit
    int geom;       // winning geom index, -1 = miss
    int outside;    // `outside` flag of the winning test
    f3 aux;         // cube: object-space face normal; sphere: object-space hit point;
                    // mesh: (bits of) the winning triangle index in aux.x
};

// boxIntersectionTest (intersections.h:48-90) without the normal (deferred).  The candidate face
// normals tmin_n / tmax_n are carried as a 3-bit code (axis*2 + (sign>0), 7 = the zero vector
// glm's default constructor leaves when no slab updates them) instead of three floats: one
// select per update instead of three.  face_from_code() rebuilds the exact vector.
PTD f3 face_from_code(int code) {
    const float s = (code & 1) ? 1.0f : -1.0f;
    const int axis = code >> 1;
    return mk(axis == 0 ? s : 0.0f, axis == 1 ? s : 0.0f, axis == 2 ? s : 0.0f);   // code 7 -> (0,0,0)
}

#ifndef PT_FASTDIV
#define PT_FASTDIV 1
#endif
// ---- correctly rounded divides that share one reciprocal ---------------------------------
// hipcc expands `n / d` into v_div_scale x2, v_rcp, a Newton step on the reciprocal, a quotient
// with two residual corrections (v_div_fmas last) and v_div_fixup: 11 instructions.  When neither
// operand needs v_div_scale's rescaling and the quotient is a normal number, the two scales are
// identities, v_div_fmas is a plain fma and v_div_fixup returns its input, so the same value comes
// out of the 8 instructions below -- and the 3 that refine the reciprocal are shared by every
// divide with the same denominator (the slab test divides twice by each direction component).
// No-rescale conditions (ISA, v_div_scale_f32): d normal, 1/d normal, |exp(n) - exp(d)| < 96,
// n/d normal, biased exp(n) > 23; they hold for 2^-40 <= |d| <= 2^40 and (n == +0 or
// 2^-25 <= |n| < 2^55).  slab_fast_ok() establishes that per wave; otherwise the plain `/` runs.
PTD float rcp_refined(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
PTD float div_by_rcp(float n, float d, float r) {
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}
// Correctly rounded sqrt and 1 / sqrt by Newton's iteration on v_rsq_f32 (round 5; rounds 2-4: v_sqrt_f32 and the
// compiler's two residual tests, then v_rcp_f32 + a refinement for the reciprocal -- 9 + 7 instructions of which six
// issue at half rate):  y = rsq(x);  g = x y;  h = y / 2;  r = 1/2 - h g;  g += g r;  h += h r;  s = g + h (x - g^2)
// (Markstein's form: eight instructions, one of them transcendental) and, for glm::normalize's 1 / s, two corrections
// q += q (1 - s q) from the float one ulp ABOVE 2h -- a start at or above 1 / s: from 2h itself the two roots per odd
// binade whose mantissa is all ones (1 / s a hair above a midpoint next to a power of two) end on the tie and round down.
// EXACT -- s = RN(sqrt x), q = RN(1 / s), bit for bit the compiler's correctly rounded sqrtf and divide -- for EVERY
// binary32 x in [2^-102, 2^128): checked exhaustively on the device (profiles/microbench/sqrt_exhaustive.hip: variant 1,
// 1.93 * 10^9 arguments; tests/test_gpu_pins.py runs the same sweep through pt_probe_sqrt).  The callers' gates
// (2^-96 <= x for the root, 2^-80 <= x <= 2^80 for normalize) lie inside.
PTD void sqrt_newton(float x, float &s, float &h) {
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    s = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
PTD float sqrt_normal_range(float x) {
    float s, h;
    sqrt_newton(x, s, h);
    return s;
}
// s = RN(sqrt x) and RN(1 / s)
PTD float rsqrt_of_root(float x) {
    float s, h;
    sqrt_newton(x, s, h);
    float q = __uint_as_float(__float_as_uint(h + h) + 1u);
    q = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), q, q);
    return __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), q, q);
}
// fl(1 / fl(sqrt x)) -- glm::normalize's factor -- for x within 2^-12 of 1, i.e. for a vector that is a unit vector up to
// rounding.  The reference normalises such vectors again in three places per bounce: q.direction in getPointOnRay
// (intersections.h:27-29, inside both intersection tests), the path's ray in the shader's getPointOnRay, and the cross
// product of two perpendicular unit vectors in the sampler (interactions.h:33-34).  With e = x - 1, 1 / sqrt(x) =
// 1 - e/2 + 3 e^2 / 8 - ..., and the chain's two roundings (the root to nearest, then its reciprocal to nearest) come out as
// "1 - e/2 rounded UP to a multiple of 2^-23": for x > 1 (e = k 2^-23) 1 - (k & ~1) 2^-24, for x < 1 (e = -k 2^-24)
// 1 + ceil(k / 4) 2^-23.  Rounding up on that grid = rounding to nearest after adding 1.5 * 2^-25 (-e/2 is a multiple of
// 2^-25: never a tie), done in [1, 2) where the grid is the float spacing: four additions, no transcendental, against
// the fourteen instructions of rsqrt_of_root.  EXACT for every float from 1 - 8190 * 2^-24 to 1 + 2897 * 2^-23 (each of
// the 11 088 checked against sqrtf and the divide: tests/test_arith_models_cpu.py; on the device by pt_probe_sqrt, whose
// sweep takes this path inside the callers' gate [1 - 2^-12, 1 + 2^-12]).
constexpr float NEAR_ONE_LO = 0x1.ffep-1f, NEAR_ONE_HI = 0x1.001p+0f;
PTD float rsqrt_near_one(float x) {
    const float e = x - 1.0f;                                   // exact
    const float w = __builtin_fmaf(e, -0.5f, 0x1.8p-25f);        // exact: a multiple of 2^-26 below 2^-11
    const float t = w + (1.0f + 0x1p-10f);                       // the one rounding, to nearest on [1, 2)'s grid
    return t - 0x1p-10f;                                         // exact
}
// every active lane agrees: no lane votes against.  (__all() compiles to select 0/1 + compare + compare with exec;
// the ballot of the NEGATED predicate is the two v_cmp themselves and one scalar compare with zero.)
#ifndef PT_WAVE_ALL_BALLOT
#define PT_WAVE_ALL_BALLOT 1
#endif
PTD bool wave_all(bool p) {
#if PT_WAVE_ALL_BALLOT
    return __builtin_amdgcn_ballot_w64(!p) == 0ull;
#else
    return __all(p);
#endif
}
// lo <= x <= hi for 0 < lo <= hi as ONE unsigned compare: positive floats order like their bit patterns, so the test is
// "bits(x) - bits(lo) does not exceed bits(hi) - bits(lo)"; a negative x, a zero of either sign and every NaN wrap to a
// distance beyond any such span and fail, which is the safe side (the callers' general forms are exact for everything).
// An integer subtraction and one compare issue in 2.5 + 4.4 cycles against the 8.8 of two float compares, and a tile
// passes seventeen of these gates.
PTD bool in_range_bits(float x, float lo, float hi) {
    return (__float_as_uint(x) - __float_as_uint(lo)) <= (__float_as_uint(hi) - __float_as_uint(lo));
}
// wave-uniform gates for the two helpers below (NaN fails; inactive lanes do not vote)
PTD bool all_in_range(float x, float lo, float hi) {
#if PT_FASTDIV
    return wave_all(in_range_bits(x, lo, hi));
#else
    (void)x; (void)lo; (void)hi;
    return false;
#endif
}
// glm normalize, v * (1 / sqrt(dot)), for 2^-80 <= dot <= 2^80 (sqrt and divide both rescale-free)
PTD f3 normalize_normal_range(f3 a, float dt) {
    return scale(a, rsqrt_of_root(dt));
}
// Wave-uniform gate for the rescale-free paths of one cube test, evaluated BEFORE the direction is
// normalised (the squares are the ones the dot product needs anyway): with v = M^-1 d, x = |v|^2,
//   2^-80 <= x <= 2^80                    -> sqrt and 1/sqrt need no rescaling,
//   min(v_k^2) >= 2^-78 x                  -> every normalised component is at least 2^-40 (and <= 1+),
//   max|qo_k| < 2^54                       -> numerators (+-0.5 - qo) are +0 or in [2^-25, 2^55).
// NaNs fail the ordered compares.  Inactive lanes do not vote.
PTD bool norm_fast_ok(float x) {
#if PT_FASTDIV
    return wave_all(in_range_bits(x, 8.271806125530277e-25f, 1.2089258196146292e24f));
#else
    (void)x;
    return false;
#endif
}
PTD bool cube_fast_ok(f3 qo, f3 v, float x) {
#if PT_FASTDIV
    const float sq_min = __builtin_fminf(__builtin_fminf(v.x * v.x, v.y * v.y), v.z * v.z);
    const float omax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(qo.x), __builtin_fabsf(qo.y)), __builtin_fabsf(qo.z));
    const bool finite = (qo.x + qo.y + qo.z) == (qo.x + qo.y + qo.z);          // fmax drops NaNs
    const bool ok = finite && in_range_bits(x, 8.271806125530277e-25f, 1.2089258196146292e24f) &&
                    sq_min >= x * 3.308722450212111e-24f && omax < 1.8014398509481984e16f;
    return wave_all(ok);
#else
    (void)qo; (void)v; (void)x;
    return false;
#endif
}

// The object-space ray both tests start from (intersections.h:50-52,105-107): qo = M^-1 * (ro, 1),
// v = M^-1 * (rd, 0), x = dot(v, v); q.direction = glm::normalize(v) = v * (1 / sqrt(x)).
template <typename P> PTD void object_ray(P inv, f3 ro, f3 rd, f3 &qo, f3 &v, float &x) {
    qo = mv_point(inv, ro);
    v = mv_dir(inv, rd);
    x = dot(v, v);
}
// glm::normalize(v) given x = dot(v, v); `fast` = norm_fast_ok(x) (wave-uniform)
PTD f3 normalize_with(f3 v, float x, bool fast) {
    if (fast) return normalize_normal_range(v, x);
    return scale(v, 1.0f / __builtin_sqrtf(x));
}

// slab part of boxIntersectionTest (intersections.h:54-84) on the object-space ray (qo, qd): true when the
// test passes; t_obj = the parameter it settles on, code = face normal code.  `fast` = cube_fast_ok(...) of
// the lanes that run this (wave-uniform among them): the rescale-free divides.
PTD bool cube_slabs(f3 qo, f3 qd, bool fast, float &t_obj, int &code_out, int &outside) {
    float t1x, t2x, t1y, t2y, t1z, t2z;
    float tax, tbx, tay, tby, taz, tbz;
    if (fast) {                                                           // wave-uniform
        const float rx = rcp_refined(qd.x), ry = rcp_refined(qd.y), rz = rcp_refined(qd.z);
        t1x = div_by_rcp(-0.5f - qo.x, qd.x, rx); t2x = div_by_rcp(+0.5f - qo.x, qd.x, rx);
        t1y = div_by_rcp(-0.5f - qo.y, qd.y, ry); t2y = div_by_rcp(+0.5f - qo.y, qd.y, ry);
        t1z = div_by_rcp(-0.5f - qo.z, qd.z, rz); t2z = div_by_rcp(+0.5f - qo.z, qd.z, rz);
        // all six quotients are finite and non-NaN here (|d| >= 2^-40, |n| < 2^55): glm's min/max
        // selects reduce to v_min/v_max (they can differ only in the sign of a zero, which no
        // comparison below distinguishes)
        tax = __builtin_fminf(t1x, t2x); tbx = __builtin_fmaxf(t1x, t2x);
        tay = __builtin_fminf(t1y, t2y); tby = __builtin_fmaxf(t1y, t2y);
        taz = __builtin_fminf(t1z, t2z); tbz = __builtin_fmaxf(t1z, t2z);
    } else {
        t1x = (-0.5f - qo.x) / qd.x; t2x = (+0.5f - qo.x) / qd.x;
        t1y = (-0.5f - qo.y) / qd.y; t2y = (+0.5f - qo.y) / qd.y;
        t1z = (-0.5f - qo.z) / qd.z; t2z = (+0.5f - qo.z) / qd.z;
        tax = t1x < t2x ? t1x : t2x; tbx = t1x > t2x ? t1x : t2x;         // glm::min / glm::max
        tay = t1y < t2y ? t1y : t2y; tby = t1y > t2y ? t1y : t2y;
        taz = t1z < t2z ? t1z : t2z; tbz = t1z > t2z ? t1z : t2z;
    }
    float tmin = -1e38f, tmax = 1e38f;
    int tmin_c = 7, tmax_c = 7;
#define PTD_SLAB(T1, T2, TA, TB, AXIS)                                    \
    {                                                                     \
        float ta = (TA), tb = (TB);                                       \
        int code = (AXIS) * 2 + ((T2) < (T1) ? 1 : 0);                    \
        if (ta > 0 && ta > tmin) { tmin = ta; tmin_c = code; }            \
        if (tb < tmax) { tmax = tb; tmax_c = code; }                      \
    }
    PTD_SLAB(t1x, t2x, tax, tbx, 0)
    PTD_SLAB(t1y, t2y, tay, tby, 1)
    PTD_SLAB(t1z, t2z, taz, tbz, 2)
#undef PTD_SLAB
    if (tmax >= tmin && tmax > 0) {
        outside = 1;
        if (tmin <= 0) { tmin = tmax; tmin_c = tmax_c; outside = 0; }
        t_obj = tmin; code_out = tmin_c;
        return true;
    }
    return false;
}

// glm normalize / length with the wave-uniform rescale-free gate
PTD f3 normalize_gated(f3 a) {
    const float x = dot(a, a);
    if (all_in_range(x, 8.271806125530277e-25f, 1.2089258196146292e24f)) return normalize_normal_range(a, x);
    return scale(a, 1.0f / __builtin_sqrtf(x));
}
PTD f3 normalize_unit(f3 a) {
    const float x = dot(a, a);
    if (all_in_range(x, NEAR_ONE_LO, NEAR_ONE_HI)) return scale(a, rsqrt_near_one(x));
    if (all_in_range(x, 8.271806125530277e-25f, 1.2089258196146292e24f)) return normalize_normal_range(a, x);
    return scale(a, 1.0f / __builtin_sqrtf(x));
}
PTD float sqrt_gated(float x) {
    if (all_in_range(x, 1.2621774483536189e-29f, 3.0e38f)) return sqrt_normal_range(x);
    return __builtin_sqrtf(x);
}
PTD float length_gated(f3 a) { return sqrt_gated(dot(a, a)); }

// shared tail of both tests (intersections.h:85-87,136-143): objP = getPointOnRay(q, t_obj);
// worldP = transform * objP; t = length(r.origin - worldP).  `fwd` = 12 floats of the transform.
template <typename P> PTD float world_distance(P fwd, f3 ro, f3 qo, f3 qd, float t_obj, f3 &obj_p) {
    obj_p = add(qo, scale(normalize_unit(qd), (t_obj - .0001f)));         // getPointOnRay (qd = glm::normalize(v))
    return length_gated(sub(ro, mv_point(fwd, obj_p)));
}

// root part of sphereIntersectionTest (intersections.h:110-134) on the object-space ray (o, d)
PTD bool sphere_roots(f3 o, f3 d, float &t_obj, int &outside) {
    float vDotDirection = dot(o, d);
    float radicand = vDotDirection * vDotDirection - (dot(o, o) - (0.5f * 0.5f));
    if (radicand < 0) return false;
    float squareRoot = sqrt_gated(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;
    if (t1 < 0 && t2 < 0) {
        return false;
    } else if (t1 > 0 && t2 > 0) {
        t_obj = (t2 < t1) ? t2 : t1;      // std::min(t1, t2)
        outside = 1;
    } else {
        t_obj = (t1 < t2) ? t2 : t1;      // std::max(t1, t2)
        outside = 0;
    }
    return true;
}


// surface normal of the winning primitive (the part of the two tests above
// that the reference evaluates for every candidate)
// `fwd` / `invt`: the 12 floats (4 columns x 3 rows) of the transform / inverse-transpose
template <typename P> PTD f3 cube_normal(P fwd, f3 face_n) {
    return normalize(mv_dir(fwd, face_from_code(__float_as_int(face_n.x))));
}
template <typename P> PTD f3 sphere_normal(P invt, f3 obj_p, int outside) {
    f3 n = normalize(mv_dir(invt, obj_p));
    return outside ? n : neg(n);
}

// glm::intersectRayTriangle (gtx/intersect.inl:37-74) on (v0, e1 = v1-v0, e2 = v2-v0);
// returns true and bary.z in tz on a hit.
//
// The reference divides first (f = 1/a) and then rejects on bary.x = f*dot(s,p) outside [0,1].
// Almost every (ray, triangle) pair is rejected there, so the divide is skipped whenever the
// outcome of that test is certain from dot(s,p) and a alone:
//   * sp < -(a*1e-30)      =>  f*sp is negative and cannot round to -0   =>  bary.x < 0
//   * sp >  a*(1 + 2^-20)  =>  f*sp > 1 after both roundings (2^-24 each) =>  bary.x > 1
// (only for a < 1e30, where f is a normal number).  Anything else -- including every NaN -- takes
// the exact path below, so the result is identical to the straight transcription in every case.
PTD bool ray_triangle(f3 orig, f3 dir, f3 v0, f3 e1, f3 e2, float &tz) {
    f3 p = cross(dir, e2);
    float a = dot(e1, p);
    if (a < 1.1920928955078125e-07f) return false;
    f3 s = sub(orig, v0);
    float sp = dot(s, p);
    if (a < 1e30f && (sp < -(a * 1e-30f) || sp > a * 1.00000095367431640625f)) return false;
    float f = 1.0f / a;
    float bx = f * sp;
    if (bx < 0.0f) return false;
    if (bx > 1.0f) return false;
    f3 q = cross(s, e1);
    float by = f * dot(dir, q);
    if (by < 0.0f) return false;
    if (by + bx > 1.0f) return false;
    tz = f * dot(e2, q);
    return tz >= 0.0f;
}

// Completion spec 8.0 "Triangles", hit-point test (oracle/ptoracle.c: pto_tri_point_ok): an accepted triangle
// counts only when the point it reports, fl(o_k + fl(d_k * tz)), lies inside the triangle's bounding box
// [min, max](v0_k, fl(v0_k + e1_k), fl(v0_k + e2_k)) widened by the mesh's pad (word 10 of the triangle
// record).  Runs on accepted hits only -- a handful per ray.
PTD bool tri_point_ok(f3 o, f3 d, float tz, f3 v0, f3 e1, f3 e2, float pad) {
    auto axis = [&](float ok, float dk, float a, float b, float c) {
        const float p = ok + dk * tz;
        const float x1 = a + b, x2 = a + c;
        const float lo = __builtin_fminf(a, __builtin_fminf(x1, x2)), hi = __builtin_fmaxf(a, __builtin_fmaxf(x1, x2));
        return p >= lo - pad && p <= hi + pad;
    };
    return axis(o.x, d.x, v0.x, e1.x, e2.x) && axis(o.y, d.y, v0.y, e1.y, e2.y) && axis(o.z, d.z, v0.z, e1.z, e2.z);
}

// ---------------------------------------------------------------------------
// shading / scattering (completion of interactions.h:69-79)
// ---------------------------------------------------------------------------
struct PathState {
    f3 o, d, c;
};

// returns true when the path stays alive; on false `ps.c` is the final colour
PTD bool shade_scatter(PathState &ps, float t, f3 n, int matId, int outside, const float *mats,
                       int iter, int pixel, int depth, bool last_bounce) {
    if (t > 0.0f) {
        const float *m = mats + matId * MAT_WORDS;
        f3 mcol = mk(m[0], m[1], m[2]);
        float emittance = m[9];
        if (emittance > 0.0f) {
            ps.c = mul(ps.c, scale(mcol, emittance));      // pathtrace.cu:247-249
            return false;
        }
        // the last bounce: whatever the scatter would produce, the path ends with colour 0 (remainingBounces reaches 0,
        // completion spec 8.0) -- no engine, no direction (wave-uniform: the whole launch takes this exit)
        if (last_bounce) {
            ps.c = mk(0.0f, 0.0f, 0.0f);
            return false;
        }
        uint32_t rng = seeded_engine(iter, pixel, depth);
        f3 P = point_on_ray(ps.o, ps.d, t);
        f3 I = ps.d;
        f3 scol = mk(m[3], m[4], m[5]);
        if (m[6] > 0.0f) {                                 // mirror
            ps.d = reflect(I, n);
            ps.o = P;
            ps.c = mul(ps.c, scol);
        } else if (m[7] > 0.0f) {                          // Fresnel dielectric
            float d0 = dot(I, n);
            f3 nn = d0 > 0.0f ? neg(n) : n;
            float ior = m[8];
            float eta = outside ? (1.0f / ior) : ior;
            float dv = dot(nn, I);
            float k = 1.0f - eta * eta * (1.0f - dv * dv);
            bool refl;
            if (k < 0.0f) {
                refl = true;
            } else {
                float r0 = (1.0f - ior) / (1.0f + ior);
                r0 = r0 * r0;
                float cm = 1.0f - (-dv);
                float c5 = (((cm * cm) * cm) * cm) * cm;
                float R = r0 + (1.0f - r0) * c5;
                float u = u01(rng);
                refl = u < R;
                if (!refl) {
                    ps.d = sub(scale(I, eta), scale(nn, (eta * dv + sqrt_gated(k))));
                    ps.o = add(P, scale(I, 0.0002f));
                }
            }
            if (refl) {
                ps.d = reflect(I, nn);
                ps.o = P;
            }
            ps.c = mul(ps.c, scol);
        } else {                                           // diffuse
            ps.d = hemisphere(n, rng);
            ps.o = P;
            ps.c = mul(ps.c, mcol);
        }
        return true;
    }
    ps.c = mk(0.0f, 0.0f, 0.0f);                           // pathtrace.cu:262-264
    return false;
}

#undef PTD
}  // namespace ptd
