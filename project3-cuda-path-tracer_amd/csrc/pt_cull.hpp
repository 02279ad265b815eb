// pt_cull.hpp -- host side (pt_init): per-primitive world-space boxes for the cull stage of the intersection
// kernels (pt_kernels.hpp, stage 1).  A ray that misses a primitive's box is not handed to the exact
// object-space test at all, so the box must contain every ray for which the REFERENCE'S OWN FLOAT ARITHMETIC
// (boxIntersectionTest / sphereIntersectionTest, src/intersections.h:48-144) could report a hit.
//
// Setting.  A = upper 3x3 of inverseTransform, b = its translation column (floats, taken as exact reals),
// u = 2^-24.  For a world ray (o, d) with |o|_1 <= R (the kernel treats every other ray as a candidate of every
// primitive) the exact object-space ray is  q* = A o + b,  v* = A d.  With  a_k = sum_j |A_kj|  (row sums),
// S_k = a_k R + |b_k|,  S = max_k S_k,  N = |A^-1|_F:
//
//  (1) what the reference computes.  multiplyMV rounds 3 products and 3 sums per component:
//        |q_k - q*_k| <= 3u S_k,      |v_k - v*_k| <= 3u a_k |d|_inf.
//      glm::normalize multiplies v by ONE float (1 / sqrt(dot)), so the computed direction is v's direction up
//      to 2u per component; relative to v* its k-th component is off by at most 3u a_k |d|_inf / |v*| + 2u
//      <= 3u a_k N + 2u   (|v*| >= |d|_2 / |A^-1|_2).
//  (2) the slab test on the float ray (q, qd):  t = fl(fl(+-0.5 - q_k) / qd_k) is the exact parameter of the plane
//      moved by at most 2u (0.5 + |q_k|): the float test is an EXACT slab test against a box whose six planes
//      moved by that much.  Its decision rule (tmin over the positive entries only, `tmax >= tmin && tmax > 0`)
//      answers "miss" whenever that exact test does: if max ta > min tb > 0 the max is positive, hence counted;
//      if min tb <= 0 the `tmax > 0` clause fails.  0/0 (origin exactly on a plane, direction parallel to it) drops
//      the axis from both comparisons, which is the slab test of the remaining two axes: still a miss.
//  (3) points of the float ray at parameter t <= |q| + 2 (beyond that both rays have left the unit cube's
//      neighbourhood) lie within  3u S_k + (sqrt3 S + 2)(3u a_k N + 2u)  of the exact ray's, per axis k.
//  =>  if the EXACT ray (as a half line t >= 0) misses the cube grown by
//        p_k = u [ 2 (0.5 + S_k) + 3 S_k + (sqrt3 S + 2)(3 a_k N + 2) ]      per object axis k,
//      the reference returns -1.  Sphere: the cube contains the sphere; radicand = (o.d)^2 - (o.o - 0.25) is
//      evaluated with an absolute error <= 16u D^2 (D^2 = |q|^2 <= 3 S^2: three rounded dots of magnitude D^2 and
//      the |d|^2 = 1 + 4u the formula ignores), so a line passing the centre at distance >= 0.5 + p has a
//      negative float radicand once p > 16u D^2; and when the line does cut the sphere behind an origin that is
//      outside the grown cube, o.d >= sqrt(p) keeps `firstTerm + squareRoot` negative under the same condition.
//      p_k(sphere) = p_k + 32u (3 S^2 + 1).
//  (4) the world box is the image of the grown cube under A^-1 (computed in binary64 from the float matrix the
//      reference actually uses, not from `transform`):  centre c = -A^-1 b,  half extent
//        H_i = sum_k |A^-1_ik| (0.5 + p_k),
//      plus 64u (R + |c_i| + H_i) for the kernel's own test (v_rcp_f32: 1 ulp; n = fl(-o / d) and two fused
//      multiply-adds per plane, t = fl(fl(c / d + n) -+ H |1/d|) (pt_k_scene.hpp: cull_box, centre / half-extent form):
//      absolute error <= u (3 |o| + 2 |c| + H) |1/d| in t, i.e. 3u (R + |c_i| + H_i) in space; the kernel's (c, H) are
//      floats with [c - H, c + H] containing [lo, hi] exactly -- centre_half below).
// All p_k carry a further factor 4.  A matrix that is singular, non-finite, or has N * max a_k > 2^20 gets the
// box (-inf, +inf): every ray is a candidate and the exact test decides, as in the reference.
//
// tests/test_gpu_parity.py::test_cull_box_gates aims rays at the faces, edges and corners of the boxes from both
// sides (offsets from 1e-7 to 10 pads), along the axes, from inside, from far outside R, with NaN / inf / zero
// components, and compares with the oracle bit for bit; tests/test_cull_cpu.py checks on the CPU that no ray the
// oracle hits is outside its primitive's box.
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace ptcull {

struct Box { float lo[3], hi[3]; };

inline bool invert3(const double A[3][3], double Ai[3][3]) {
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    const double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    const double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    Ai[0][0] = c00 * id; Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id; Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
    Ai[1][0] = c01 * id; Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id; Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
    Ai[2][0] = c02 * id; Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id; Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            if (!std::isfinite(Ai[i][j])) return false;
    return true;
}

// inverseTransform as 16 floats m[col][row] (glm layout) -> A, b; false when not usable for culling
struct Affine { double A[3][3], b[3], Ai[3][3], c[3]; bool ok; };
inline Affine analyse(const float *inv16) {
    Affine f{};
    f.ok = true;
    for (int k = 0; k < 3; ++k) {
        for (int j = 0; j < 3; ++j) f.A[k][j] = (double)inv16[j * 4 + k];
        f.b[k] = (double)inv16[3 * 4 + k];
    }
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(f.b[k]) || std::fabs(f.b[k]) > 0x1p60) f.ok = false;
        for (int j = 0; j < 3; ++j)
            if (!std::isfinite(f.A[k][j]) || std::fabs(f.A[k][j]) > 0x1p40) f.ok = false;
    }
    if (f.ok && !invert3(f.A, f.Ai)) f.ok = false;
    if (f.ok) {
        double amax = 0.0, nf = 0.0;
        for (int k = 0; k < 3; ++k) {
            double a = 0.0;
            for (int j = 0; j < 3; ++j) { a += std::fabs(f.A[k][j]); nf += f.Ai[k][j] * f.Ai[k][j]; }
            amax = std::fmax(amax, a);
        }
        nf = std::sqrt(nf);
        if (!(nf * amax <= 0x1p20) || !(nf <= 0x1p40)) f.ok = false;
        for (int i = 0; i < 3; ++i) f.c[i] = -(f.Ai[i][0] * f.b[0] + f.Ai[i][1] * f.b[1] + f.Ai[i][2] * f.b[2]);
    }
    return f;
}

inline float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

// inv16[n]: the inverseTransform of each primitive; is_sphere[n]; skip[n]: no box wanted (meshes).
// extra_points: further world points rays may start from (the camera), 3 doubles each.
// Returns R (the |origin|_1 bound) and fills `out` (infinite boxes where culling is off).
inline float make_boxes(const float *const *inv16, const bool *is_sphere, const bool *skip, int n,
                        const double *extra_points, int n_extra, std::vector<Box> &out) {
    const float inf = std::numeric_limits<float>::infinity();
    out.assign((size_t)n, Box{{-inf, -inf, -inf}, {inf, inf, inf}});
    std::vector<Affine> aff((size_t)n);
    // R: twice the largest coordinate (1-norm over a point's three) of the scene's finite part, plus 1
    double reach = 0.0;
    for (int e = 0; e < n_extra; ++e) {
        const double s = std::fabs(extra_points[3 * e]) + std::fabs(extra_points[3 * e + 1]) + std::fabs(extra_points[3 * e + 2]);
        if (std::isfinite(s)) reach = std::fmax(reach, s);
    }
    for (int g = 0; g < n; ++g) {
        if (skip[g]) { aff[(size_t)g].ok = false; continue; }
        aff[(size_t)g] = analyse(inv16[g]);
        const Affine &f = aff[(size_t)g];
        if (!f.ok) continue;
        double s = 0.0;
        for (int i = 0; i < 3; ++i) s += std::fabs(f.c[i]) + 0.5 * (std::fabs(f.Ai[i][0]) + std::fabs(f.Ai[i][1]) + std::fabs(f.Ai[i][2]));
        if (std::isfinite(s)) reach = std::fmax(reach, s);
    }
    double R = 2.0 * reach + 1.0;
    // the kernel forms n = -o * (1/d) with 1/d clamped to +-2^100: |o| must stay below 2^27 for that product to be
    // finite (an axis-parallel ray from farther out would see -inf for both planes of a slab it runs inside and be
    // culled).  Scenes that reach beyond 2^27 keep this bound; their far rays are `wild` -- candidates of everything.
    if (!(R < 0x1p27)) R = 0x1p27;
    const double u = 0x1p-24;
    for (int g = 0; g < n; ++g) {
        const Affine &f = aff[(size_t)g];
        if (!f.ok) continue;
        double a[3], S[3], Smax = 0.0, N = 0.0;
        for (int k = 0; k < 3; ++k) {
            a[k] = std::fabs(f.A[k][0]) + std::fabs(f.A[k][1]) + std::fabs(f.A[k][2]);
            S[k] = a[k] * R + std::fabs(f.b[k]);
            Smax = std::fmax(Smax, S[k]);
            for (int j = 0; j < 3; ++j) N += f.Ai[k][j] * f.Ai[k][j];
        }
        N = std::sqrt(N);
        double p[3];
        for (int k = 0; k < 3; ++k) {
            p[k] = u * (2.0 * (0.5 + S[k]) + 3.0 * S[k] + (1.7320508075688772 * Smax + 2.0) * (3.0 * a[k] * N + 2.0));
            if (is_sphere[g]) p[k] += 32.0 * u * (3.0 * Smax * Smax + 1.0);
            p[k] *= 4.0;
        }
        Box bx;
        bool fin = true;
        for (int i = 0; i < 3; ++i) {
            double H = 0.0;
            for (int k = 0; k < 3; ++k) H += std::fabs(f.Ai[i][k]) * (0.5 + p[k]);
            H += 64.0 * u * (R + std::fabs(f.c[i]) + H);
            bx.lo[i] = round_down(f.c[i] - H);
            bx.hi[i] = round_up(f.c[i] + H);
            if (!std::isfinite(bx.lo[i]) || !std::isfinite(bx.hi[i])) fin = false;
        }
        if (fin) out[(size_t)g] = bx;
    }
    return round_down(R);
}

// The box as the kernel reads it: centre c = fl((lo + hi) / 2) and the smallest float half extent H with
// [c - H, c + H] containing [lo, hi] in exact arithmetic (c, lo, hi are floats: the differences are evaluated in
// binary64, bumped one binary64 ulp up in case they were inexact, and rounded up).  Infinite boxes: c = 0, H = +inf.
inline void centre_half(float lo, float hi, float &c, float &H) {
    if (!std::isfinite(lo) || !std::isfinite(hi)) { c = 0.0f; H = std::numeric_limits<float>::infinity(); return; }
    c = (float)(((double)lo + (double)hi) * 0.5);
    const double h = std::fmax((double)hi - (double)c, (double)c - (double)lo);
    H = round_up(std::nextafter(h, std::numeric_limits<double>::infinity()));
    if (!(H >= 0.0f)) H = std::numeric_limits<float>::infinity();
}

// The kernel's EXACT one-axis early miss for a CUBE (pt_kernels.hpp, cull_scene): with q_k = row k of the
// inverseTransform applied to the origin and v_k = the same row applied to the direction, both evaluated in the
// reference's own operation order ((m0 x + m1 y) + (m2 z + m3), glm mat4 * vec4), "|q_k| > 0.5 and q_k v_k > 0" means the
// origin lies beyond slab k and the ray heads away from it: both slab parameters of the axis are negative (their
// signs are those of the numerators -0.5 - q_k, 0.5 - q_k times that of q.direction[k] = v_k * (1 / sqrt(dot(v, v))),
// a positive factor), so tmax < 0 and boxIntersectionTest returns -1 whatever the other axes say
// (intersections.h:56-77).  Needs every entry of the matrix finite and bounded (then v is finite, dot(v, v) is neither
// NaN nor overflowing for the non-wild rays the kernel applies this to).  The row chosen is the one with the largest
// norm -- the thinnest world extent, i.e. the faces with the largest area: this is what takes a path's OWN wall out
// of its candidates.  Returns the mode: 0..2 = row k has no off-diagonal entries (q_k = fl(fl(m_kk o_k) + m_k3), two
// operations: the other products are exact zeros), 4 = general row (row[0..2], row[3] = translation), 3 = none.
inline int reject_row(const float *inv16, float row[4]) {
    row[0] = row[1] = row[2] = row[3] = 0.0f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r)
            if (!std::isfinite(inv16[c * 4 + r]) || std::fabs(inv16[c * 4 + r]) > 0x1p40f) return 3;
    int best = -1;
    double best_norm = 0.0;
    for (int k = 0; k < 3; ++k) {
        double nn = 0.0;
        for (int j = 0; j < 3; ++j) nn += (double)inv16[j * 4 + k] * (double)inv16[j * 4 + k];
        if (nn > best_norm) { best_norm = nn; best = k; }
    }
    if (best < 0) return 3;
    for (int j = 0; j < 4; ++j) row[j] = inv16[j * 4 + best];
    bool diag = row[best] != 0.0f;
    for (int j = 0; j < 3; ++j)
        if (j != best && row[j] != 0.0f) diag = false;
    return diag ? best : 4;
}

}  // namespace ptcull
