/* Exhaustive proof that the fused (fma) evaluation of the shared sin/cos used by the HIP kernels
 * (csrc/pt_device.hpp: sincos_shared) returns the same two binary32 values as its definition
 * (oracle/ptoracle.c: pto_sincos -- every product and sum rounded separately) for EVERY float in [0, 6.3],
 * the whole domain of the callers (u01 * 2 pi).  Test infrastructure: built and run by
 * tests/test_oracle_golden.py::test_fused_sincos_equals_the_definition (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 * Prints: "<mismatching arguments> <arguments whose binary64 results differ> <arguments checked>". */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static const double TWO_OVER_PI = 6.36619772367581382433e-01, PIO2_1 = 1.57079632673412561417e+00,
    PIO2_1T = 6.07710050650619224932e-11, MAGIC = 6755399441055744.0,
    S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
    S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
    C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
    C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
static inline void finish(double kd, double sn, double cs, double *so, double *co) {
    int q = (int)kd & 3;
    *so = (q & 1) ? cs : sn; *co = (q & 1) ? sn : cs;
    *so = (q & 2) ? -*so : *so; *co = ((q + 1) & 2) ? -*co : *co;
}
static inline void definition(float x, double *so, double *co) {
    double xd = (double)x;
    double kd = (xd * TWO_OVER_PI + MAGIC) - MAGIC;
    double r = (xd - kd * PIO2_1) - kd * PIO2_1T;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + (r * z) * ps;
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    finish(kd, sn, cs, so, co);
}
static inline void fused(float x, double *so, double *co) {
    double xd = (double)x;
    double kd = fma(xd, TWO_OVER_PI, MAGIC) - MAGIC;
    double r = fma(-kd, PIO2_1T, fma(-kd, PIO2_1, xd));
    double z = r * r;
    double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
    double sn = fma(r * z, ps, r);
    double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    finish(kd, sn, cs, so, co);
}
int main(void) {
    const float hi = 6.3f;
    uint32_t hib; memcpy(&hib, &hi, 4);
    long long bad = 0, dd = 0;
#pragma omp parallel for reduction(+:bad,dd) schedule(static)
    for (int64_t b = 0; b <= (int64_t)hib; ++b) {
        uint32_t u = (uint32_t)b; float x; memcpy(&x, &u, 4);
        double s0, c0, s1, c1;
        definition(x, &s0, &c0); fused(x, &s1, &c1);
        const float fs0 = (float)s0, fc0 = (float)c0, fs1 = (float)s1, fc1 = (float)c1;
        if (memcmp(&fs0, &fs1, 4) || memcmp(&fc0, &fc1, 4)) bad++;
        if (memcmp(&s0, &s1, 8) || memcmp(&c0, &c1, 8)) dd++;
    }
    printf("%lld %lld %u\n", bad, dd, hib + 1);
    return bad != 0;
}
