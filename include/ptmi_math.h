/* ptmi_math.h — the numerics contract shared by the HIP kernels, the host code
 * and the CPU oracle.
 *
 * Why this exists: the reference calls cosf/sinf/powf/tan from whatever libm
 * the toolchain links (CUDA libdevice on the GPU, MSVC/glibc on the host;
 * integrator.h:68-70, integrator.h:398-400, sensor.h:42,57-63).  Those differ
 * in the last ulp between vendors, CPUs (glibc ifunc/FMA variants) and
 * devices, and a one-ulp change in a bounce direction can flip a hit/miss
 * decision.  To make "same scene + same seed => same bits" hold between the
 * CPU oracle and the gfx950 kernels, every transcendental on the path is
 * evaluated here in binary64 with nothing but IEEE + - * / (no fma, compile
 * with -ffp-contract=off) and rounded ONCE to binary32.  The binary64 result
 * is within ~1e-16 relative of the true value, so the rounded float equals the
 * correctly rounded float except when the true value lies within ~1e-9 ulp of
 * a rounding boundary; a good libm agrees with it on all but those inputs
 * where the libm itself is not correctly rounded.
 *
 * Polynomial coefficients for sin/cos are the classic fdlibm k_sin/k_cos
 * minimax sets (|x| <= pi/4, error < 2^-57).
 */
#ifndef PTMI_MATH_H
#define PTMI_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PTMI_HD __host__ __device__ __forceinline__
#elif defined(__cplusplus)
#define PTMI_HD static inline
#else
#define PTMI_HD static inline
#endif

/* (double)M_PI as glibc/<cmath> defines it (sensor.h:7-9 falls back to the same
 * literal); the reference mixes it into float expressions, which therefore
 * evaluate in binary64 (integrator.h:67, sensor.h:41). */
#define PTMI_PI_D 3.14159265358979323846

PTMI_HD uint64_t ptmi_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
PTMI_HD double ptmi_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

/* floor(x + 0.5) for |x| < 2^31 without calling libm. */
PTMI_HD int ptmi_round_half_up(double x) {
    double y = x + 0.5;
    int k = (int)y;          /* truncation toward zero */
    if ((double)k > y) k -= 1; /* fix up negatives */
    return k;
}

/* sin and cos of x (radians) in binary64, |x| <~ 1e5.
 * Cody-Waite reduction by pi/2 with a 33-bit head so k*head is exact. */
PTMI_HD void ptmi_sincos_d(double x, double* s_out, double* c_out) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1   = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double PIO2_1T  = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
    int k = ptmi_round_half_up(x * INV_PIO2);
    double kd = (double)k;
    double r = (x - kd * PIO2_1) - kd * PIO2_1T;
    double z = r * r;
    /* k_sin */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = S6;
    ps = ps * z + S5; ps = ps * z + S4; ps = ps * z + S3; ps = ps * z + S2; ps = ps * z + S1;
    double sr = r + (r * z) * ps;
    /* k_cos */
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = C6;
    pc = pc * z + C5; pc = pc * z + C4; pc = pc * z + C3; pc = pc * z + C2; pc = pc * z + C1;
    double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    double s, c;
    switch (k & 3) {
        case 0:  s = sr;  c = cr;  break;
        case 1:  s = cr;  c = -sr; break;
        case 2:  s = -sr; c = -cr; break;
        default: s = -cr; c = sr;  break;
    }
    *s_out = s; *c_out = c;
}

/* cosf(x)/sinf(x) replacement (integrator.h:68-69, sensor.h:61-63). */
PTMI_HD void ptmi_sincosf(float x, float* s, float* c) {
    double sd, cd;
    ptmi_sincos_d((double)x, &sd, &cd);
    *s = (float)sd; *c = (float)cd;
}

/* tan(double) replacement for the camera (sensor.h:42). */
PTMI_HD double ptmi_tan_d(double x) {
    double s, c;
    ptmi_sincos_d(x, &s, &c);
    return s / c;
}

/* natural log of a positive finite double */
PTMI_HD double ptmi_log_d(double x) {
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    uint64_t b = ptmi_d2u(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = ptmi_u2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    if (m > 1.41421356237309514547) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 25.0;
    p = p * z + 1.0 / 23.0; p = p * z + 1.0 / 21.0; p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0; p = p * z + 1.0 / 9.0;  p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;  p = p * z + 1.0 / 3.0;
    double lm = 2.0 * s + (2.0 * s) * (z * p);
    double ed = (double)e;
    return ed * LN2_HI + (lm + ed * LN2_LO);
}

/* exp of a double in [-700, 700] */
PTMI_HD double ptmi_exp_d(double x) {
    const double INV_LN2 = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    int k = ptmi_round_half_up(x * INV_LN2);
    double kd = (double)k;
    double r = (x - kd * LN2_HI) - kd * LN2_LO;
    double p = 1.0 / 87178291200.0;          /* 1/14! */
    p = p * r + 1.0 / 6227020800.0;          /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    double scale = ptmi_u2d((uint64_t)(k + 1023) << 52);
    return p * scale;
}

/* powf(x, y) replacement for the gamma step (integrator.h:397-400): x in [0,1]. */
PTMI_HD float ptmi_powf(float x, float y) {
    if (x != x) return x;
    if (!(x > 0.0f)) return 0.0f;
    double l = ptmi_log_d((double)x);
    return (float)ptmi_exp_d((double)y * l);
}

/* expf(x) replacement for the grid filters' Gaussian weights (grid_filter.h:35-37): any float x.
 * exp(-104) < 2^-150 (half the smallest denormal), exp(88.73) > FLT_MAX. */
PTMI_HD float ptmi_expf(float x) {
    if (x != x) return x;
    if (x < -104.0f) return 0.0f;
    if (x > 88.75f) return __builtin_inff();
    return (float)ptmi_exp_d((double)x);
}

/* ---- atan / atan2 / acos in binary64 (grid.h:307-308 worldToSpherical uses acosf / atan2f) ------------------
 * fdlibm s_atan.c reduction and minimax polynomial (error < 1 ulp of binary64), IEEE binary64 sqrt for acos;
 * as everywhere in this file: no fma, results rounded once to binary32 by the float wrappers. */
PTMI_HD double ptmi_atan_d(double x) {
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01, aT2 = 1.42857142725034663711e-01,
                 aT3 = -1.11111104054623557880e-01, aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02, aT8 = 4.97687799461593236017e-02,
                 aT9 = -3.65315727442169155270e-02, aT10 = 1.62858201153657823623e-02;
    const int neg = x < 0.0;
    double ax = neg ? -x : x;
    double hi = 0.0, lo = 0.0;
    int id = -1;
    if (ax >= 0.4375) {
        if (ax < 0.6875)      { id = 0; ax = (2.0 * ax - 1.0) / (2.0 + ax); hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
        else if (ax < 1.1875) { id = 1; ax = (ax - 1.0) / (ax + 1.0);       hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
        else if (ax < 2.4375) { id = 2; ax = (ax - 1.5) / (1.0 + 1.5 * ax); hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
        else                  { id = 3; ax = -1.0 / ax;                     hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
    }
    const double z = ax * ax, w = z * z;
    const double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    double r;
    if (id < 0) r = ax - ax * (s1 + s2);
    else r = hi - ((ax * (s1 + s2) - lo) - ax);
    return neg ? -r : r;
}
PTMI_HD double ptmi_atan2_d(double y, double x) {
    const double PI = 3.14159265358979311600e+00, PI_LO = 1.2246467991473531772e-16;
    if (x != x || y != y) return x + y;
    if (y == 0.0) return (x > 0.0 || (x == 0.0 && !(ptmi_d2u(x) >> 63))) ? y : ((ptmi_d2u(y) >> 63) ? -PI : PI);
    if (x == 0.0) return y < 0.0 ? -0.5 * PI : 0.5 * PI;
    double z = ptmi_atan_d((y < 0.0 ? -y : y) / (x < 0.0 ? -x : x));
    if (x > 0.0) return y < 0.0 ? -z : z;
    z = PI - (z - PI_LO);
    return y < 0.0 ? -z : z;
}
PTMI_HD float ptmi_atan2f(float y, float x) { return (float)ptmi_atan2_d((double)y, (double)x); }
/* acosf for |x| <= 1 (callers clamp, grid.h:307): acos x = atan2(sqrt((1 - x)(1 + x)), x); both factors are exact
 * in binary64 for a binary32 x */
PTMI_HD float ptmi_acosf(float x) {
    const double xd = (double)x;
    return (float)ptmi_atan2_d(__builtin_sqrt((1.0 - xd) * (1.0 + xd)), xd);
}

#endif /* PTMI_MATH_H */
