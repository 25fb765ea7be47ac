/*
 * rayrs_numeric.h -- the numerical contract of the rayrs MI355X hot path.
 *
 * The reference (Frojdholm/rayrs) draws its random numbers from rand 0.7.3's
 * OS-seeded thread RNG (19 call sites, e.g. rayrs-lib/src/lib.rs:206-207,
 * :539; material.rs:985-986, :1009-1011, :1144-1146, :379, :525, :579) and
 * calls the platform libm for sin/cos/ln/exp/acos/tan/atan2.  Neither is
 * reproducible across machines, so this build fixes both:
 *
 *   1. a counter-based RNG keyed by (seed, pixel, sample, draw index), and
 *   2. portable elementary functions whose every operation is an IEEE-754
 *      binary64 add/sub/mul/div/sqrt/fma, evaluated in a fixed order.
 *
 * The same header is compiled by gcc (CPU checker) and by hipcc for gfx950
 * (the kernel), both with -ffp-contract=off, so the two sides produce the
 * same bits for the same inputs.  Algorithms follow the published fdlibm /
 * musl formulations (Sun Microsystems, "Freely Distributable LIBM",
 * permission to use/copy/modify granted provided the notice is kept); fused
 * multiply-adds are written explicitly with rr_fma so that no compiler
 * decision can change a rounding.
 *
 * Accuracy (checked in tests/test_numeric.py against glibc): <= 2 ulp on the
 * domains the path tracer uses.
 */
#ifndef RAYRS_NUMERIC_H
#define RAYRS_NUMERIC_H

#include <stdint.h>

/* RR_FN: small helpers, always inlined.  RR_BIG: the elementary functions; in
 * device code they are real functions (not inlined into every call site), which
 * keeps the kernel's register allocation low enough for three waves per SIMD.
 * Inlining does not change any arithmetic. */
#if defined(__HIPCC__)
#define RR_FN __host__ __device__ static inline
#if defined(__HIP_DEVICE_COMPILE__)
#define RR_BIG __device__ static __attribute__((noinline))
#else
#define RR_BIG __host__ __device__ static inline
#endif
#else
#define RR_FN static inline
#define RR_BIG static inline
#endif

/* ------------------------------------------------------------------ bits */

RR_FN uint64_t rr_f64_bits(double x) {
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    return u;
}

RR_FN double rr_bits_f64(uint64_t u) {
    double x;
    __builtin_memcpy(&x, &u, 8);
    return x;
}

RR_FN double rr_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
RR_FN double rr_sqrt(double x) { return __builtin_sqrt(x); }
RR_FN double rr_fabs(double x) { return __builtin_fabs(x); }
RR_FN double rr_floor(double x) { return __builtin_floor(x); }
RR_FN double rr_ceil(double x) { return __builtin_ceil(x); }
/* Rust f64::max / f64::min: a NaN operand is ignored (IEEE maxNum/minNum). */
RR_FN double rr_max(double a, double b) { return __builtin_fmax(a, b); }
RR_FN double rr_min(double a, double b) { return __builtin_fmin(a, b); }
RR_FN int rr_isnan(double x) { return x != x; }

/* ------------------------------------------------------------------- RNG */

#define RR_GOLDEN 0x9E3779B97F4A7C15ULL

/* SplitMix64 finaliser (Steele, Lea, Flood 2014). */
RR_FN uint64_t rr_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* Key of one path: pixel = row * width + col in IMAGE coordinates (origin
 * upper left), sample = index of the path inside the pixel's spp loop
 * (rayrs/src/main.rs:68). */
RR_FN uint64_t rr_path_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    uint64_t h = rr_mix64(seed + RR_GOLDEN);
    h = rr_mix64(h ^ (pixel * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL));
    h = rr_mix64(h ^ (sample * 0xABC98388FB8FAC03ULL + 0x2545F4914F6CDD1DULL));
    return h;
}

/* The draw-th uniform of a path, in program order of the reference's
 * rand::random::<f64>() calls.  53 random bits -> [0, 1), the convention of
 * rand 0.7's Standard distribution for f64. */
RR_FN uint64_t rr_draw_bits(uint64_t key, uint32_t draw) {
    return rr_mix64(key + ((uint64_t)draw + 1ULL) * RR_GOLDEN);
}

RR_FN double rr_bits_to_u01(uint64_t bits) {
    return (double)(bits >> 11) * 0x1.0p-53;
}

RR_FN double rr_uniform(uint64_t key, uint32_t draw) {
    return rr_bits_to_u01(rr_draw_bits(key, draw));
}

/* -------------------------------------------------------- sin / cos / tan */

#define RR_PI 3.14159265358979323846264338327950288
#define RR_FRAC_1_PI 0.318309886183790671537767526745028724
#define RR_FRAC_PI_2 1.57079632679489661923132169163975144

/* kernel sin on |x| <= pi/4 with tail y (fdlibm k_sin.c) */
RR_FN double rr_ksin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = rr_fma(z, rr_fma(z, rr_fma(z, rr_fma(z, S6, S5), S4), S3), S2);
    /* x - ((z*(0.5*y - v*r) - y) - v*S1) */
    double t = rr_fma(-v, r, 0.5 * y);
    t = rr_fma(z, t, -y);
    t = rr_fma(-v, S1, t);
    return x - t;
}

/* kernel cos on |x| <= pi/4 with tail y (fdlibm k_cos.c, musl form) */
RR_FN double rr_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double w = z * z;
    double r = rr_fma(w * w, rr_fma(z, rr_fma(z, C6, C5), C4),
                      z * rr_fma(z, rr_fma(z, C3, C2), C1));
    double hz = 0.5 * z;
    double ww = 1.0 - hz;
    return ww + (((1.0 - ww) - hz) + rr_fma(z, r, -(x * y)));
}

/* Cody-Waite reduction x = k*pi/2 + (r + t), |r| <= pi/4 (+ rounding).
 * Exact for |x| <= 2*pi; adequate (error << 1 ulp of the result) up to
 * |x| ~ 1e5.  Returns k mod 4 in *q. */
RR_FN void rr_rem_pio2(double x, double* r, double* t, int* q) {
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632679489655800e+00;  /* 0x1.921fb54442d18p+0 */
    const double PIO2_LO = 6.12323399573676603587e-17;  /* pi/2 - PIO2_HI */
    double k = rr_floor(rr_fma(x, INV_PIO2, 0.5));
    double hi = rr_fma(-k, PIO2_HI, x);
    double rr = rr_fma(-k, PIO2_LO, hi);
    *t = rr_fma(-k, PIO2_LO, hi - rr);
    *r = rr;
    /* k is an integer of small magnitude */
    *q = ((int)k) & 3;
}

RR_BIG double rr_sin(double x) {
    double r, t;
    int q;
    rr_rem_pio2(x, &r, &t, &q);
    switch (q) {
        case 0: return rr_ksin(r, t);
        case 1: return rr_kcos(r, t);
        case 2: return -rr_ksin(r, t);
        default: return -rr_kcos(r, t);
    }
}

RR_BIG double rr_cos(double x) {
    double r, t;
    int q;
    rr_rem_pio2(x, &r, &t, &q);
    switch (q) {
        case 0: return rr_kcos(r, t);
        case 1: return -rr_ksin(r, t);
        case 2: return -rr_kcos(r, t);
        default: return rr_ksin(r, t);
    }
}

/* sin and cos of the same argument (one reduction).  Returned by value: through pointers the
 * results of this (non-inlined, on the device) function would travel through scratch memory. */
typedef struct {
    double s, c;
} rr_sincos_t;

RR_BIG rr_sincos_t rr_sincos(double x) {
    double r, t;
    int q;
    rr_rem_pio2(x, &r, &t, &q);
    double ks = rr_ksin(r, t);
    double kc = rr_kcos(r, t);
    rr_sincos_t out;
    switch (q) {
        case 0: out.s = ks; out.c = kc; break;
        case 1: out.s = kc; out.c = -ks; break;
        case 2: out.s = -ks; out.c = -kc; break;
        default: out.s = -kc; out.c = ks; break;
    }
    return out;
}

/* tan as a quotient of the two kernels.  tan(acos(0)) = tan(0x1.921fb54442d18p+0)
 * is finite (1.633e16), as it is for the reference's libm. */
RR_BIG double rr_tan(double x) {
    double r, t;
    int q;
    rr_rem_pio2(x, &r, &t, &q);
    double ks = rr_ksin(r, t);
    double kc = rr_kcos(r, t);
    return (q & 1) ? -(kc / ks) : (ks / kc);
}

/* ------------------------------------------------------------------- log */

RR_BIG double rr_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = rr_f64_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -1.0 / (x * x); /* log(+-0) = -inf */
        if (hx >> 31) return (x - x) / 0.0;       /* log(-#) = NaN */
        k -= 54;                                  /* subnormal: scale up */
        x *= 0x1p54;
        u = rr_f64_bits(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffULL);
    x = rr_bits_f64(u);

    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * rr_fma(w, rr_fma(w, Lg6, Lg4), Lg2);
    double t2 = z * rr_fma(w, rr_fma(w, rr_fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double dk = (double)k;
    return rr_fma(dk, ln2_hi, (rr_fma(dk, ln2_lo, s * (hfsq + R)) - hfsq) + f);
}

/* ------------------------------------------------------------------- exp */

RR_BIG double rr_exp(double x) {
    const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                 P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    uint64_t u = rr_f64_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int sign = (int)(hx >> 31);
    hx &= 0x7fffffffu;
    double hi, lo;
    int k;

    if (hx >= 0x4086232bu) { /* |x| >= 708.39 or NaN */
        if (rr_isnan(x)) return x;
        if (x > 709.782712893383973096) return x * 0x1p1023; /* overflow -> inf */
        if (x < -745.13321910194110842) return 0.0;
    }
    if (hx > 0x3fd62e42u) { /* |x| > 0.5 ln2 */
        if (hx >= 0x3ff0a2b2u) /* |x| >= 1.5 ln2 */
            k = (int)(invln2 * x + (sign ? -0.5 : 0.5));
        else
            k = 1 - sign - sign;
        hi = rr_fma(-(double)k, ln2hi, x);
        lo = (double)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x3e300000u) { /* |x| > 2**-28 */
        k = 0;
        hi = x;
        lo = 0.0;
    } else {
        return 1.0 + x;
    }
    double xx = x * x;
    double c = x - xx * rr_fma(xx, rr_fma(xx, rr_fma(xx, rr_fma(xx, P5, P4), P3), P2), P1);
    double y = 1.0 + ((x * c / (2.0 - c) - lo) + hi);
    if (k == 0) return y;
    /* y * 2**k with a single rounding, also into the subnormal range */
    if (k > 1000) {
        y *= 0x1p1000;
        k -= 1000;
    } else if (k < -1000) {
        double big = rr_bits_f64((uint64_t)(0x3ff + (k + 1000)) << 52);
        return (y * big) * 0x1p-1000;
    }
    return y * rr_bits_f64((uint64_t)(0x3ff + k) << 52);
}

/* ------------------------------------------------------------------ acos */

RR_FN double rr_acos_R(double z) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double p = z * rr_fma(z, rr_fma(z, rr_fma(z, rr_fma(z, rr_fma(z, pS5, pS4), pS3), pS2), pS1), pS0);
    double q = rr_fma(z, rr_fma(z, rr_fma(z, rr_fma(z, qS4, qS3), qS2), qS1), 1.0);
    return p / q;
}

RR_BIG double rr_acos(double x) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    uint64_t u = rr_f64_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    uint32_t ix = hx & 0x7fffffffu;
    if (ix >= 0x3ff00000u) { /* |x| >= 1 or NaN */
        uint32_t lx = (uint32_t)u;
        if (((ix - 0x3ff00000u) | lx) == 0) {
            if (hx >> 31) return 2.0 * pio2_hi + 0x1p-120;
            return 0.0;
        }
        return 0.0 / (x - x); /* NaN */
    }
    if (ix < 0x3fe00000u) { /* |x| < 0.5 */
        if (ix <= 0x3c600000u) return pio2_hi + 0x1p-120; /* |x| < 2**-57 */
        return pio2_hi - (x - (pio2_lo - x * rr_acos_R(x * x)));
    }
    if (hx >> 31) { /* x < -0.5 */
        double z = (1.0 + x) * 0.5;
        double s = rr_sqrt(z);
        double w = rr_acos_R(z) * s - pio2_lo;
        return 2.0 * (pio2_hi - (s + w));
    }
    /* x > 0.5 */
    double z = (1.0 - x) * 0.5;
    double s = rr_sqrt(z);
    double df = rr_bits_f64(rr_f64_bits(s) & 0xffffffff00000000ULL);
    double c = (z - df * df) / (s + df);
    double w = rr_acos_R(z) * s + c;
    return 2.0 * (df + w);
}

/* ----------------------------------------------------------- atan / atan2 */

RR_FN double rr_atan_inl(double x) {
    const double atanhi0 = 4.63647609000806093515e-01, atanhi1 = 7.85398163397448278999e-01,
                 atanhi2 = 9.82793723247329054082e-01, atanhi3 = 1.57079632679489655800e+00;
    const double atanlo0 = 2.26987774529616870924e-17, atanlo1 = 3.06161699786838301793e-17,
                 atanlo2 = 1.39033110312309984516e-17, atanlo3 = 6.12323399573676603587e-17;
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    uint64_t u = rr_f64_bits(x);
    uint32_t ix = (uint32_t)(u >> 32);
    uint32_t sign = ix >> 31;
    ix &= 0x7fffffffu;
    int id;
    double hi = 0.0, lo = 0.0;
    if (ix >= 0x44100000u) { /* |x| >= 2**66 or NaN */
        if (rr_isnan(x)) return x;
        double z = atanhi3 + 0x1p-120;
        return sign ? -z : z;
    }
    if (ix < 0x3fdc0000u) { /* |x| < 0.4375 */
        if (ix < 0x3e400000u) return x; /* |x| < 2**-27 */
        id = -1;
    } else {
        x = rr_fabs(x);
        if (ix < 0x3ff30000u) {     /* |x| < 1.1875 */
            if (ix < 0x3fe60000u) { /* 7/16 <= |x| < 11/16 */
                id = 0; hi = atanhi0; lo = atanlo0;
                x = (2.0 * x - 1.0) / (2.0 + x);
            } else { /* 11/16 <= |x| < 19/16 */
                id = 1; hi = atanhi1; lo = atanlo1;
                x = (x - 1.0) / (x + 1.0);
            }
        } else {
            if (ix < 0x40038000u) { /* |x| < 2.4375 */
                id = 2; hi = atanhi2; lo = atanlo2;
                x = (x - 1.5) / (1.0 + 1.5 * x);
            } else { /* 2.4375 <= |x| < 2**66 */
                id = 3; hi = atanhi3; lo = atanlo3;
                x = -1.0 / x;
            }
        }
    }
    double z = x * x;
    double w = z * z;
    double s1 = z * rr_fma(w, rr_fma(w, rr_fma(w, rr_fma(w, rr_fma(w, aT10, aT8), aT6), aT4), aT2), aT0);
    double s2 = w * rr_fma(w, rr_fma(w, rr_fma(w, rr_fma(w, aT9, aT7), aT5), aT3), aT1);
    if (id < 0) return x - x * (s1 + s2);
    z = hi - ((x * (s1 + s2) - lo) - x);
    return sign ? -z : z;
}

/* the same as a real function for callers that want one (a nested call from rr_atan2 would make
 * that function spill its return address to scratch on the device) */
RR_BIG double rr_atan(double x) { return rr_atan_inl(x); }

RR_BIG double rr_atan2(double y, double x) {
    const double pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
    if (rr_isnan(x) || rr_isnan(y)) return x + y;
    uint64_t ux = rr_f64_bits(x), uy = rr_f64_bits(y);
    uint32_t ix = (uint32_t)(ux >> 32), lx = (uint32_t)ux;
    uint32_t iy = (uint32_t)(uy >> 32), ly = (uint32_t)uy;
    if (((ix - 0x3ff00000u) | lx) == 0) return rr_atan_inl(y); /* x = 1.0 */
    uint32_t m = ((iy >> 31) & 1u) | ((ix >> 30) & 2u);    /* 2*sign(x)+sign(y) */
    ix &= 0x7fffffffu;
    iy &= 0x7fffffffu;
    if ((iy | ly) == 0) { /* y = 0 */
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi;
            default: return -pi;
        }
    }
    if ((ix | lx) == 0) return (m & 1) ? -pi / 2 : pi / 2; /* x = 0 */
    if (ix == 0x7ff00000u) {                               /* x = inf */
        if (iy == 0x7ff00000u) {
            switch (m) {
                case 0: return pi / 4;
                case 1: return -pi / 4;
                case 2: return 3 * pi / 4;
                default: return -3 * pi / 4;
            }
        } else {
            switch (m) {
                case 0: return 0.0;
                case 1: return -0.0;
                case 2: return pi;
                default: return -pi;
            }
        }
    }
    /* |y/x| > 2**64 or y = inf */
    if (ix + (64u << 20) < iy || iy == 0x7ff00000u) return (m & 1) ? -pi / 2 : pi / 2;
    double z;
    if ((m & 2) && iy + (64u << 20) < ix) /* |y/x| < 2**-64, x < 0 */
        z = 0.0;
    else
        z = rr_atan_inl(rr_fabs(y / x));
    switch (m) {
        case 0: return z;
        case 1: return -z;
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

#endif /* RAYRS_NUMERIC_H */
