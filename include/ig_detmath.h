/* ig_detmath.h — a small deterministic float32 math library.
 *
 * The reference's stdlib calls the platform libm (math_builtins::sin/cos/acos,
 * src/artic/core/sampling.art:12-20, src/artic/light/area.art:160-181) and is
 * compiled with -ffast-math, so its transcendental results are
 * platform-defined. A path tracer amplifies 1-ulp differences into different
 * paths, so to compare the HIP device against the CPU oracle at 1e-4 relative
 * L2 both must round identically. Every function here is built only from
 * IEEE-754 correctly rounded operations (+ - * / sqrt fma, conversions, bit
 * operations), so gcc on x86-64 (with -mfma) and hipcc on gfx950 (whose
 * default fp32 divide/sqrt are correctly rounded and which keeps denormals)
 * produce bit-identical results. Compile every user with -ffp-contract=off.
 *
 * This header plays the role of libm for BOTH the product (HIP kernels) and
 * the test oracle; it holds no rendering algorithm.
 *
 * Polynomials: Cephes single-precision sinf/cosf/asinf/expf/logf (public domain, S. Moshier).
 */
#ifndef IG_DETMATH_H
#define IG_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define IGM_FN __host__ __device__ static inline
#else
#define IGM_FN static inline
#endif

#define IGM_PI 3.14159265359f      /* flt_pi, src/artic/core/common.art:7 */
#define IGM_INV_PI 0.31830988618379067154f
#define IGM_FLT_EPS 1.1920928955e-07f
#define IGM_FLT_MAX 3.4028234664e+38f

IGM_FN uint32_t igm_bits(float f)
{
    union {
        float f;
        uint32_t u;
    } c;
    c.f = f;
    return c.u;
}

IGM_FN float igm_float(uint32_t u)
{
    union {
        float f;
        uint32_t u;
    } c;
    c.u = u;
    return c.f;
}

IGM_FN float igm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
IGM_FN float igm_sqrt(float a) { return __builtin_sqrtf(a); }
IGM_FN float igm_abs(float a) { return igm_float(igm_bits(a) & 0x7FFFFFFFu); }
IGM_FN float igm_copysign(float mag, float sgn) { return igm_float((igm_bits(mag) & 0x7FFFFFFFu) | (igm_bits(sgn) & 0x80000000u)); }
IGM_FN int igm_signbit(float a) { return (igm_bits(a) >> 31) != 0; }
IGM_FN float igm_rint(float a) { return __builtin_rintf(a); }
IGM_FN float igm_floor(float a) { return __builtin_floorf(a); }

/* IEEE minNum / maxNum with -0 < +0 (what v_min_f32 / v_max_f32 compute). */
IGM_FN float igm_min(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fminf(a, b);
#else
    if (a != a)
        return b;
    if (b != b)
        return a;
    if (a == b)
        return igm_float(igm_bits(a) | igm_bits(b));
    return a < b ? a : b;
#endif
}

IGM_FN float igm_max(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(a, b);
#else
    if (a != a)
        return b;
    if (b != b)
        return a;
    if (a == b)
        return igm_float(igm_bits(a) & igm_bits(b));
    return a > b ? a : b;
#endif
}

IGM_FN float igm_clamp(float v, float l, float u) { return igm_min(u, igm_max(l, v)); } /* clampf, common.art:261 */

/* ---- sin / cos ------------------------------------------------------------
 * Cody-Waite reduction by pi/2 in three parts (products exact under fma),
 * Cephes minimax polynomials on [-pi/4, pi/4]. Valid for |x| < ~1e5, more
 * than the path needs (arguments are 2*pi*u and spherical-rectangle angles). */
#define IGM_PIO2_1 1.5703125f
#define IGM_PIO2_2 4.837512969970703125e-4f
#define IGM_PIO2_3 7.549789954891882e-8f

IGM_FN float igm_sin_poly(float r)
{
    const float z = r * r;
    float p       = igm_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p             = igm_fma(p, z, -1.6666654611e-1f);
    return igm_fma(p * z, r, r);
}

IGM_FN float igm_cos_poly(float r)
{
    const float z = r * r;
    float p       = igm_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p             = igm_fma(p, z, 4.166664568298827e-2f);
    return igm_fma(p, z * z, igm_fma(-0.5f, z, 1.0f));
}

IGM_FN float igm_reduce_pio2(float x, int* quadrant)
{
    const float fn = igm_rint(x * 0.63661977236758134308f);
    float r        = igm_fma(-fn, IGM_PIO2_1, x);
    r              = igm_fma(-fn, IGM_PIO2_2, r);
    r              = igm_fma(-fn, IGM_PIO2_3, r);
    *quadrant      = (int)fn & 3;
    return r;
}

IGM_FN float igm_sin(float x)
{
    int q;
    const float r = igm_reduce_pio2(x, &q);
    const float s = (q & 1) ? igm_cos_poly(r) : igm_sin_poly(r);
    return (q & 2) ? -s : s;
}

IGM_FN float igm_cos(float x)
{
    int q;
    const float r = igm_reduce_pio2(x, &q);
    const float c = (q & 1) ? igm_sin_poly(r) : igm_cos_poly(r);
    return ((q + 1) & 2) ? -c : c;
}

/* ---- asin / acos (Cephes asinf / acosf), |x| <= 1 -------------------------- */
IGM_FN float igm_asin(float xx)
{
    const float a = igm_abs(xx);
    float x, z;
    int flag = 0;
    if (a > 0.5f) {
        z    = 0.5f * (1.0f - a);
        x    = igm_sqrt(z);
        flag = 1;
    } else {
        x = a;
        z = x * x;
    }
    float p = igm_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
    p       = igm_fma(p, z, 4.5470025998e-2f);
    p       = igm_fma(p, z, 7.4953002686e-2f);
    p       = igm_fma(p, z, 1.6666752422e-1f);
    float r = igm_fma(p * z, x, x);
    if (flag) {
        r = r + r;
        r = 1.5707963267948966192f - r;
    }
    return igm_copysign(r, xx);
}

IGM_FN float igm_acos(float x)
{
    if (x < -0.5f)
        return 3.14159265358979323846f - 2.0f * igm_asin(igm_sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f)
        return 2.0f * igm_asin(igm_sqrt(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - igm_asin(x);
}

/* ---- atan / atan2 (Cephes atanf): reduction to [0, tan(pi/8)] and a degree-4 minimax polynomial in z^2 ---- */
IGM_FN float igm_atan_pos(float x) /* x >= 0 */
{
    float y, z;
    if (x > 2.414213562373095f) { /* tan(3 pi / 8) */
        y = 1.5707963267948966192f;
        z = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi / 8) */
        y = 0.7853981633974483096f;
        z = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
        z = x;
    }
    const float w = z * z;
    float p       = igm_fma(8.05374449538e-2f, w, -1.38776856032e-1f);
    p             = igm_fma(p, w, 1.99777106478e-1f);
    p             = igm_fma(p, w, -3.33329491539e-1f);
    return y + igm_fma(p * w, z, z);
}

IGM_FN float igm_atan2(float y, float x)
{
    const float ax = igm_abs(x), ay = igm_abs(y);
    float r;
    if (ax == 0.0f && ay == 0.0f)
        r = 0.0f;
    else if (ax == 0.0f)
        r = 1.5707963267948966192f;
    else
        r = igm_atan_pos(ay / ax);
    if (igm_signbit(x))
        r = 3.14159265358979323846f - r;
    return igm_copysign(r, y);
}

/* ---- exp (Cephes expf): x = n ln2 + r, |r| <= ln2 / 2, degree-5 polynomial, scaling by 2^n through the exponent ---- */
IGM_FN float igm_exp(float x)
{
    if (x > 88.72283905206835f)
        return igm_float(0x7F800000u); /* +inf */
    if (x < -87.33654475055310898657f)
        return 0.0f;
    const float fn = igm_floor(igm_fma(x, 1.44269504088896341f, 0.5f));
    float r        = igm_fma(-fn, 0.693359375f, x);
    r              = igm_fma(-fn, -2.12194440e-4f, r);
    const float z  = r * r;
    float p        = igm_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
    p              = igm_fma(p, r, 8.3334519073e-3f);
    p              = igm_fma(p, r, 4.1665795894e-2f);
    p              = igm_fma(p, r, 1.6666665459e-1f);
    p              = igm_fma(p, r, 5.0000001201e-1f);
    const float e  = igm_fma(p, z, r) + 1.0f;
    const int n    = (int)fn; /* -126 <= n <= 128 here */
    /* 2^n in two factors so that n = 128 and the subnormal end stay representable */
    const int h    = n / 2;
    return (e * igm_float((uint32_t)(h + 127) << 23)) * igm_float((uint32_t)(n - h + 127) << 23);
}

/* ---- log (Cephes logf): x = m 2^e with sqrt(1/2) <= m < sqrt(2), degree-9 polynomial in m - 1 ---- */
IGM_FN float igm_log(float x)
{
    if (x != x || x < 0.0f)
        return igm_float(0x7FC00000u); /* NaN */
    if (x == 0.0f)
        return igm_float(0xFF800000u); /* -inf */
    if (x == igm_float(0x7F800000u))
        return x;
    int e = 0;
    uint32_t b = igm_bits(x);
    if ((b & 0x7F800000u) == 0) { /* subnormal: scale by 2^23 first */
        x = x * 8388608.0f;
        b = igm_bits(x);
        e = -23;
    }
    e += (int)(b >> 23) - 126;                               /* frexp: m in [0.5, 1) */
    float m = igm_float((b & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p       = igm_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
    p             = igm_fma(p, m, 1.1676998740e-1f);
    p             = igm_fma(p, m, -1.2420140846e-1f);
    p             = igm_fma(p, m, 1.4249322787e-1f);
    p             = igm_fma(p, m, -1.6668057665e-1f);
    p             = igm_fma(p, m, 2.0000714765e-1f);
    p             = igm_fma(p, m, -2.4999993993e-1f);
    p             = igm_fma(p, m, 3.3333331174e-1f);
    const float fe = (float)e;
    float y        = (p * m) * z;
    y              = igm_fma(-2.12194440e-4f, fe, y);
    y              = igm_fma(-0.5f, z, y);
    return igm_fma(0.693359375f, fe, m + y);
}

/* x^p for x > 0 as exp(p log x) (relative error ~ |p log x| * 1e-7: for display values, not for sampling) */
IGM_FN float igm_pow(float x, float p) { return x <= 0.0f ? 0.0f : igm_exp(p * igm_log(x)); }

#endif /* IG_DETMATH_H */
