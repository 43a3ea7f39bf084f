/* pt_detmath.h — transcendental functions with ONE definition on x86 and gfx950.
 *
 * The reference calls libm/libdevice for logf/sinf/cosf (inside curand_normal,
 * src/random.cu:11-13), atan2f/acosf (src/mathops.h:712-720). None of those is
 * bit-portable across CPU and GPU, and the Box–Muller jitter feeds the primary
 * ray (src/renderer.cu:379-384), so hit-ID parity between the CPU oracle and the
 * HIP kernel needs a shared definition. These are cephes-style single-precision
 * kernels written with IEEE +,-,*,/ and sqrt only (no fma, no table lookups):
 * compiled with -ffp-contract=off on both sides they return the same bits.
 * Accuracy vs glibc is checked in tests/test_detmath.py (a few ulp).
 *
 * This header is part of the boundary contract: the oracle and the kernel both
 * include it; everything else in the oracle is written independently.
 */
#ifndef PT_DETMATH_H_
#define PT_DETMATH_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD static inline
#endif

PT_HD float pt_bits_to_float(uint32_t u) {
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}
PT_HD uint32_t pt_float_to_bits(float f) {
    union { uint32_t u; float f; } c;
    c.f = f;
    return c.u;
}

#define PT_PI_F 3.14159265358979323846f
#define PT_1_PI_F 0.318309886183790671538f
#define PT_PI_D 3.14159265358979323846

/* Natural log for x in (0, +inf), normal or subnormal. */
PT_HD float pt_logf(float x) {
    uint32_t ix = pt_float_to_bits(x);
    int e = 0;
    if (ix < 0x00800000u) { /* subnormal: scale by 2^23 */
        x = x * 8388608.0f;
        ix = pt_float_to_bits(x);
        e = -23;
    }
    e += (int)(ix >> 23) - 126; /* x = m * 2^e, m in [0.5, 1) */
    float m = pt_bits_to_float((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = y * m + -1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m + -1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m + -1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m + -2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    r = r + 0.693359375f * fe;
    return r;
}

/* sin and cos for x in [0, 8] (Box–Muller angles lie in (0, 2*pi]). */
PT_HD void pt_sincosf(float x, float *s, float *c) {
    /* j = nearest multiple of pi/2 */
    int j = (int)(x * 0.636619772367581343f + 0.5f);
    float fj = (float)j;
    /* Cody–Waite: pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188216e-8 */
    float r = ((x - fj * 1.5703125f) - fj * 4.837512969970703125e-4f) - fj * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z + -1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z + -1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    switch (j & 3) {
    case 0: *s = sp;  *c = cp;  break;
    case 1: *s = cp;  *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}

/* atan for any finite x. */
PT_HD float pt_atanf(float xx) {
    float x = xx < 0.0f ? -xx : xx;
    float y;
    if (x > 2.414213562373095f) { /* tan(3pi/8) */
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return xx < 0.0f ? -y : y;
}

/* atan2(y, x), result in [-pi, pi]; (0,0) -> 0. */
PT_HD float pt_atan2f(float y, float x) {
    if (x == 0.0f) {
        if (y > 0.0f) return 1.5707963267948966f;
        if (y < 0.0f) return -1.5707963267948966f;
        return 0.0f;
    }
    float a = pt_atanf(y / x);
    if (x < 0.0f) {
        a = (y < 0.0f) ? a - PT_PI_F : a + PT_PI_F;
    }
    return a;
}

#if defined(__HIPCC__)
#define PT_SQRTF(x) __builtin_sqrtf(x)
#else
#define PT_SQRTF(x) __builtin_sqrtf(x)
#endif

/* acos for x in [-1, 1]. */
PT_HD float pt_acosf(float x) {
    float a = x < 0.0f ? -x : x;
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = PT_SQRTF(z);
        float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
                   + 1.6666752422e-1f) * z * s + s;
        float r = p + p; /* acos(|x|) */
        return x < 0.0f ? PT_PI_F - r : r;
    }
    float z = x * x;
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f) * z * x + x;
    return 1.5707963267948966f - p;
}

#endif /* PT_DETMATH_H_ */
