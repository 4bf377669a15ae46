// rt_pmath_oracle.h -- the oracle's statement of the portable elementary functions.
//
// TEST INFRASTRUCTURE (see rt_oracle.cpp).  The path tracer's sin/cos/asin/acos/atan2/log
// are library calls in the reference (CUDA libm), whose last-bit behaviour is not part of
// the reference source.  Product and oracle therefore share one SPECIFICATION of these
// functions (IEEE + - * / sqrt only, fixed order, Cephes single-precision polynomials) so
// that whole images can be compared bit for bit; this file is the oracle's own copy of
// that specification, kept separate from ray-tracer_amd/csrc/rt_pmath.h on purpose.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#define RT_PM_FN static inline

namespace pm {

RT_PM_FN uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
RT_PM_FN float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// sin and cos of x (|x| < ~8000 keeps full accuracy; larger arguments lose bits, as
// single-precision libraries without Payne-Hanek do).
RT_PM_FN void sincos(float x, float* s_out, float* c_out) {
    const float ax = fabsf(x);
    // quadrant count in units of pi/4, made even (Cephes: j odd -> j+1)
    int j = (int)(ax * 1.27323954473516f);
    j += j & 1;
    const float y = (float)j;
    // Cody-Waite three-constant reduction of pi/4
    float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const int q = (j >> 1) & 3;  // multiples of pi/2
    float s = (q & 1) ? pc : ps;
    float c = (q & 1) ? ps : pc;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
    if (x < 0.0f) s = -s;
    *s_out = s;
    *c_out = c;
}
RT_PM_FN float sin(float x) {
    float s, c;
    sincos(x, &s, &c);
    return s;
}
RT_PM_FN float cos(float x) {
    float s, c;
    sincos(x, &s, &c);
    return c;
}

// asin on [-1, 1]; arguments outside return NaN like the C library.
RT_PM_FN float asin(float x) {
    const float a = fabsf(x);
    if (a > 1.0f) return u2f(0x7fc00000u);
    float z, w;
    const bool big = a > 0.5f;
    if (big) {
        z = 0.5f * (1.0f - a);
        w = sqrtf(z);
    } else {
        w = a;
        z = w * w;
    }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * w + w;
    if (big) p = 1.5707963267948966f - (p + p);
    return x < 0.0f ? -p : p;
}

RT_PM_FN float acos(float x) {
    if (x < -1.0f || x > 1.0f) return u2f(0x7fc00000u);
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966f - asin(x);
}

// atan for x >= 0
RT_PM_FN float atan_pos(float x) {
    float y = 0.0f;
    if (x > 2.414213562373095f) {  // tan(3 pi/8)
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {  // tan(pi/8)
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    }
    const float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return y;
}

// atan2(y, x) in (-pi, pi]
RT_PM_FN float atan2(float y, float x) {
    if (x != x || y != y) return u2f(0x7fc00000u);
    const float ay = fabsf(y), ax = fabsf(x);
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    float a;
    if (ax == 0.0f) a = 1.5707963267948966f;
    else a = atan_pos(ay / ax);  // [0, pi/2]
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}

// natural logarithm for finite x > 0 (the RNG feeds (0,1]); x <= 0 gives -inf / NaN.
RT_PM_FN float log(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -u2f(0x7f800000u) : u2f(0x7fc00000u);
    uint32_t u = f2u(x);
    int e = 0;
    if (u < 0x00800000u) {  // subnormal: scale by 2^23
        x = x * 8388608.0f;
        u = f2u(x);
        e = -23;
    }
    // x = m * 2^e with m in [0.5, 1)
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m +
                   1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m +
               3.3333331174e-1f) * m * z;
    const float fe = (float)e;
    y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    float r = m + y;
    r += 0.693359375f * fe;
    return r;
}

}  // namespace pm
