// Transcendentals the Beckmann microfacet model needs: exp, log, erf, erfinv.
//
// The reference reaches them through Enoki (enoki/special.h + libm scalar
// fallbacks, include/mitsuba/render/microfacet.h:3,193,266,372-394); Enoki is
// not vendored in the reference checkout and libm is not bit-reproducible
// across glibc / ROCm device libs, so — exactly like sincos in warp.h — the
// path uses ONE shared float32 implementation on both sides, built from
// correctly rounded {+,-,*,/,sqrt,fma} only:
//   exp_, log_ : Cephes single-precision expf / logf (public-domain algorithm
//                restated: Cody-Waite reduction + degree-5 / degree-8 minimax),
//                <= 1 ulp over the normal range;
//   erf_       : Cephes ndtrf.c erff / erfcf scheme (x*P(x^2) for |x| < 1,
//                1 - exp(-x^2)/x * Q(1/x^2) beyond), max rel. error 3.8e-7;
//   atan2_, acos_ : Cephes atanf / asinf / acosf (range reduction + odd minimax
//                polynomials), <= 2 ulp; used by the environment map's lat-long
//                lookup (src/emitters/envmap.cpp:140-142);
//   erfinv_    : M. Giles, "Approximating the erfinv function" (GPU Computing
//                Gems 2010), single-precision variant, max rel. error 2.8e-7.
// Accuracy figures are measured against scipy.special (tests/test_oracle_kat.py);
// they are the accuracy class of the routines they stand in for.
#pragma once
#include "base.h"

namespace miw {

#define MIW_INV_SQRT_PI 0.56418958354775628695f

// 2^n * y for n in [-150, 128] without double rounding on the way to a denormal
MIW_HD float ldexp_(float y, int n) {
    if (n > 127) { y *= 1.7014118346046923e+38f; n -= 127; }             // 2^127
    if (n < -126) { y *= 1.1754943508222875e-38f; n += 126; }            // 2^-126 (exact: y is normal)
    if (n < -126) return y * 0.f;
    return y * u2f((uint32_t) (n + 127) << 23);
}

MIW_HD float exp_(float x) {
    if (!(x == x)) return x;
    if (x > 88.72283935546875f) return MIW_INFINITY;
    if (x < -103.97208404541016f) return 0.f;
    float fn = __builtin_floorf(fmadd(x, 1.44269504088896341f, 0.5f));
    float r = fnmadd(fn, 0.693359375f, x);
    r = fnmadd(fn, -2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmadd(p, r, 1.3981999507e-3f);
    p = fmadd(p, r, 8.3334519073e-3f);
    p = fmadd(p, r, 4.1665795894e-2f);
    p = fmadd(p, r, 1.6666665459e-1f);
    p = fmadd(p, r, 5.0000001201e-1f);
    float y = fmadd(p, z, r) + 1.f;
    return ldexp_(y, (int) fn);
}

MIW_HD float log_(float x) {
    if (!(x == x) || x < 0.f) return __builtin_nanf("");
    if (x == 0.f) return -MIW_INFINITY;
    if (!isfinite_(x)) return x;
    int e = 0;
    uint32_t b = f2u(x);
    if (b < 0x00800000u) { x *= 8388608.f; b = f2u(x); e = -23; }        // denormal: scale by 2^23
    e += (int) (b >> 23) - 126;
    float m = u2f((b & 0x007fffffu) | 0x3f000000u);                      // [0.5, 1)
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.f; } else m = m - 1.f;
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmadd(y, m, -1.1514610310e-1f);
    y = fmadd(y, m, 1.1676998740e-1f);
    y = fmadd(y, m, -1.2420140846e-1f);
    y = fmadd(y, m, 1.4249322787e-1f);
    y = fmadd(y, m, -1.6668057665e-1f);
    y = fmadd(y, m, 2.0000714765e-1f);
    y = fmadd(y, m, -2.4999993993e-1f);
    y = fmadd(y, m, 3.3333331174e-1f);
    y = y * m * z;
    float fe = (float) e;
    y = fmadd(-2.12194440e-4f, fe, y);
    y = fnmadd(0.5f, z, y);
    float r = m + y;
    return fmadd(0.693359375f, fe, r);
}

MIW_HD float erf_(float x) {
    float a = abs_(x);
    if (a < 1.f) {
        float z = x * x;
        float p = 7.853861353153693e-5f;
        p = fmadd(p, z, -8.010193625184903e-4f);
        p = fmadd(p, z, 5.188327685732524e-3f);
        p = fmadd(p, z, -2.685381193529856e-2f);
        p = fmadd(p, z, 1.128358514861418e-1f);
        p = fmadd(p, z, -3.761262582423300e-1f);
        p = fmadd(p, z, 1.128379165726710e+0f);
        return x * p;
    }
    if (!(a == a)) return x;
    float ez = exp_(-(a * a)), q = 1.f / a, y = q * q, p;
    if (a < 2.f) {
        p = 2.326819970068386e-2f;
        p = fmadd(p, y, -1.387039388740657e-1f);
        p = fmadd(p, y, 3.687424674597105e-1f);
        p = fmadd(p, y, -5.824733027278666e-1f);
        p = fmadd(p, y, 6.210004621745983e-1f);
        p = fmadd(p, y, -4.944515323274145e-1f);
        p = fmadd(p, y, 3.404879937665872e-1f);
        p = fmadd(p, y, -2.741127028184656e-1f);
        p = fmadd(p, y, 5.638259427386472e-1f);
    } else {
        p = -1.047766399936249e+1f;
        p = fmadd(p, y, 1.297719955372516e+1f);
        p = fmadd(p, y, -7.495518717768503e+0f);
        p = fmadd(p, y, 2.921019019210786e+0f);
        p = fmadd(p, y, -1.015265279202700e+0f);
        p = fmadd(p, y, 4.218463358204948e-1f);
        p = fmadd(p, y, -2.820767439740514e-1f);
        p = fmadd(p, y, 5.641895067754075e-1f);
    }
    float r = 1.f - ez * q * p;
    return mulsign(r, x);
}

MIW_HD float erfinv_(float x) {
    float a = abs_(x);
    if (!(a < 1.f)) return a == 1.f ? mulsign(MIW_INFINITY, x) : __builtin_nanf("");
    float w = -log_((1.f - x) * (1.f + x)), p;
    if (w < 5.f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmadd(p, w, 3.43273939e-07f);
        p = fmadd(p, w, -3.5233877e-06f);
        p = fmadd(p, w, -4.39150654e-06f);
        p = fmadd(p, w, 0.00021858087f);
        p = fmadd(p, w, -0.00125372503f);
        p = fmadd(p, w, -0.00417768164f);
        p = fmadd(p, w, 0.246640727f);
        p = fmadd(p, w, 1.50140941f);
    } else {
        w = __builtin_sqrtf(w) - 3.f;
        p = -0.000200214257f;
        p = fmadd(p, w, 0.000100950558f);
        p = fmadd(p, w, 0.00134934322f);
        p = fmadd(p, w, -0.00367342844f);
        p = fmadd(p, w, 0.00573950773f);
        p = fmadd(p, w, -0.0076224613f);
        p = fmadd(p, w, 0.00943887047f);
        p = fmadd(p, w, 1.00167406f);
        p = fmadd(p, w, 2.83297682f);
    }
    return p * x;
}

// ---- inverse trigonometric functions (environment map lookups) ---------------------------
MIW_HD float atan_(float x) {
    float a = abs_(x), y;
    if (!(a == a)) return x;
    if (a > 2.414213562373095f)       { y = 1.5707963267948966f; a = -(1.f / a); }          // > tan(3 pi / 8)
    else if (a > 0.4142135623730950f) { y = 0.7853981633974483f; a = (a - 1.f) / (a + 1.f); }  // > tan(pi / 8)
    else y = 0.f;
    float z = a * a;
    float p = 8.05374449538e-2f;
    p = fmadd(p, z, -1.38776856032e-1f);
    p = fmadd(p, z, 1.99777106478e-1f);
    p = fmadd(p, z, -3.33329491539e-1f);
    y = y + fmadd(p * z, a, a);
    return mulsign(y, x);
}

MIW_HD float atan2_(float y, float x) {
    if (!(x == x) || !(y == y)) return x + y;
    if (x == 0.f) {
        if (y == 0.f) return (f2u(x) & 0x80000000u) ? mulsign(MIW_PI, y) : y;      // atan2(+-0, -0) = +-pi, (+-0, +0) = +-0
        return mulsign(1.5707963267948966f, y);
    }
    if (y == 0.f) return x > 0.f ? y : mulsign(MIW_PI, y);
    float w = x < 0.f ? mulsign(MIW_PI, y) : 0.f;
    return w + atan_(y / x);
}

MIW_HD float asin_(float x) {
    float a = abs_(x), z, r;
    if (a > 1.f) return __builtin_nanf("");
    bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.f - a); r = __builtin_sqrtf(z); }
    else     { r = a; z = a * a; }
    float p = 4.2163199048e-2f;
    p = fmadd(p, z, 2.4181311049e-2f);
    p = fmadd(p, z, 4.5470025998e-2f);
    p = fmadd(p, z, 7.4953002686e-2f);
    p = fmadd(p, z, 1.6666752422e-1f);
    float v = fmadd(p * z, r, r);
    if (big) v = 1.5707963267948966f - (v + v);
    return mulsign(v, x);
}

MIW_HD float acos_(float x) {
    if (x < -1.f || x > 1.f) return __builtin_nanf("");
    if (x < -0.5f) return MIW_PI - 2.f * asin_(__builtin_sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f)  return 2.f * asin_(__builtin_sqrtf(0.5f * (1.f - x)));
    return 1.5707963267948966f - asin_(x);
}
MIW_HD float safe_acos(float x) { return acos_(clamp_(x, -1.f, 1.f)); }

} // namespace miw
