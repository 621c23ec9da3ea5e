// Sample warps on the path (include/mitsuba/core/warp.h) plus the one
// transcendental they need.
//
// sincos: the reference calls libm's sincos through enoki (warp.h:87). libm is
// not bit-reproducible across glibc / ROCm device libs, so the path uses ONE
// shared float32 implementation (Cody-Waite 3-term reduction by pi/2 + the
// classic cephes minimax polynomials on [-pi/4, pi/4], fma-evaluated) on both
// sides. |error| <= ~1.5 ulp for |x| < 1e4, which is the accuracy class of the
// libm routine it stands in for.
#pragma once
#include "base.h"

namespace miw {

MIW_HD void sincos_(float x, float &s_out, float &c_out) {
    // quadrant: n = round(x * 2/pi)
    float fn = __builtin_floorf(fmadd(x, 0.636619772367581343f, 0.5f));
    int n = (int) fn;
    // r = x - n*pi/2 in three exact-product steps
    float r = fnmadd(fn, 1.5703125f, x);
    r = fnmadd(fn, 4.837512969970703125e-4f, r);
    r = fnmadd(fn, 7.549789948768648e-8f, r);
    float z = r * r;
    // sin(r), r in [-pi/4, pi/4]
    float ps = fmadd(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmadd(ps, z, -1.6666654611e-1f);
    float sr = fmadd(ps * z, r, r);
    // cos(r)
    float pc = fmadd(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmadd(pc, z, 4.166664568298827e-2f);
    float cr = fmadd(pc * z, z, fnmadd(0.5f, z, 1.f));
    // rotate by quadrant
    float s = (n & 1) ? cr : sr,
          c = (n & 1) ? sr : cr;
    if (n & 2)       s = -s;
    if ((n + 1) & 2) c = -c;
    s_out = s; c_out = c;
}

// warp.h:54-90 — Shirley/Cline concentric map
MIW_HD V2 square_to_uniform_disk_concentric(V2 sample) {
    float x = fmsub(2.f, sample.x, 1.f),
          y = fmsub(2.f, sample.y, 1.f);
    bool is_zero = (x == 0.f) && (y == 0.f),
         quadrant_1_or_3 = abs_(x) < abs_(y);
    float r  = quadrant_1_or_3 ? y : x,
          rp = quadrant_1_or_3 ? x : y;
    float phi = .25f * MIW_PI * rp / r;
    if (quadrant_1_or_3) phi = .5f * MIW_PI - phi;
    if (is_zero) phi = 0.f;
    float s, c;
    sincos_(phi, s, c);
    return v2(r * c, r * s);
}

// warp.h:153-156
MIW_HD V2 square_to_uniform_triangle(V2 sample) {
    float t = safe_sqrt(1.f - sample.x);
    return v2(1.f - t, t * sample.y);
}

// warp.h:325-334
MIW_HD V3 square_to_cosine_hemisphere(V2 sample) {
    V2 p = square_to_uniform_disk_concentric(sample);
    float z = safe_sqrt(1.f - squared_norm2(p));
    return v3(p.x, p.y, z);
}
// warp.h:343-349 (TestDomain = false)
MIW_HD float square_to_cosine_hemisphere_pdf(V3 v) { return MIW_INV_PI * v.z; }

} // namespace miw
