// Spectral machinery of the scalar_spectral variant: wavelength sampling, CIE 1931 matching
// functions, the sRGB upsampling model and the constant "textures" a plugin parameter can hold.
//
// Follows include/mitsuba/core/spectrum.h:148-314 (cie1931_xyz, spectrum_to_xyz,
// sample_rgb_spectrum, sample_wavelength), include/mitsuba/core/math.h:419-442 (sample_shifted),
// include/mitsuba/render/srgb.h:9-23 (srgb_model_eval), src/spectra/{uniform,srgb,srgb_d65,d65,
// regular}.cpp and include/mitsuba/core/distr_1d.h:378-393 (regular spectrum lookup).
// In the scalar_rgb build a texture is three floats and none of this is compiled in.
#pragma once
#include "base.h"
#include "special.h"
#include "cie_data.h"

namespace miw {

// What a BSDF / emitter parameter evaluates (src/libcore/xml.cpp:1073-1170 decides which one an
// <rgb> / <spectrum> tag becomes). v[] by type:
//   TEX_RGB       r, g, b                      (scalar_rgb: `srgb`, `srgb_d65`, `uniform` all collapse to this)
//   TEX_UNIFORM   value                        (src/spectra/uniform.cpp)
//   TEX_SRGB      c0, c1, c2                   (src/spectra/srgb.cpp: srgb_model_fetch coefficients)
//   TEX_D65       scale / 10568                (src/spectra/d65.cpp -> regular spectrum)
//   TEX_SRGB_D65  c0, c1, c2, d65 scale / 10568 (src/spectra/srgb_d65.cpp)
enum : uint32_t { TEX_RGB = 0, TEX_UNIFORM = 1, TEX_SRGB = 2, TEX_D65 = 3, TEX_SRGB_D65 = 4 };
struct TexRec { uint32_t type; float v[4]; };

#if MIW_SPECTRAL

#define MIW_WAVELENGTH_MIN 360.f
#define MIW_WAVELENGTH_MAX 830.f
#define MIW_CIE_Y_NORMALIZATION ((float) (1.0 / 106.7502593994140625))

#if defined(__HIPCC__)
__device__ __constant__ const float miw_cie1931_dev[MIW_CIE_SAMPLES * 3] = { MIW_CIE1931_TABLE };
__device__ __constant__ const float miw_d65_dev[MIW_CIE_SAMPLES] = { MIW_D65_TABLE };
#endif
static const float miw_cie1931_host[MIW_CIE_SAMPLES * 3] = { MIW_CIE1931_TABLE };
static const float miw_d65_host[MIW_CIE_SAMPLES] = { MIW_D65_TABLE };

MIW_HD const float *cie1931_table() {
#if defined(__HIP_DEVICE_COMPILE__)
    return miw_cie1931_dev;
#else
    return miw_cie1931_host;
#endif
}
MIW_HD const float *d65_table() {
#if defined(__HIP_DEVICE_COMPILE__)
    return miw_d65_dev;
#else
    return miw_d65_host;
#endif
}

// math.h:419-428 — [x, x + 1/4, x + 2/4, x + 3/4], wrapped
MIW_HD void sample_shifted4(float sample, float *out) {
    for (int i = 0; i < 4; ++i) {
        float v = sample + (float) i / 4.f;
        if (v > 1.f) v -= 1.f;
        out[i] = v;
    }
}

// atanh / cosh through the shared log_ / exp_ (enoki's are log/exp compositions as well)
MIW_HD float atanh_(float x) { return 0.5f * log_((1.f + x) / (1.f - x)); }
MIW_HD float cosh_(float x) { float e = exp_(x); return 0.5f * (e + 1.f / e); }

// spectrum.h:271-285 + :305-314 (sample_wavelength, spectral branch)
MIW_HD void sample_wavelengths(float sample, Wavelengths &wl, Spec &weight) {
    float s[4];
    sample_shifted4(sample, s);
    for (int i = 0; i < 4; ++i) {
        float w = 538.f - atanh_(0.8569106254698279f - 1.8275019724092267f * s[i]) * 138.88888888888889f;
        float tmp = cosh_(0.0072f * (w - 538.f));
        wl.l[i] = w;
        weight.c[i] = 253.82f * tmp * tmp;
    }
}

// spectrum.h:148-178, one wavelength
MIW_HD V3 cie1931_xyz(float wavelength) {
    float t = (wavelength - 360.f) * ((float) (MIW_CIE_SAMPLES - 1) / (830.f - 360.f));
    bool active = wavelength >= 360.f && wavelength <= 830.f;
    int i0 = (int) t;
    if (i0 < 0) i0 = 0;
    if (i0 > MIW_CIE_SAMPLES - 2) i0 = MIW_CIE_SAMPLES - 2;
    if (!(t == t)) i0 = 0;
    const float *tb = cie1931_table();
    float w1 = t - (float) i0, w0 = 1.f - w1;
    if (!active) return v3(0.f);
    return v3(fmadd(w0, tb[i0], w1 * tb[i0 + 1]),
              fmadd(w0, tb[MIW_CIE_SAMPLES + i0], w1 * tb[MIW_CIE_SAMPLES + i0 + 1]),
              fmadd(w0, tb[2 * MIW_CIE_SAMPLES + i0], w1 * tb[2 * MIW_CIE_SAMPLES + i0 + 1]));
}

// hmean over 4 entries: ((a + b) + (c + d)) * 0.25  (enoki hsum of a 4-wide array reduces pairwise)
MIW_HD float hmean4(float a, float b, float c, float d) { return ((a + b) + (c + d)) * 0.25f; }

// spectrum.h:212-218
MIW_HD V3 spectrum_to_xyz(Spec value, const Wavelengths &wl) {
    V3 m[4];
    for (int i = 0; i < 4; ++i) m[i] = cie1931_xyz(wl.l[i]);
    return v3(hmean4(m[0].x * value.c[0], m[1].x * value.c[1], m[2].x * value.c[2], m[3].x * value.c[3]),
              hmean4(m[0].y * value.c[0], m[1].y * value.c[1], m[2].y * value.c[2], m[3].y * value.c[3]),
              hmean4(m[0].z * value.c[0], m[1].z * value.c[1], m[2].z * value.c[2], m[3].z * value.c[3]));
}

// srgb.h:9-23, one wavelength
MIW_HD float srgb_model_eval(const float *coeff, float wavelength) {
    float v = fmadd(fmadd(coeff[0], wavelength, coeff[1]), wavelength, coeff[2]);
    if (!isfinite_(coeff[2]) && coeff[2] == coeff[2]) return fmadd(sign_(coeff[2]), .5f, .5f);
    return max_(0.f, fmadd(.5f * v, rsqrt(fmadd(v, v, 1.f)), .5f));
}

// d65.cpp:55-69 -> regular.cpp:68-72 -> ContinuousDistribution::eval_pdf, distr_1d.h:378-393
MIW_HD float d65_eval(float scale, float wavelength) {
    bool active = wavelength >= 360.f && wavelength <= 830.f;
    float x = (wavelength - 360.f) * 0.2f;                 // m_inv_interval_size = float(1 / 5.0)
    int i = (int) x;
    if (i < 0) i = 0;
    if (i > MIW_CIE_SAMPLES - 2) i = MIW_CIE_SAMPLES - 2;
    if (!(x == x)) i = 0;
    const float *tb = d65_table();
    float y0 = active ? tb[i] * scale : 0.f, y1 = active ? tb[i + 1] * scale : 0.f;   // masked gathers read 0
    float w1 = x - (float) i, w0 = 1.f - w1;
    return fmadd(w0, y0, w1 * y1);
}

MIW_HD Spec tex_eval(const TexRec &t, const Wavelengths &wl) {
    Spec r;
    for (int i = 0; i < 4; ++i) {
        const float w = wl.l[i];
        float v;
        switch (t.type) {
            case TEX_UNIFORM:  v = (w >= MIW_WAVELENGTH_MIN && w <= MIW_WAVELENGTH_MAX) ? t.v[0] : 0.f; break;   // uniform.cpp
            case TEX_SRGB:     v = srgb_model_eval(t.v, w); break;
            case TEX_D65:      v = d65_eval(t.v[0], w); break;
            case TEX_SRGB_D65: v = d65_eval(t.v[3], w) * srgb_model_eval(t.v, w); break;                          // srgb_d65.cpp
            default:           v = 0.f; break;
        }
        r.c[i] = v;
    }
    return r;
}

#else   // scalar_rgb

MIW_HD Spec tex_eval(const TexRec &t, const Wavelengths &) { return v3(t.v[0], t.v[1], t.v[2]); }

#endif

} // namespace miw
