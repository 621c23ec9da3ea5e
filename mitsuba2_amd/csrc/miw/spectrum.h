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
//   TEX_BITMAP    (float) index into the scene's bitmap table (src/textures/bitmap.cpp), looked up at si.uv
enum : uint32_t { TEX_RGB = 0, TEX_UNIFORM = 1, TEX_SRGB = 2, TEX_D65 = 3, TEX_SRGB_D65 = 4, TEX_BITMAP = 5 };
struct TexRec { uint32_t type; float v[4]; };

// BitmapTextureImpl (bitmap.cpp:244-262): `data` = width * height * channels floats, row-major; channels 1 = a
// scalar (raw or luminance-free greyscale image), 3 = linear RGB (scalar_rgb) or the sRGB-model coefficients the
// host fetched per texel (scalar_spectral, bitmap.cpp:156-165); to_uv = the 2 x 3 affine part of the `to_uv`
// transform (its three columns, transform.h:324-348).
enum : uint32_t { BITMAP_NEAREST = 0, BITMAP_BILINEAR = 1 };
enum : uint32_t { BITMAP_REPEAT = 0, BITMAP_MIRROR = 1, BITMAP_CLAMP = 2 };
struct BitmapRec {
    const float *data;
    uint32_t width, height, channels, filter, wrap, pad;
    float to_uv[6];
};

// What a texture lookup needs of the surface interaction: wavelengths, uv (and the scene's bitmap table).
// A null table (or a constant record) evaluates exactly as before bitmaps existed.
// `tables`: the scene's buffer of per-plugin float tables (roughplastic's transmittance table), or nullptr.
struct TexCtx {
    Wavelengths wl; V2 uv; const BitmapRec *bitmaps; const float *tables;
    MIW_HD TexCtx(const Wavelengths &w) : wl(w), uv(v2(0.f, 0.f)), bitmaps(nullptr), tables(nullptr) { }
    MIW_HD TexCtx(const Wavelengths &w, V2 uv_, const BitmapRec *b, const float *t = nullptr) : wl(w), uv(uv_), bitmaps(b), tables(t) { }
};

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

// ---- bitmap texture, src/textures/bitmap.cpp:385-460 ---------------------------------------------------------
MIW_HD int bitmap_floor2int(float x) {
    if (!(x == x)) return 0;
    int i = (int) x;
    return ((float) i > x) ? i - 1 : i;
}
// :385-399 (enoki::divisor rounds toward zero like operator/)
MIW_HD int bitmap_wrap(int value, int res, uint32_t mode) {
    if (mode == BITMAP_CLAMP) return value < 0 ? 0 : (value > res - 1 ? res - 1 : value);
    const int div = value / res;
    int mod = value - div * res;
    if (mod < 0) mod += res;
    if (mode == BITMAP_MIRROR && !(((div & 1) == 0) != (value < 0))) mod = res - 1 - mod;
    return mod;
}
// one texel as a Spec: a scalar broadcasts, an RGB triple is the colour (or, in spectral builds, the
// coefficient triple of the sRGB model evaluated at the sample's wavelengths, :439-444)
MIW_HD Spec bitmap_texel(const BitmapRec &b, int x, int y, const Wavelengths &wl) {
    const float *p = b.data + ((size_t) y * b.width + (size_t) x) * b.channels;
    if (b.channels == 1) return spec(p[0]);
#if MIW_SPECTRAL
    Spec r;
    for (int i = 0; i < 4; ++i) r.c[i] = srgb_model_eval(p, wl.l[i]);
    return r;
#else
    (void) wl;
    return v3(p[0], p[1], p[2]);
#endif
}
MIW_HD Spec spec_fmadd(float a, Spec x, Spec y) {            // fmadd(a, x, y) per channel
#if MIW_SPECTRAL
    Spec r; for (int i = 0; i < 4; ++i) r.c[i] = fmadd(a, x.c[i], y.c[i]); return r;
#else
    return v3(fmadd(a, x.x, y.x), fmadd(a, x.y, y.y), fmadd(a, x.z, y.z));
#endif
}
// BitmapTextureImpl::interpolate, :401-460
MIW_HD Spec bitmap_eval(const BitmapRec &b, V2 uv_in, const Wavelengths &wl) {
    // m_transform.transform_affine(si.uv), transform.h:90-98
    float u = b.to_uv[4], v = b.to_uv[5];
    u = fmadd(b.to_uv[0], uv_in.x, u); v = fmadd(b.to_uv[1], uv_in.x, v);
    u = fmadd(b.to_uv[2], uv_in.y, u); v = fmadd(b.to_uv[3], uv_in.y, v);
    const int w = (int) b.width, h = (int) b.height;
    if (b.filter == BITMAP_BILINEAR) {
        u = fmadd(u, (float) w, -.5f); v = fmadd(v, (float) h, -.5f);          // :415
        const int xi = bitmap_floor2int(u), yi = bitmap_floor2int(v);          // :418
        const float w1x = u - (float) xi, w1y = v - (float) yi, w0x = 1.f - w1x, w0y = 1.f - w1y;   // :421-422
        const int x0 = bitmap_wrap(xi, w, b.wrap), x1 = bitmap_wrap(xi + 1, w, b.wrap),
                  y0 = bitmap_wrap(yi, h, b.wrap), y1 = bitmap_wrap(yi + 1, h, b.wrap);
        const Spec v00 = bitmap_texel(b, x0, y0, wl), v10 = bitmap_texel(b, x1, y0, wl),
                   v01 = bitmap_texel(b, x0, y1, wl), v11 = bitmap_texel(b, x1, y1, wl);
        const Spec v0 = spec_fmadd(w0x, v00, w1x * v10), v1 = spec_fmadd(w0x, v01, w1x * v11);   // :447-448
        return spec_fmadd(w0y, v0, w1y * v1);                                  // :450
    }
    u *= (float) w; v *= (float) h;                                            // :453
    return bitmap_texel(b, bitmap_wrap(bitmap_floor2int(u), w, b.wrap), bitmap_wrap(bitmap_floor2int(v), h, b.wrap), wl);
}

// A plugin parameter at a surface point: a constant (tex_eval above) or a bitmap lookup.
MIW_HD Spec tex_eval(const TexRec &t, const TexCtx &tc) {
    if (tc.bitmaps && t.type == TEX_BITMAP) return bitmap_eval(tc.bitmaps[(uint32_t) t.v[0]], tc.uv, tc.wl);
    return tex_eval(t, tc.wl);
}

} // namespace miw
