// BSDF kernels on the path: smooth diffuse, smooth dielectric, rough conductor
// (GGX / Beckmann microfacet with visible-normal sampling), smooth conductor, smooth plastic,
// the two-sided adapter, plus Fresnel terms.
//
// Follows: src/bsdfs/diffuse.cpp:78-135, src/bsdfs/dielectric.cpp:201-320,
// src/bsdfs/roughconductor.cpp:196-382, src/bsdfs/conductor.cpp:202-261, src/bsdfs/plastic.cpp:176-301,
// src/bsdfs/twosided.cpp:94-172, include/mitsuba/render/microfacet.h:184-418,
// include/mitsuba/render/fresnel.h:34-116,275-294.
// Unpolarized RGB only (the scalar_rgb variant); TransportMode::Radiance.
#pragma once
#include "base.h"
#include "warp.h"
#include "special.h"
#include "spectrum.h"
#include "shape.h"

namespace miw {

// BSDFFlags subset (include/mitsuba/render/bsdf.h:40-124)
enum : uint32_t {
    BSDF_DiffuseReflection = 0x00002, BSDF_GlossyReflection = 0x00008, BSDF_GlossyTransmission = 0x00010,
    BSDF_Transmission = 0x00004 | 0x00010 | 0x00040,
    BSDF_DeltaReflection   = 0x00020, BSDF_DeltaTransmission = 0x00040,
    BSDF_Smooth = 0x00002 | 0x00004 | 0x00008 | 0x00010,
    BSDF_Delta  = 0x00001 | 0x00020 | 0x00040,
};

enum : uint32_t { BSDF_TYPE_DIFFUSE = 0, BSDF_TYPE_DIELECTRIC = 1, BSDF_TYPE_ROUGHCONDUCTOR = 2,
                  BSDF_TYPE_CONDUCTOR = 3, BSDF_TYPE_PLASTIC = 4, BSDF_TYPE_ROUGHDIELECTRIC = 5, BSDF_TYPE_ROUGHPLASTIC = 6,
                  BSDF_TYPE_COUNT = 7 };
#define MIW_ROUGH_TRANSMITTANCE_RES 64      /* roughplastic.cpp:12 */
// record flag bits: 0-1 belong to the type (roughconductor: GGX, sample_visible; plastic: nonlinear, has
// specular_reflectance); bit 8 marks a record wrapped by the twosided adapter, whose back side is record `back`
enum : uint32_t { BSDF_REC_TWOSIDED = 0x100u };
enum : uint32_t { MF_BECKMANN = 0, MF_GGX = 1 };

// 128-byte material record (host fills it from the plugin's Properties). Scalars in p[], the
// plugin's spectrum-valued parameters as texture records (spectrum.h):
//   diffuse:        tex[0] reflectance
//   dielectric:     p[0] eta (= int_ior/ext_ior), tex[0] specular_reflectance, tex[1] specular_transmittance
//   roughconductor: p[0] alpha_u, p[1] alpha_v, tex[0] eta, tex[1] k, tex[2] specular_reflectance;
//                   flags bit0 = GGX, bit1 = sample_visible
//   conductor:      tex[0] eta, tex[1] k, tex[2] specular_reflectance
//   plastic:        p[0] eta, p[1] 1/eta^2, p[2] fdr_int, p[3] specular_sampling_weight (plastic.cpp:163-174),
//                   tex[0] diffuse_reflectance, tex[1] specular_reflectance; flags bit0 = nonlinear,
//                   bit1 = specular_reflectance given
//   roughdielectric: p[0] alpha_u, p[1] alpha_v, p[2] eta, p[3] 1/eta (roughdielectric.cpp:160,200), tex[0]
//                   specular_reflectance, tex[1] specular_transmittance; flags bit0 = GGX, bit1 = sample_visible,
//                   bit2 / bit3 = specular_reflectance / specular_transmittance given
//   roughplastic:   p[0] alpha, p[1] eta, p[2] 1/eta^2, p[3] internal reflectance, p[4] specular_sampling_weight
//                   (roughplastic.cpp:339-371), p[5] = (float) offset of its 64-entry external-transmittance table in
//                   the scene's table buffer (TexCtx::tables); tex[0] diffuse_reflectance, tex[1] specular_reflectance;
//                   flags bit0 = GGX, bit1 = sample_visible, bit2 = specular_reflectance given, bit4 = nonlinear
//   twosided:       the FRONT record with BSDF_REC_TWOSIDED set; `back` = table index of the back side's record
// (scalar_rgb callers may fill only p[] in the legacy layout — diffuse p[0..2]; dielectric p[1..3],
//  p[4..6]; roughconductor p[2..4], p[5..7], p[8..10] — the uploader derives the TEX_RGB records.)
struct BsdfRec { uint32_t type, flags; float p[14]; TexRec tex[3]; uint32_t back; };

struct BSDFSample { V3 wo; float pdf, eta; uint32_t sampled_type; };

// texture slots a record of this type reads
MIW_HD uint32_t bsdf_tex_slots(uint32_t type) {
    return type == BSDF_TYPE_DIFFUSE ? 1u
         : (type == BSDF_TYPE_DIELECTRIC || type == BSDF_TYPE_PLASTIC || type == BSDF_TYPE_ROUGHDIELECTRIC ||
            type == BSDF_TYPE_ROUGHPLASTIC) ? 2u : 3u;
}

MIW_HD uint32_t bsdf_flags(const BsdfRec &b) {
    switch (b.type) {
        case BSDF_TYPE_DIFFUSE:    return BSDF_DiffuseReflection;
        case BSDF_TYPE_DIELECTRIC: return BSDF_DeltaReflection | BSDF_DeltaTransmission;
        case BSDF_TYPE_CONDUCTOR:  return BSDF_DeltaReflection;                          // conductor.cpp:202
        case BSDF_TYPE_PLASTIC:    return BSDF_DeltaReflection | BSDF_DiffuseReflection; // plastic.cpp:156-158
        case BSDF_TYPE_ROUGHDIELECTRIC: return BSDF_GlossyReflection | BSDF_GlossyTransmission;   // roughdielectric.cpp:189-194
        case BSDF_TYPE_ROUGHPLASTIC: return BSDF_GlossyReflection | BSDF_DiffuseReflection;         // roughplastic.cpp:176-178
        default:                   return BSDF_GlossyReflection;
    }
}

// ---- Fresnel ---------------------------------------------------------------------
// fresnel.h:34-70
MIW_HD void fresnel(float cos_theta_i, float eta, float &r_out, float &cos_theta_t_out,
                    float &eta_it_out, float &eta_ti_out) {
    bool outside = cos_theta_i >= 0.f;
    float rcp_eta = rcp(eta),
          eta_it = outside ? eta : rcp_eta,
          eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fnmadd(fnmadd(cos_theta_i, cos_theta_i, 1.f), eta_ti * eta_ti, 1.f);
    float cos_theta_i_abs = abs_(cos_theta_i);
    float cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
    bool index_matched = eta == 1.f,
         special_case = index_matched || cos_theta_i_abs == 0.f;
    float r_sc = index_matched ? 0.f : 1.f;
    float a_s = fnmadd(eta_it, cos_theta_t_abs, cos_theta_i_abs) /
                 fmadd(eta_it, cos_theta_t_abs, cos_theta_i_abs);
    float a_p = fnmadd(eta_it, cos_theta_i_abs, cos_theta_t_abs) /
                 fmadd(eta_it, cos_theta_i_abs, cos_theta_t_abs);
    float r = .5f * (sqr(a_s) + sqr(a_p));
    if (special_case) r = r_sc;
    r_out = r;
    cos_theta_t_out = mulsign_neg(cos_theta_t_abs, cos_theta_i);
    eta_it_out = eta_it; eta_ti_out = eta_ti;
}

// fresnel.h:92-116 (one colour channel)
MIW_HD float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float cos_theta_i_2 = cos_theta_i * cos_theta_i,
          sin_theta_i_2 = 1.f - cos_theta_i_2,
          sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    float temp_1   = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2,
          a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
          a        = safe_sqrt(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + cos_theta_i_2,
          term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4,
          term_4 = term_2 * sin_theta_i_2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return .5f * (r_s + r_p);
}

// fresnel.h:275-294
MIW_HD V3 reflect(V3 wi) { return v3(-wi.x, -wi.y, wi.z); }
MIW_HD V3 reflect(V3 wi, V3 m) { return fmsub3(m, 2.f * dot(wi, m), wi); }
MIW_HD V3 refract(V3 wi, float cos_theta_t, float eta_ti) {
    return v3(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t);
}

// ---- Microfacet distribution (microfacet.h) ----------------------------------------
struct Microfacet { uint32_t type; float alpha_u, alpha_v; bool sample_visible; };

MIW_HD Microfacet microfacet_make(uint32_t type, float au, float av, bool sv) {
    Microfacet d; d.type = type; d.sample_visible = sv;
    d.alpha_u = max_(au, 1e-4f); d.alpha_v = max_(av, 1e-4f);   // configure(), :415-418
    return d;
}

// :184-202
MIW_HD float mf_eval(const Microfacet &d, V3 m) {
    float alpha_uv = d.alpha_u * d.alpha_v,
          cos_theta = m.z,
          cos_theta_2 = sqr(cos_theta),
          result;
    if (d.type == MF_BECKMANN)
        result = exp_(-(sqr(m.x / d.alpha_u) + sqr(m.y / d.alpha_v)) / cos_theta_2)
                 / (MIW_PI * alpha_uv * sqr(cos_theta_2));
    else
        result = rcp(MIW_PI * alpha_uv *
                     sqr(sqr(m.x / d.alpha_u) + sqr(m.y / d.alpha_v) + sqr(m.z)));
    return (result * cos_theta > 1e-20f) ? result : 0.f;
}

// :331-355
MIW_HD float mf_smith_g1(const Microfacet &d, V3 v, V3 m) {
    float xy_alpha_2 = sqr(d.alpha_u * v.x) + sqr(d.alpha_v * v.y),
          tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z),
          result;
    if (d.type == MF_BECKMANN) {
        float a = rsqrt(tan_theta_alpha_2), a_sqr = sqr(a);
        result = a >= 1.6f ? 1.f
                           : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
    } else {
        result = 2.f / (1.f + __builtin_sqrtf(1.f + tan_theta_alpha_2));
    }
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (dot(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
MIW_HD float mf_G(const Microfacet &d, V3 wi, V3 wo, V3 m) {      // :319-321
    return mf_smith_g1(d, wi, m) * mf_smith_g1(d, wo, m);
}

// :358-411
MIW_HD V2 mf_sample_visible_11(uint32_t type, float cos_theta_i, V2 sample) {
    if (type == MF_BECKMANN) {                                     // :359-395
        float tan_theta_i = safe_sqrt(fnmadd(cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i;
        float cot_theta_i = rcp(tan_theta_i);
        float maxval = erf_(cot_theta_i);
        sample.x = max_(min_(sample.x, 1.f - 1e-6f), 1e-6f);
        sample.y = max_(min_(sample.y, 1.f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.f) * erf_(__builtin_sqrtf(-log_(sample.x)));
        sample.x *= 1.f + maxval + MIW_INV_SQRT_PI * tan_theta_i * exp_(-sqr(cot_theta_i));
        for (int i = 0; i < 3; ++i) {                              // three Newton iterations
            float slope = erfinv_(x),
                  value = 1.f + x + MIW_INV_SQRT_PI * tan_theta_i * exp_(-sqr(slope)) - sample.x,
                  derivative = 1.f - slope * tan_theta_i;
            x -= value / derivative;
        }
        return v2(erfinv_(x), erfinv_(fmsub(2.f, sample.y, 1.f)));
    }
    V2 p = square_to_uniform_disk_concentric(sample);              // :396-410
    float s = .5f * (1.f + cos_theta_i);
    p.y = lerp_(safe_sqrt(1.f - sqr(p.x)), p.y, s);
    float x = p.x, y = p.y,
          z = safe_sqrt(1.f - squared_norm2(p));
    float sin_theta_i = safe_sqrt(1.f - sqr(cos_theta_i));
    float nrm = rcp(fmadd(sin_theta_i, y, cos_theta_i * z));
    return v2(fmsub(cos_theta_i, y, sin_theta_i * z) * nrm, x * nrm);
}

// :214-223
MIW_HD float mf_pdf(const Microfacet &d, V3 wi, V3 m) {
    float result = mf_eval(d, m);
    if (d.sample_visible) result *= mf_smith_g1(d, wi, m) * abs_dot(wi, m) / wi.z;
    else                  result *= m.z;
    return result;
}

// :234-316
MIW_HD void mf_sample(const Microfacet &d, V3 wi, V2 sample, V3 &m_out, float &pdf_out) {
    if (!d.sample_visible) {
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2, pdf;
        if (d.alpha_u == d.alpha_v) {                     // :240-243
            sincos_((2.f * MIW_PI) * sample.y, sin_phi, cos_phi);
            alpha_2 = d.alpha_u * d.alpha_u;
        } else {                                          // :244-255 (tan = sin/cos of the shared sincos)
            float s, c;
            sincos_((2.f * MIW_PI) * sample.y, s, c);
            float ratio = d.alpha_v / d.alpha_u,
                  tmp   = ratio * (s / c);
            cos_phi = rsqrt(fmadd(tmp, tmp, 1.f));
            cos_phi = mulsign(cos_phi, abs_(sample.y - .5f) - .25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = rcp(sqr(cos_phi / d.alpha_u) + sqr(sin_phi / d.alpha_v));
        }
        if (d.type == MF_BECKMANN) {                      // :258-266
            cos_theta = rsqrt(fnmadd(alpha_2, log_(1.f - sample.x), 1.f));
            cos_theta_2 = sqr(cos_theta);
            float cos_theta_3 = max_(cos_theta_2 * cos_theta, 1e-20f);
            pdf = (1.f - sample.x) / (MIW_PI * d.alpha_u * d.alpha_v * cos_theta_3);
        } else {                                          // :267-277
            float tan_theta_m_2 = alpha_2 * sample.x / (1.f - sample.x);
            cos_theta = rsqrt(1.f + tan_theta_m_2);
            cos_theta_2 = sqr(cos_theta);
            float temp = 1.f + tan_theta_m_2 / alpha_2,
                  cos_theta_3 = max_(cos_theta_2 * cos_theta, 1e-20f);
            pdf = rcp(MIW_PI * d.alpha_u * d.alpha_v * cos_theta_3 * sqr(temp));
        }
        float sin_theta = __builtin_sqrtf(1.f - cos_theta_2);
        m_out = v3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
        pdf_out = pdf;
    } else {
        float sin_phi, cos_phi, cos_theta;
        V3 wi_p = normalize(v3(d.alpha_u * wi.x, d.alpha_v * wi.y, wi.z));   // step 1
        sincos_phi(wi_p, sin_phi, cos_phi);
        cos_theta = wi_p.z;
        V2 slope = mf_sample_visible_11(d.type, cos_theta, sample);                  // step 2
        slope = v2(fmsub(cos_phi, slope.x, sin_phi * slope.y) * d.alpha_u,   // step 3
                   fmadd(sin_phi, slope.x, cos_phi * slope.y) * d.alpha_v);
        V3 m = normalize(v3(-slope.x, -slope.y, 1.f));                       // step 4
        pdf_out = mf_eval(d, m) * mf_smith_g1(d, wi, m) * abs_dot(wi, m) / wi.z;
        m_out = m;
    }
}

// ---- SmoothDiffuse (diffuse.cpp) ------------------------------------------------
MIW_HD Spec diffuse_sample(const BsdfRec &b, V3 wi, V2 sample2, BSDFSample &bs, const TexCtx &tc) {
    float cos_theta_i = wi.z;
    bs.wo = v3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0;
    if (!(cos_theta_i > 0.f)) return spec(0.f);                      // :88-91
    bs.wo = square_to_cosine_hemisphere(sample2);
    bs.pdf = square_to_cosine_hemisphere_pdf(bs.wo);
    bs.eta = 1.f;
    bs.sampled_type = BSDF_DiffuseReflection;
    return (bs.pdf > 0.f) ? tex_eval(b.tex[0], tc) : spec(0.f);      // :101
}
MIW_HD Spec diffuse_eval(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return spec(0.f);
    return tex_eval(b.tex[0], tc) * MIW_INV_PI * cos_theta_o;        // :116-117
}
MIW_HD float diffuse_pdf(V3 wi, V3 wo) {
    float pdf = square_to_cosine_hemisphere_pdf(wo);
    return (wi.z > 0.f && wo.z > 0.f) ? pdf : 0.f;
}

// ---- SmoothDielectric (dielectric.cpp:201-310, unpolarized branch :289-307) -------
MIW_HD Spec dielectric_sample(const BsdfRec &b, V3 wi, float sample1, BSDFSample &bs, const TexCtx &tc) {
    float cos_theta_i = wi.z;
    float r_i, cos_theta_t, eta_it, eta_ti;
    fresnel(cos_theta_i, b.p[0], r_i, cos_theta_t, eta_it, eta_ti);
    float t_i = 1.f - r_i;
    bool selected_r = sample1 <= r_i;                                // :221
    bs.pdf = selected_r ? r_i : t_i;
    bs.sampled_type = selected_r ? BSDF_DeltaReflection : BSDF_DeltaTransmission;
    bs.wo = selected_r ? reflect(wi) : refract(wi, cos_theta_t, eta_ti);
    bs.eta = selected_r ? 1.f : eta_it;
    Spec weight = spec(1.f);                                         // :290
    if (selected_r) weight = weight * tex_eval(b.tex[0], tc);        // :296-297
    else {
        weight = weight * tex_eval(b.tex[1], tc);                    // :299-300
        weight = weight * sqr(eta_ti);                               // :302-307 (Radiance mode)
    }
    return weight;
}

// ---- RoughConductor (roughconductor.cpp) -------------------------------------------
MIW_HD Microfacet rc_distr(const BsdfRec &b) {
    return microfacet_make((b.flags & 1u) ? MF_GGX : MF_BECKMANN, b.p[0], b.p[1], (b.flags & 2u) != 0);
}
MIW_HD Spec rc_fresnel(const BsdfRec &b, float c, const TexCtx &tc) {
    Spec eta = tex_eval(b.tex[0], tc), k = tex_eval(b.tex[1], tc), r;
#if MIW_SPECTRAL
    for (int i = 0; i < 4; ++i) r.c[i] = fresnel_conductor(c, eta.c[i], k.c[i]);
#else
    r = v3(fresnel_conductor(c, eta.x, k.x), fresnel_conductor(c, eta.y, k.y), fresnel_conductor(c, eta.z, k.z));
#endif
    return r;
}
// :196-275
MIW_HD Spec roughconductor_sample(const BsdfRec &b, V3 wi, V2 sample2, BSDFSample &bs, const TexCtx &tc) {
    bs.wo = v3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0;
    float cos_theta_i = wi.z;
    if (!(cos_theta_i > 0.f)) return spec(0.f);
    Microfacet distr = rc_distr(b);
    V3 m;
    mf_sample(distr, wi, sample2, m, bs.pdf);
    bs.wo = reflect(wi, m);
    bs.eta = 1.f;
    bs.sampled_type = BSDF_GlossyReflection;
    bool active = bs.pdf != 0.f && bs.wo.z > 0.f;
    float weight;
    if (distr.sample_visible) weight = mf_smith_g1(distr, bs.wo, m);
    else weight = mf_G(distr, wi, bs.wo, m) * dot(wi, m) / (cos_theta_i * m.z);
    bs.pdf /= 4.f * dot(bs.wo, m);
    Spec F = rc_fresnel(b, dot(wi, m), tc);
    Spec w = spec(weight) * tex_eval(b.tex[2], tc);
    return active ? F * w : spec(0.f);
}
// :277-345
MIW_HD Spec roughconductor_eval(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return spec(0.f);
    V3 H = normalize(wo + wi);
    Microfacet distr = rc_distr(b);
    float D = mf_eval(distr, H);
    bool active = D != 0.f;
    float G = mf_G(distr, wi, wo, H);
    float res = D * G / (4.f * wi.z);
    Spec F = rc_fresnel(b, dot(wi, H), tc);
    Spec result = spec(res) * tex_eval(b.tex[2], tc);
    return active ? F * result : spec(0.f);
}
// :347-382
MIW_HD float roughconductor_pdf(const BsdfRec &b, V3 wi, V3 wo) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    V3 m = normalize(wo + wi);
    bool active = cos_theta_i > 0.f && cos_theta_o > 0.f && dot(wi, m) > 0.f && dot(wo, m) > 0.f;
    if (!active) return 0.f;
    Microfacet distr = rc_distr(b);
    float result;
    if (distr.sample_visible)
        result = mf_eval(distr, m) * mf_smith_g1(distr, wi, m) / (4.f * cos_theta_i);
    else
        result = mf_pdf(distr, wi, m) / (4.f * dot(wo, m));
    return result;
}

// ---- SmoothConductor (conductor.cpp:217-261, unpolarized branch :253-255) ---------------
MIW_HD Spec conductor_sample(const BsdfRec &b, V3 wi, BSDFSample &bs, const TexCtx &tc) {
    bs.wo = v3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0;
    float cos_theta_i = wi.z;
    if (!(cos_theta_i > 0.f)) return spec(0.f);                      // :223-229
    bs.sampled_type = BSDF_DeltaReflection;
    bs.wo = reflect(wi);
    bs.eta = 1.f;
    bs.pdf = 1.f;
    return tex_eval(b.tex[2], tc) * rc_fresnel(b, cos_theta_i, tc);   // :254
}

// ---- SmoothPlastic (plastic.cpp) ---------------------------------------------------------
MIW_HD float plastic_fresnel(float cos_theta, float eta) {
    float r, ct, a, c;
    fresnel(cos_theta, eta, r, ct, a, c);
    return r;
}
// diff = diffuse_reflectance / (1 - (nonlinear ? diff * fdr_int : fdr_int)), :237-238, :264-265
MIW_HD Spec plastic_diffuse(const BsdfRec &b, const TexCtx &tc) {
    Spec value = tex_eval(b.tex[0], tc);
    const float fdr_int = b.p[2];
    if (b.flags & 1u) {
#if MIW_SPECTRAL
        for (int i = 0; i < 4; ++i) value.c[i] = value.c[i] / (1.f - value.c[i] * fdr_int);
#else
        value = v3(value.x / (1.f - value.x * fdr_int), value.y / (1.f - value.y * fdr_int), value.z / (1.f - value.z * fdr_int));
#endif
    } else {
        const float den = 1.f - fdr_int;
#if MIW_SPECTRAL
        for (int i = 0; i < 4; ++i) value.c[i] = value.c[i] / den;
#else
        value = v3(value.x / den, value.y / den, value.z / den);
#endif
    }
    return value;
}
// :176-244 (all components enabled: the path integrator's BSDFContext)
MIW_HD Spec plastic_sample(const BsdfRec &b, V3 wi, float sample1, V2 sample2, BSDFSample &bs, const TexCtx &tc) {
    bs.wo = v3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0;
    float cos_theta_i = wi.z;
    if (!(cos_theta_i > 0.f)) return spec(0.f);                      // :187-193
    const float eta = b.p[0], inv_eta_2 = b.p[1], ssw = b.p[3];
    float f_i = plastic_fresnel(cos_theta_i, eta),
          prob_specular = f_i * ssw,
          prob_diffuse = (1.f - f_i) * (1.f - ssw);
    prob_specular = prob_specular / (prob_specular + prob_diffuse);  // :203
    prob_diffuse = 1.f - prob_specular;
    bs.eta = 1.f;
    if (sample1 < prob_specular) {                                   // :207, :213-224
        bs.wo = reflect(wi);
        bs.pdf = prob_specular;
        bs.sampled_type = BSDF_DeltaReflection;
        Spec value = spec(f_i / bs.pdf);
        if (b.flags & 2u) value = value * tex_eval(b.tex[1], tc);
        return value;
    }
    bs.wo = square_to_cosine_hemisphere(sample2);                    // :226-241
    bs.pdf = prob_diffuse * square_to_cosine_hemisphere_pdf(bs.wo);
    bs.sampled_type = BSDF_DiffuseReflection;
    float f_o = plastic_fresnel(bs.wo.z, eta);
    Spec value = plastic_diffuse(b, tc);
    return value * (inv_eta_2 * (1.f - f_i) * (1.f - f_o) / prob_diffuse);
}
// :246-270
MIW_HD Spec plastic_eval(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return spec(0.f);
    const float eta = b.p[0], inv_eta_2 = b.p[1];
    float f_i = plastic_fresnel(cos_theta_i, eta), f_o = plastic_fresnel(cos_theta_o, eta);
    Spec diff = plastic_diffuse(b, tc);
    return diff * (square_to_cosine_hemisphere_pdf(wo) * inv_eta_2 * (1.f - f_i) * (1.f - f_o));
}
// :272-298
MIW_HD float plastic_pdf(const BsdfRec &b, V3 wi, V3 wo) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return 0.f;
    const float eta = b.p[0], ssw = b.p[3];
    float f_i = plastic_fresnel(cos_theta_i, eta),
          prob_specular = f_i * ssw,
          prob_diffuse = (1.f - f_i) * (1.f - ssw);
    prob_diffuse = prob_diffuse / (prob_specular + prob_diffuse);
    return square_to_cosine_hemisphere_pdf(wo) * prob_diffuse;
}

// ---- RoughDielectric (roughdielectric.cpp:204-441; both lobes enabled, TransportMode::Radiance) ------------
MIW_HD V3 refract(V3 wi, V3 m, float cos_theta_t, float eta_ti) {            // fresnel.h:310-313
    return fmsub3(m, fmadd(dot(wi, m), eta_ti, cos_theta_t), wi * eta_ti);
}
MIW_HD V3 mulsign3(V3 v, float s) { return v3(mulsign(v.x, s), mulsign(v.y, s), mulsign(v.z, s)); }
MIW_HD Microfacet rd_distr(const BsdfRec &b) {
    return microfacet_make((b.flags & 1u) ? MF_GGX : MF_BECKMANN, b.p[0], b.p[1], (b.flags & 2u) != 0);
}
// :204-316
MIW_HD Spec roughdielectric_sample(const BsdfRec &b, V3 wi, float sample1, V2 sample2, BSDFSample &bs, const TexCtx &tc) {
    bs.wo = v3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0;
    const float eta = b.p[2];
    float cos_theta_i = wi.z;
    bool active = cos_theta_i != 0.f;                                // :219
    Microfacet distr = rd_distr(b), sample_distr = distr;
    if (!distr.sample_visible) {                                     // :230-232, microfacet.h:173-176
        float scale = 1.2f - .2f * __builtin_sqrtf(abs_(cos_theta_i));
        sample_distr.alpha_u *= scale; sample_distr.alpha_v *= scale;
    }
    V3 m;
    mf_sample(sample_distr, mulsign3(wi, cos_theta_i), sample2, m, bs.pdf);   // :235-237
    active = active && bs.pdf != 0.f;
    float F, cos_theta_t, eta_it, eta_ti;
    fresnel(dot(wi, m), eta, F, cos_theta_t, eta_it, eta_ti);        // :240-241
    bool selected_r = sample1 <= F && active;                        // :247
    Spec weight = spec(1.f);
    bs.pdf *= selected_r ? F : 1.f - F;
    bool selected_t = !selected_r && active;
    bs.eta = selected_r ? 1.f : eta_it;                              // :261-265
    bs.sampled_type = selected_r ? BSDF_GlossyReflection : BSDF_GlossyTransmission;
    float dwh_dwo = 0.f;
    if (selected_r) {                                                // :270-279
        bs.wo = reflect(wi, m);
        if (b.flags & 4u) weight = weight * tex_eval(b.tex[0], tc);
        dwh_dwo = rcp(4.f * dot(bs.wo, m));
    }
    if (selected_t) {                                                // :282-300
        bs.wo = refract(wi, m, cos_theta_t, eta_ti);
        Spec factor = spec(sqr(eta_ti));
        if (b.flags & 8u) factor = factor * tex_eval(b.tex[1], tc);
        weight = weight * factor;
        dwh_dwo = (sqr(bs.eta) * dot(bs.wo, m)) / sqr(dot(wi, m) + bs.eta * dot(bs.wo, m));
    }
    if (distr.sample_visible) weight = weight * mf_smith_g1(distr, bs.wo, m);                  // :302-306
    else weight = weight * (mf_G(distr, wi, bs.wo, m) * dot(wi, m) / (cos_theta_i * m.z));
    bs.pdf *= abs_(dwh_dwo);
    return active ? weight : spec(0.f);
}
// :318-386
MIW_HD Spec roughdielectric_eval(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    const float m_eta = b.p[2], m_inv_eta = b.p[3];
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    bool active = cos_theta_i != 0.f;
    bool reflect_ = cos_theta_i * cos_theta_o > 0.f;
    float eta = cos_theta_i > 0.f ? m_eta : m_inv_eta, inv_eta = cos_theta_i > 0.f ? m_inv_eta : m_eta;
    V3 m = normalize(wi + wo * (reflect_ ? 1.f : eta));              // :339
    m = mulsign3(m, m.z);                                            // :342
    Microfacet distr = rd_distr(b);
    float D = mf_eval(distr, m);
    float F, ct, a, c;
    fresnel(dot(wi, m), m_eta, F, ct, a, c);
    float G = mf_G(distr, wi, wo, m);
    if (!active) return spec(0.f);
    if (reflect_) {                                                  // :363-370
        Spec value = spec(F * D * G / (4.f * abs_(cos_theta_i)));
        if (b.flags & 4u) value = value * tex_eval(b.tex[0], tc);
        return value;
    }
    float scale = sqr(inv_eta);                                      // :376
    Spec value = spec(abs_((scale * (1.f - F) * D * G * eta * eta * dot(wi, m) * dot(wo, m)) /
                           (cos_theta_i * sqr(dot(wi, m) + eta * dot(wo, m)))));             // :379-381
    if (b.flags & 8u) value = value * tex_eval(b.tex[1], tc);
    return value;
}
// :388-441
MIW_HD float roughdielectric_pdf(const BsdfRec &b, V3 wi, V3 wo) {
    const float m_eta = b.p[2], m_inv_eta = b.p[3];
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    bool active = cos_theta_i != 0.f;
    bool reflect_ = cos_theta_i * cos_theta_o > 0.f;
    float eta = cos_theta_i > 0.f ? m_eta : m_inv_eta;
    V3 m = normalize(wi + wo * (reflect_ ? 1.f : eta));
    m = mulsign3(m, m.z);
    active = active && dot(wi, m) * wi.z > 0.f && dot(wo, m) * wo.z > 0.f;     // :415-416
    float dwh_dwo = reflect_ ? rcp(4.f * dot(wo, m))
                             : (eta * eta * dot(wo, m)) / sqr(dot(wi, m) + eta * dot(wo, m));   // :419-421
    Microfacet sample_distr = rd_distr(b);
    if (!sample_distr.sample_visible) {                              // :434-435
        float scale = 1.2f - .2f * __builtin_sqrtf(abs_(wi.z));
        sample_distr.alpha_u *= scale; sample_distr.alpha_v *= scale;
    }
    float prob = mf_pdf(sample_distr, mulsign3(wi, wi.z), m);        // :438
    float F, ct, a, c;
    fresnel(dot(wi, m), m_eta, F, ct, a, c);
    prob *= reflect_ ? F : 1.f - F;                                  // :440-443
    return active ? prob * abs_(dwh_dwo) : 0.f;
}

// ---- dispatch (the BSDF plugin vtable, flattened) -------------------------------------
// ---- RoughPlastic (roughplastic.cpp:182-334; both components enabled) ------------------------------------------
MIW_HD Microfacet rp_distr(const BsdfRec &b) { return microfacet_make((b.flags & 1u) ? MF_GGX : MF_BECKMANN, b.p[0], b.p[0], (b.flags & 2u) != 0); }
// lerp_gather over the plugin's external-transmittance table, :296-307
MIW_HD float rp_transmittance(const BsdfRec &b, const TexCtx &tc, float x) {
    const float *data = tc.tables + (uint32_t) b.p[5];
    x *= (float) (MIW_ROUGH_TRANSMITTANCE_RES - 1);
    uint32_t index = (uint32_t) x;
    if (index > (uint32_t) (MIW_ROUGH_TRANSMITTANCE_RES - 2)) index = MIW_ROUGH_TRANSMITTANCE_RES - 2;
    const float v0 = data[index], v1 = data[index + 1], t = x - (float) index;
    return fmadd(v1, t, fnmadd(v0, t, v0));                          // enoki lerp(a, b, t) = fmadd(b, t, fnmadd(a, t, a))
}
MIW_HD void rp_probabilities(const BsdfRec &b, float t_i, float &prob_specular, float &prob_diffuse) {   // :206-215
    prob_specular = (1.f - t_i) * b.p[4];
    prob_diffuse = t_i * (1.f - b.p[4]);
    prob_specular = prob_specular / (prob_specular + prob_diffuse);
    prob_diffuse = 1.f - prob_specular;
}
// :309-334
MIW_HD float roughplastic_pdf(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    const float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return 0.f;
    float prob_specular, prob_diffuse;
    rp_probabilities(b, rp_transmittance(b, tc, cos_theta_i), prob_specular, prob_diffuse);
    const V3 H = normalize(wo + wi);
    const Microfacet distr = rp_distr(b);
    float result;
    if (distr.sample_visible) result = mf_eval(distr, H) * mf_smith_g1(distr, wi, H) / (4.f * cos_theta_i);
    else result = mf_pdf(distr, wi, H) / (4.f * dot(wo, H));
    result *= prob_specular;
    result += prob_diffuse * square_to_cosine_hemisphere_pdf(wo);
    return result;
}
// :243-294
MIW_HD Spec roughplastic_eval(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    const float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (!(cos_theta_i > 0.f && cos_theta_o > 0.f)) return spec(0.f);
    const Microfacet distr = rp_distr(b);
    const V3 H = normalize(wo + wi);
    const float D = mf_eval(distr, H);
    float F, ct, a, c;
    fresnel(dot(wi, H), b.p[1], F, ct, a, c);
    const float G = mf_G(distr, wi, wo, H);
    Spec value = spec(F * D * G / (4.f * cos_theta_i));
    if (b.flags & 4u) value = value * tex_eval(b.tex[1], tc);
    const float t_i = rp_transmittance(b, tc, cos_theta_i), t_o = rp_transmittance(b, tc, cos_theta_o);
    Spec diff = tex_eval(b.tex[0], tc);
    const float fdr = b.p[3];
#if MIW_SPECTRAL
    for (int i = 0; i < 4; ++i) diff.c[i] = diff.c[i] / (1.f - ((b.flags & 16u) ? diff.c[i] * fdr : fdr));
#else
    diff = v3(diff.x / (1.f - ((b.flags & 16u) ? diff.x * fdr : fdr)), diff.y / (1.f - ((b.flags & 16u) ? diff.y * fdr : fdr)),
              diff.z / (1.f - ((b.flags & 16u) ? diff.z * fdr : fdr)));
#endif
    return value + diff * (MIW_INV_PI * b.p[2] * cos_theta_o * t_i * t_o);
}
// :182-241
MIW_HD Spec roughplastic_sample(const BsdfRec &b, V3 wi, float sample1, V2 sample2, BSDFSample &bs, const TexCtx &tc) {
    bs.wo = v3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0;
    const float cos_theta_i = wi.z;
    if (!(cos_theta_i > 0.f)) return spec(0.f);
    float prob_specular, prob_diffuse;
    rp_probabilities(b, rp_transmittance(b, tc, cos_theta_i), prob_specular, prob_diffuse);
    bs.eta = 1.f;
    if (sample1 < prob_specular) {
        V3 m; float pdf_m;
        mf_sample(rp_distr(b), wi, sample2, m, pdf_m);
        bs.wo = reflect(wi, m);
        bs.sampled_type = BSDF_GlossyReflection;
    } else {
        bs.wo = square_to_cosine_hemisphere(sample2);
        bs.sampled_type = BSDF_DiffuseReflection;
    }
    bs.pdf = roughplastic_pdf(b, wi, bs.wo, tc);
    if (!(bs.pdf > 0.f)) return spec(0.f);
    return roughplastic_eval(b, wi, bs.wo, tc) / bs.pdf;
}

// Argument order matches BSDF::sample(ctx, si, sample1, sample2) (bsdf.h:328-340).
// `Ext` = false compiles the plugins out that only scenes marked "extended" by the uploader contain (roughplastic).
// `Trio` = true: the caller knows every record to be diffuse, dielectric or roughconductor (BASELINE configs 3 and 4):
// the other plugins are compiled out — a kernel's code size is what its waves stream through the instruction cache.
template <bool Ext = true, bool Trio = false>
MIW_HD Spec bsdf_sample(const BsdfRec &b, V3 wi, float sample1, V2 sample2, BSDFSample &bs, const TexCtx &tc) {
    if (Trio) {
        if (b.type == BSDF_TYPE_DIFFUSE) return diffuse_sample(b, wi, sample2, bs, tc);
        if (b.type == BSDF_TYPE_DIELECTRIC) return dielectric_sample(b, wi, sample1, bs, tc);
        return roughconductor_sample(b, wi, sample2, bs, tc);
    }
    if (Ext && b.type == BSDF_TYPE_ROUGHPLASTIC) return roughplastic_sample(b, wi, sample1, sample2, bs, tc);
    switch (b.type) {
        case BSDF_TYPE_DIFFUSE:    return diffuse_sample(b, wi, sample2, bs, tc);
        case BSDF_TYPE_DIELECTRIC: return dielectric_sample(b, wi, sample1, bs, tc);
        case BSDF_TYPE_CONDUCTOR:  return conductor_sample(b, wi, bs, tc);
        case BSDF_TYPE_PLASTIC:    return plastic_sample(b, wi, sample1, sample2, bs, tc);
        case BSDF_TYPE_ROUGHDIELECTRIC: return roughdielectric_sample(b, wi, sample1, sample2, bs, tc);
        default:                   return roughconductor_sample(b, wi, sample2, bs, tc);
    }
}
template <bool Ext = true, bool Trio = false>
MIW_HD Spec bsdf_eval(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    if (Trio) {
        if (b.type == BSDF_TYPE_DIFFUSE) return diffuse_eval(b, wi, wo, tc);
        if (b.type == BSDF_TYPE_DIELECTRIC) return spec(0.f);
        return roughconductor_eval(b, wi, wo, tc);
    }
    if (Ext && b.type == BSDF_TYPE_ROUGHPLASTIC) return roughplastic_eval(b, wi, wo, tc);
    switch (b.type) {
        case BSDF_TYPE_DIFFUSE:    return diffuse_eval(b, wi, wo, tc);
        case BSDF_TYPE_DIELECTRIC: return spec(0.f);                 // dielectric.cpp:312-315
        case BSDF_TYPE_CONDUCTOR:  return spec(0.f);                 // conductor.cpp:263-266
        case BSDF_TYPE_PLASTIC:    return plastic_eval(b, wi, wo, tc);
        case BSDF_TYPE_ROUGHDIELECTRIC: return roughdielectric_eval(b, wi, wo, tc);
        default:                   return roughconductor_eval(b, wi, wo, tc);
    }
}
template <bool Ext = true, bool Trio = false>
MIW_HD float bsdf_pdf(const BsdfRec &b, V3 wi, V3 wo, const TexCtx &tc) {
    if (Trio) {
        if (b.type == BSDF_TYPE_DIFFUSE) return diffuse_pdf(wi, wo);
        if (b.type == BSDF_TYPE_DIELECTRIC) return 0.f;
        return roughconductor_pdf(b, wi, wo);
    }
    if (Ext && b.type == BSDF_TYPE_ROUGHPLASTIC) return roughplastic_pdf(b, wi, wo, tc);
    switch (b.type) {
        case BSDF_TYPE_DIFFUSE:    return diffuse_pdf(wi, wo);
        case BSDF_TYPE_DIELECTRIC: return 0.f;                       // dielectric.cpp:317-320
        case BSDF_TYPE_CONDUCTOR:  return 0.f;                       // conductor.cpp:268-271
        case BSDF_TYPE_PLASTIC:    return plastic_pdf(b, wi, wo);
        case BSDF_TYPE_ROUGHDIELECTRIC: return roughdielectric_pdf(b, wi, wo);
        default:                   return roughconductor_pdf(b, wi, wo);
    }
}

// ---- TwoSidedBRDF (twosided.cpp) -------------------------------------------------------
// A shape's BSDF as the integrator sees it: the record itself, or — for a twosided record — the front
// record when cos(theta_i) > 0, the back record with wi (and wo) mirrored when cos(theta_i) < 0 (:105-124),
// nothing when cos(theta_i) == 0. `flags` is BSDF::flags() of the plugin (twosided: both sides, :76-86).
struct BsdfSide { const BsdfRec *b; bool flip, none; uint32_t flags; };
MIW_HD BsdfSide bsdf_side(const BsdfRec *table, uint32_t index, V3 wi) {
    BsdfSide s; s.b = table + index; s.flip = false; s.none = false; s.flags = bsdf_flags(*s.b);
    if (s.b->flags & BSDF_REC_TWOSIDED) {
        const BsdfRec *back = table + s.b->back;
        s.flags |= bsdf_flags(*back);
        if (wi.z < 0.f) { s.flip = true; s.b = back; }
        else if (!(wi.z > 0.f)) s.none = true;
    }
    return s;
}
MIW_HD V3 bsdf_mirror(V3 w) { return v3(w.x, w.y, w.z * -1.f); }      // `wi.z() *= -1.f`
template <bool Ext = true, bool Trio = false>
MIW_HD Spec bsdf_side_sample(const BsdfSide &s, V3 wi, float sample1, V2 sample2, BSDFSample &bs, const TexCtx &tc) {
    if (s.none) { bs.wo = v3(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0; return spec(0.f); }
    Spec v = bsdf_sample<Ext, Trio>(*s.b, s.flip ? bsdf_mirror(wi) : wi, sample1, sample2, bs, tc);
    if (s.flip) bs.wo.z *= -1.f;                                     // :121
    return v;
}
// (one inlined copy of the plugin code each: the mirrored directions are selected first, not the results)
template <bool Ext = true, bool Trio = false>
MIW_HD Spec bsdf_side_eval(const BsdfSide &s, V3 wi, V3 wo, const TexCtx &tc) {
    if (s.none) return spec(0.f);
    return bsdf_eval<Ext, Trio>(*s.b, s.flip ? bsdf_mirror(wi) : wi, s.flip ? bsdf_mirror(wo) : wo, tc);
}
template <bool Ext = true, bool Trio = false>
MIW_HD float bsdf_side_pdf(const BsdfSide &s, V3 wi, V3 wo, const TexCtx &tc) {
    if (s.none) return 0.f;
    return bsdf_pdf<Ext, Trio>(*s.b, s.flip ? bsdf_mirror(wi) : wi, s.flip ? bsdf_mirror(wo) : wo, tc);
}

} // namespace miw
