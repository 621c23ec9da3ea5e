// Flat scene view handed to every hot-path function (pointers are HBM addresses
// on the device, ordinary host addresses in the CPU checker), the area emitter
// (src/emitters/area.cpp), Scene::sample_emitter_direction / pdf_emitter_direction
// (src/librender/scene.cpp:164-231) and the perspective sensor
// (src/sensors/perspective.cpp:182-216).
#pragma once
#include "base.h"
#include "shape.h"
#include "bsdf.h"
#include "envmap.h"

namespace miw {

#define MIW_MISS 0xffffffffu     /* "no triangle" in the hit queue */

// 64-byte BVH2 node: both child boxes live in the parent so one aligned 64 B
// fetch feeds two slab tests. child >= 0: inner node index; child < 0: leaf,
// ~child = (first_tri << 4) | (count - 1), count in 1..16; an absent child has
// an inverted box (lo = +inf, hi = -inf) and child = -1... see bvh.h.
struct BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    int32_t parent;      // -1 for the root
    int32_t pad;
};

struct ShapeRec {
    uint32_t bsdf;       // index into bsdfs
    int32_t  emitter;    // index into emitters or -1
    uint32_t flags;      // bit0: has vertex normals; bit3 (SHAPE_HAS_TEXCOORDS): has vertex texture coordinates
    uint32_t pad;
};

struct EmitterRec {
    TexRec radiance;     // area.cpp:55 `radiance` (srgb_d65 / d65 texture in spectral builds, RGB otherwise)
    uint32_t shape;
    uint32_t tri_first;  // first face of this emitter in emit_tri / emit_pmf / emit_cdf
    uint32_t tri_count;
    uint32_t valid_lo, valid_hi;
    float sum, normalization;
    uint32_t flags;      // bit0: emit_vnorm valid for this emitter; bit1: the shape is analytic rectangle `tri_first`
                         // (no face tables; normalization = its 1 / surface area)
    uint32_t type;       // 0: area light on `shape`; 1: the environment map (SceneView::env)
};
enum : uint32_t { EMITTER_AREA = 0, EMITTER_ENVMAP = 1 };
enum : uint32_t { SHAPE_HAS_NORMALS = 1u, SHAPE_HAS_TEXCOORDS = 8u };     // ShapeRec::flags (= MI_SHAPE_* of include/miwave.h)

struct SceneView {
    const BvhNode *nodes;   uint32_t node_count;
    const Tri     *tris;    uint32_t tri_count;     // BVH leaf order
    const float   *tri_vn;                          // 9 floats per tri (leaf order) or nullptr
    const float   *tri_uv;                          // 6 floats per face (u0 v0 u1 v1 u2 v2), indexed by Tri::prim, or nullptr
    const ShapeRec *shapes; uint32_t shape_count;
    const BsdfRec  *bsdfs;  uint32_t bsdf_count;
    const EmitterRec *emitters; uint32_t emitter_count;
    float emitter_count_f, emitter_count_inv;       // (float) emitter_count and 1 / it (scene_view_prepare: kernel arguments, not per-launch vector math)
    const float *emit_tri;                          // 9 floats per emitter face
    const float *emit_vnorm;                        // 9 floats per emitter face or nullptr
    const float *emit_pmf, *emit_cdf;
    const EnvmapRec *env;                           // environment emitter or nullptr (scene.h:150-151)
    const AnalyticRec *rects;   uint32_t rect_count;    // analytic rectangles (Tri::pad - 1 indexes this table)
    const BitmapRec *bitmaps;                       // bitmap textures (TexRec TEX_BITMAP indexes this table) or nullptr
    const float *bsdf_tables;                       // per-plugin float tables (roughplastic: BsdfRec::p[5] = offset) or nullptr
    float accept_pad;                               // shape.h: the bounds rule of every triangle hit
    const void *tri_bounds;                         // device only: TriBounds per packet of a tiny scene (miwave.hip)
    const void *leaf_boxes;                         // device only: padded SAH leaf boxes of a tiny scene (miwave.hip)
    const struct Bvh4Node *nodes4;                  // device only: the 4-wide quantised tree the phase machine walks (bvh4.h) or nullptr
    const struct Bvh8Node *nodes8;                  // device only: the 8-wide quantised tree (bvh8.h) of a view whose tris / tri_vn are in that tree's order, or nullptr
    const float *env_top; uint32_t env_top_count, env_top_base;   // device only: the environment warp's top levels in LDS (envmap.h: EnvTop); count 0: none
};
MIW_HD EnvTop env_top(const SceneView &sc) { EnvTop t; t.p = sc.env_top; t.count = sc.env_top_count; t.base = sc.env_top_base; return t; }
MIW_HD void scene_view_prepare(SceneView &v) {
    v.emitter_count_f = (float) v.emitter_count;
    v.emitter_count_inv = v.emitter_count ? 1.f / (float) v.emitter_count : 0.f;
}

MIW_HD PrimCtx prim_ctx(const SceneView &sc) { PrimCtx c; c.rects = sc.rects; c.accept_pad = sc.accept_pad; return c; }

// Scene::ray_intersect's second half (scene_native.inl:32-40 -> PreliminaryIntersection::compute_surface_interaction,
// interaction.h:571-596): the SurfaceInteraction of hit record (t, u, v, triangle `tri_idx` in leaf order) of the ray
// (ray_o(), ray_d), plus the two lookups the integrators make on it: si.bsdf() (bsdf.h:485-500) and si.emitter()
// (scene.h:243-253). Analytic = false / Texcoords = false compile the analytic-shape branch / the texture-coordinate
// path out (scenes the caller knows to have none). One definition for the path kernels and for mi_ray_intersect.
template <bool Analytic, bool Texcoords, typename RayO>
MIW_HD void hit_surface_interaction(const SceneView &sc, uint32_t tri_idx, float t, float u, float v, RayO ray_o, V3 ray_d,
                                    SurfaceInteraction &si, uint32_t &bsdf_index, int32_t &emitter) {
    const Tri &tr = sc.tris[tri_idx];
    const ShapeRec &shape = sc.shapes[tr.shape];
    if (Analytic && tr.pad) {                            // analytic shape: its own compute_surface_interaction
        const AnalyticRec &a = sc.rects[tr.pad - 1u];
        if (a.kind == ANALYTIC_SPHERE) compute_surface_interaction_sphere(a, t, ray_o(), ray_d, si);
        else compute_surface_interaction_rect(a, t, u, v, ray_o(), ray_d, si);
    } else {
        const float *vn = (shape.flags & 1u) ? sc.tri_vn + 9 * (size_t) tri_idx : nullptr;
        const float *tc = (Texcoords && (shape.flags & SHAPE_HAS_TEXCOORDS)) ? sc.tri_uv + 6 * (size_t) tr.prim : nullptr;
        compute_surface_interaction(ld3(tr.p0), ld3(tr.p1), ld3(tr.p2), vn, tc, t, u, v, ray_d, si);
    }
    si.shape = tr.shape; si.prim = tr.prim;
    emitter = shape.emitter; bsdf_index = shape.bsdf;
}

struct DirectionSample { V3 p, n, d; float dist, pdf; uint32_t emitter; };

MIW_HD MeshSampler emitter_mesh(const SceneView &sc, const EmitterRec &e) {
    MeshSampler m;
    m.tri = sc.emit_tri + 9 * (size_t) e.tri_first;
    m.vnorm = (e.flags & 1u) ? sc.emit_vnorm + 9 * (size_t) e.tri_first : nullptr;
    m.pmf = sc.emit_pmf + e.tri_first;
    m.cdf = sc.emit_cdf + e.tri_first;
    m.count = e.tri_count;
    m.valid_lo = e.valid_lo; m.valid_hi = e.valid_hi;
    m.sum = e.sum; m.normalization = e.normalization;
    return m;
}

// ---- Sphere as an emitter shape: sphere.cpp:146-275 -----------------------------------------------------------
// warp.h:255-260 (circ(z) = sqrt(1 - z^2), frozen as safe_sqrt(fnmadd(z, z, 1)))
MIW_HD V3 square_to_uniform_sphere(V2 sample) {
    float z = fnmadd(2.f, sample.y, 1.f), r = safe_sqrt(fnmadd(z, z, 1.f)), s, c;
    sincos_((2.f * MIW_PI) * sample.x, s, c);
    return v3(r * c, r * s, z);
}
// Sphere::sample_direction (:169-246): uniform over the cone the sphere subtends from outside, over the surface from
// inside. Fills ds.p, ds.n, ds.d, ds.dist, ds.pdf.
MIW_HD void sphere_sample_direction(const AnalyticRec &r, V3 ref_p, V2 sample, DirectionSample &ds) {
    const V3 center = ld3(r.n);
    const float radius = r.radius;
    V3 dc_v = center - ref_p;
    float dc_2 = squared_norm(dc_v);
    float radius_adj = radius * (r.flip ? (1.f + MIW_RAY_EPSILON) : (1.f - MIW_RAY_EPSILON));
    if (dc_2 > sqr(radius_adj)) {                          // :181-221
        float inv_dc = rsqrt(dc_2),
              sin_theta_max = radius * inv_dc,
              sin_theta_max_2 = sqr(sin_theta_max),
              inv_sin_theta_max = rcp(sin_theta_max),
              cos_theta_max = safe_sqrt(1.f - sin_theta_max_2);
        float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - sqr(fmadd(cos_theta_max - 1.f, sample.x, 1.f))
                                                          : sin_theta_max_2 * sample.x,
              cos_theta = safe_sqrt(1.f - sin_theta_2);
        float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * safe_sqrt(fnmadd(sin_theta_2, sqr(inv_sin_theta_max), 1.f)),
              sin_alpha = safe_sqrt(fnmadd(cos_alpha, cos_alpha, 1.f));
        float sin_phi, cos_phi;
        sincos_(sample.y * (2.f * MIW_PI), sin_phi, cos_phi);
        Frame f; f.n = dc_v * -inv_dc;                     // Frame3f(dc_v * -inv_dc)
        coordinate_system(f.n, f.s, f.t);
        V3 d = to_world(f, v3(cos_phi * sin_alpha, sin_phi * sin_alpha, cos_alpha));
        ds.p = v3(fmadd(d.x, radius, center.x), fmadd(d.y, radius, center.y), fmadd(d.z, radius, center.z));
        ds.n = d;
        ds.d = ds.p - ref_p;
        float dist2 = squared_norm(ds.d);
        ds.dist = __builtin_sqrtf(dist2);
        ds.d = ds.d / ds.dist;
        ds.pdf = (.5f * MIW_INV_PI) / (1.f - cos_theta_max);   // square_to_uniform_cone_pdf, warp.h:481-490
        if (ds.dist == 0.f) ds.pdf = 0.f;
    } else {                                               // :224-236
        V3 d = square_to_uniform_sphere(sample);
        ds.p = v3(fmadd(d.x, radius, center.x), fmadd(d.y, radius, center.y), fmadd(d.z, radius, center.z));
        ds.n = d;
        ds.d = ds.p - ref_p;
        float dist2 = squared_norm(ds.d);
        ds.dist = __builtin_sqrtf(dist2);
        ds.d = ds.d / ds.dist;
        ds.pdf = r.inv_area * dist2 / abs_dot(ds.d, ds.n);
    }
    if (r.flip) ds.n = -ds.n;                              // :243-244
}
// Sphere::pdf_direction, :248-262
MIW_HD float sphere_pdf_direction(const AnalyticRec &r, V3 ref_p, V3 ds_d, float ds_dist, V3 ds_n) {
    float sin_alpha = r.radius * rcp(norm(ld3(r.n) - ref_p)),
          cos_alpha = safe_sqrt(1.f - sin_alpha * sin_alpha);
    return sin_alpha < (1.f - MIW_EPSILON) ? (.5f * MIW_INV_PI) / (1.f - cos_alpha)       // math::OneMinusEpsilon
                                           : r.inv_area * sqr(ds_dist) / abs_dot(ds_d, ds_n);
}

// The environment map behind the Spectrum type of the build. scalar_rgb: the texel colours (envmap.h). scalar_spectral
// (eval_spectrum's spectral branch, envmap.cpp:286-306): a texel holds the three coefficients of the sRGB upsampling model of its
// normalised colour and the scale 2 * hmax(rgb) (the constructor, :101-110; mi_envmap::rgba arrives in that form); the four texels
// around uv are evaluated at the sample's wavelengths, spectra and scales are interpolated separately, and the result is
// s * D65(lambda) * f * m_scale with m_d65 = Texture::D65(1.f).
#if MIW_SPECTRAL
MIW_HD Spec env_eval_uv_spec(const EnvmapRec &e, V2 uv, const Wavelengths &wl) {
    uv.x *= (float) (e.width - 1u); uv.y *= (float) (e.height - 1u);
    uint32_t px = (uint32_t) uv.x, py = (uint32_t) uv.y;
    if (px > e.width - 2u) px = e.width - 2u;
    if (py > e.height - 2u) py = e.height - 2u;
    const float w1x = uv.x - (float) px, w1y = uv.y - (float) py, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const float *p = e.data + 4 * ((size_t) px + (size_t) py * e.width);
    const float *q = p + 4 * (size_t) e.width;
    const float f0 = fmadd(w0x, p[3], w1x * p[7]), f1 = fmadd(w0x, q[3], w1x * q[7]);      // :296-297
    const float f = fmadd(w0y, f0, w1y * f1);                                              // :300
    Spec r;
    for (int i = 0; i < MIW_SPEC_N; ++i) {
        const float l = wl.l[i];
        const float s00 = srgb_model_eval(p, l), s10 = srgb_model_eval(p + 4, l),          // :289-292
                    s01 = srgb_model_eval(q, l), s11 = srgb_model_eval(q + 4, l);
        const float s0 = fmadd(w0x, s00, w1x * s10), s1 = fmadd(w0x, s01, w1x * s11);      // :294-295
        const float sv = fmadd(w0y, s0, w1y * s1);                                         // :299
        const float wp = d65_eval(1.f * (1.f / 10568.f), l);                               // :303-305, d65.cpp:61-62
        r.c[i] = sv * wp * f * e.scale;                                                    // :307
    }
    return r;
}
MIW_HD Spec env_eval_spec(const EnvmapRec &e, V3 d_world, const Wavelengths &wl) {
    return env_eval_uv_spec(e, env_dir_to_uv(xf_vector(e.to_local, d_world)), wl);        // envmap.cpp:134-146
}
// EnvironmentMapEmitter::sample_direction, envmap.cpp:157-190: env_sample_direction of envmap.h with the spectral lookup
MIW_HD Spec env_sample_direction_spec(const EnvmapRec &e, V3 ref_p, V2 sample, V3 &d_out, float &dist_out, float &pdf_out,
                                      V3 &p_out, V3 &n_out, const Wavelengths &wl, EnvTop top = env_top_none()) {
    float pdf;
    V2 uv = hier2d_sample(e, sample, pdf, top);
    float theta = uv.y * MIW_PI, phi = uv.x * (2.f * MIW_PI);
    float st, ct, sp, cp;
    sincos_(theta, st, ct); sincos_(phi, sp, cp);
    V3 d = v3(cp * st, sp * st, ct);
    d = v3(d.y, d.z, -d.x);
    float dist = 2.f * e.radius;
    float inv_sin_theta = env_inv_sin_theta(d);
    d = xf_vector(e.to_world, d);
    p_out = ref_p + d * dist;
    n_out = -d;
    float ds_pdf = pdf > 0.f ? pdf * inv_sin_theta * (1.f / (2.f * sqr(MIW_PI))) : 0.f;
    d_out = d; dist_out = dist; pdf_out = ds_pdf;
    return env_eval_uv_spec(e, uv, wl) / ds_pdf;
}
#else
MIW_HD Spec env_sample_direction_spec(const EnvmapRec &e, V3 ref_p, V2 sample, V3 &d, float &dist, float &pdf, V3 &p, V3 &n, const Wavelengths &,
                                      EnvTop top = env_top_none()) {
    return env_sample_direction(e, ref_p, sample, d, dist, pdf, p, n, top);
}
MIW_HD Spec env_eval_spec(const EnvmapRec &e, V3 d, const Wavelengths &) { return env_eval(e, d); }
#endif

// Endpoint::sample_direction of emitter `index` (endpoint.h:119-139): AreaLight (area.cpp:121-166 +
// shape.cpp:292-309) or the environment map (envmap.cpp:157-190). Returns radiance / pdf; `ds.pdf == 0`: no sample.
// Analytic = false: the caller knows the scene to hold no analytic shapes, hence no sphere / rectangle lights (their code is compiled out).
template <bool Analytic = true>
MIW_HD Spec emitter_sample_direction(const SceneView &sc, uint32_t index, V3 ref_p, V2 sample, DirectionSample &ds, const Wavelengths &wl) {
    const EmitterRec &e = sc.emitters[index];
    Spec value;
    ds.emitter = index;
    if (e.type == EMITTER_ENVMAP) {
        value = env_sample_direction_spec(*sc.env, ref_p, sample, ds.d, ds.dist, ds.pdf, ds.p, ds.n, wl, env_top(sc));
    } else {
        if (Analytic && (e.flags & 2u) && sc.rects[e.tri_first].kind == ANALYTIC_SPHERE) {
            sphere_sample_direction(sc.rects[e.tri_first], ref_p, sample, ds);      // the sphere's own sample_direction
        } else {
            // Shape::sample_direction, shape.cpp:292-309
            PositionSample ps = (Analytic && (e.flags & 2u)) ? rect_sample_position(sc.rects[e.tri_first], sample)
                                                             : mesh_sample_position(emitter_mesh(sc, e), sample);
            ds.p = ps.p; ds.n = ps.n; ds.pdf = ps.pdf;
            ds.d = ds.p - ref_p;
            float dist_squared = squared_norm(ds.d);
            ds.dist = __builtin_sqrtf(dist_squared);
            ds.d = ds.d / ds.dist;
            float dp = abs_dot(ds.d, ds.n);
            ds.pdf *= (dp != 0.f) ? dist_squared / dp : 0.f;
        }
        // AreaLight::sample_direction, area.cpp:131-136,165
        bool active = dot(ds.d, ds.n) < 0.f && ds.pdf != 0.f;
        value = tex_eval(e.radiance, wl) / ds.pdf;
        if (!active) value = spec(0.f);
    }
    return value;
}

// scene.cpp:164-200, *without* the visibility test (the shadow ray is a separate stage of the kernels;
// mi_sample_emitter_direction traces it on request). Returns the unoccluded emitter value; `ds.pdf == 0`
// means "no sample" (path.cpp:160).
template <bool Analytic = true>
MIW_HD Spec sample_emitter_direction(const SceneView &sc, V3 ref_p, V2 sample, DirectionSample &ds, const Wavelengths &wl) {
    if (sc.emitter_count == 0) {                       // scene.cpp:208-211
        ds.p = ds.n = ds.d = v3(0.f); ds.dist = 0.f; ds.pdf = 0.f; ds.emitter = 0;
        return spec(0.f);
    }
    uint32_t index = 0;
    float emitter_pdf = 1.f;
    if (sc.emitter_count > 1) {                        // scene.cpp:180-188
        float n = sc.emitter_count_f;
        emitter_pdf = sc.emitter_count_inv;
        uint32_t i = (uint32_t) (sample.x * n);
        index = i < sc.emitter_count - 1 ? i : sc.emitter_count - 1;
        sample.x = (sample.x - (float) index * emitter_pdf) * n;
    }
    Spec value = emitter_sample_direction<Analytic>(sc, index, ref_p, sample, ds, wl);
    if (sc.emitter_count > 1) {                        // scene.cpp:195-197
        ds.pdf *= emitter_pdf;
        value = value * rcp(emitter_pdf);
    }
    return value;
}

// Endpoint::pdf_direction of emitter `emitter` (area.cpp:168-187 + shape.cpp:311-323, envmap.cpp:192-208).
// `ds_d`, `ds_dist`, `ds_n` come from DirectionSample(si_bsdf, si) (records.h:167-173).
// `ref_p` = it.p, the point the direction leaves from (only the sphere's pdf_direction needs it).
template <bool Analytic = true>
MIW_HD float emitter_pdf_direction(const SceneView &sc, uint32_t emitter, V3 ds_d, float ds_dist, V3 ds_n, V3 ref_p) {
    const EmitterRec &e = sc.emitters[emitter];
    if (e.type == EMITTER_ENVMAP) return env_pdf_direction(*sc.env, ds_d);
    float dp = dot(ds_d, ds_n);
    bool active = dp < 0.f;
    float pdf;
    if (Analytic && (e.flags & 2u) && sc.rects[e.tri_first].kind == ANALYTIC_SPHERE) {
        pdf = sphere_pdf_direction(sc.rects[e.tri_first], ref_p, ds_d, ds_dist, ds_n);
    } else {
        pdf = e.normalization;
        float adp = abs_dot(ds_d, ds_n);
        pdf *= (adp != 0.f) ? (ds_dist * ds_dist) / adp : 0.f;
    }
    return active ? pdf : 0.f;
}
// scene.cpp:216-231
template <bool Analytic = true>
MIW_HD float pdf_emitter_direction(const SceneView &sc, uint32_t emitter, V3 ds_d, float ds_dist, V3 ds_n, V3 ref_p) {
    float value = emitter_pdf_direction<Analytic>(sc, emitter, ds_d, ds_dist, ds_n, ref_p);
    if (sc.emitter_count > 1) value = value * sc.emitter_count_inv;
    return value;
}

// AreaLight::eval, area.cpp:63-71
MIW_HD Spec emitter_eval(const EmitterRec &e, V3 wi, const Wavelengths &wl) {
    return wi.z > 0.f ? tex_eval(e.radiance, wl) : spec(0.f);
}


// path.cpp:223-227
MIW_HD float mis_weight(float pdf_a, float pdf_b) {
    pdf_a *= pdf_a; pdf_b *= pdf_b;
    return pdf_a > 0.f ? pdf_a / (pdf_a + pdf_b) : 0.f;
}

// ---- perspective sensor -------------------------------------------------------------------
// The host precomputes sample_to_camera (sensor.h:196-231, inverse) and the
// camera-to-world matrix (transform.h:241-269) once; perspective.cpp:182-216
// is what runs per sample. `position_sample` is already divided by the crop
// size (integrator.cpp:254-256).
struct SensorRec {
    float sample_to_camera[16];
    float to_world[16];
    float near_clip, far_clip;
    float pp_offset[2];          // principal point offset (perspective.cpp:111-116)
};

// perspective.cpp:186-214. `origin` = sensor_origin(s): the same for every ray, so the render kernels take it from a kernel
// argument (RenderParams::cam_o) instead of holding three vector registers of loop-invariant fmas.
MIW_HD V3 sensor_origin(const SensorRec &s) { return xf_point_affine(s.to_world, v3(0.f)); }
MIW_HD Ray sensor_sample_ray(const SensorRec &s, V2 position_sample, V3 origin) {
    V3 near_p = xf_point_persp(s.sample_to_camera,
                               v3(position_sample.x + s.pp_offset[0],
                                  position_sample.y + s.pp_offset[1], 0.f));
    V3 d = normalize(near_p);
    float inv_z = rcp(d.z);
    Ray r;
    r.mint = s.near_clip * inv_z;
    r.maxt = s.far_clip * inv_z;
    r.o = origin;
    r.d = xf_vector(s.to_world, d);
    return r;
}
MIW_HD Ray sensor_sample_ray(const SensorRec &s, V2 position_sample) { return sensor_sample_ray(s, position_sample, sensor_origin(s)); }

} // namespace miw
