// Environment map emitter (src/emitters/envmap.cpp) and the hierarchical sample
// warp it importance-samples with (Hierarchical2D<Float, 0>,
// include/mitsuba/core/distr_2d.h:336-747; bilinear patch warps
// include/mitsuba/core/warp.h:358-434). scalar_rgb branch: texel = linear RGB,
// alpha unused, no spectral upsampling.
//
// The MIP hierarchy is built once on the host exactly as the reference builds
// it (distr_2d.h:372-462: patch averages in float, sum in double, level k+1 =
// sums of 2x2 of level k, 2x2 patches stored contiguously) and handed over as
// one float array + per-level offsets; everything below runs per sample.
#pragma once
#include "base.h"
#include "warp.h"
#include "special.h"

namespace miw {

#define MIW_ENV_MAX_LEVELS 20
#define MIW_INV_TWO_PI 0.15915494309189533577f

struct EnvmapRec {
    const float *data;          // width * height * 4 (RGBA, row-major) — m_data
    const float *levels;        // Hierarchical2D storage: level 0 .. n_levels-1, concatenated
    uint32_t width, height;
    uint32_t n_levels;
    uint32_t level_offset[MIW_ENV_MAX_LEVELS];   // float offset of level l in `levels`
    uint32_t level_width[MIW_ENV_MAX_LEVELS];    // Level::width
    uint32_t level_height[MIW_ENV_MAX_LEVELS];   // rows of the level's array (even for levels >= 1: zero-padded like the widths)
    float patch_size[2], inv_patch_size[2];      // distr_2d.h:226-227
    uint32_t max_patch_index[2];                 // :385
    float scale;                // m_scale
    float to_world[16], to_local[16];            // m_world_transform and its inverse, column-major
    float radius;               // m_bsphere.radius after set_scene (envmap.cpp:128-132)
    uint32_t emitter_index;     // position in Scene::m_emitters
};

// warp.h:360-366
MIW_HD float interval_to_linear(float v0, float v1, float sample) {
    return abs_(v0 - v1) > 1e-4f * (v0 + v1)
        ? (v0 - safe_sqrt(lerp_(sqr(v0), sqr(v1), sample))) / (v0 - v1)
        : sample;
}
// warp.h:393-407
MIW_HD V2 square_to_bilinear(float v00, float v10, float v01, float v11, V2 sample, float &pdf) {
    float r0 = v00 + v10, r1 = v01 + v11;
    sample.y = interval_to_linear(r0, r1, sample.y);
    float c0 = lerp_(v00, v01, sample.y),
          c1 = lerp_(v10, v11, sample.y);
    sample.x = interval_to_linear(c0, c1, sample.x);
    pdf = lerp_(c0, c1, sample.x);
    return sample;
}
// warp.h:429-434
MIW_HD float square_to_bilinear_pdf(float v00, float v10, float v01, float v11, V2 sample) {
    return lerp_(lerp_(v00, v10, sample.x), lerp_(v01, v11, sample.x), sample.y);
}

// Level::index, distr_2d.h:724-727 (2x2 patches contiguous)
MIW_HD uint32_t hier2d_index(uint32_t x, uint32_t y, uint32_t width) {
    return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width);
}

// A second copy of the hierarchy's TOP levels (the smallest ones: levels n_levels - count .. n_levels - 1, `base` = float offset
// of the first of them in EnvmapRec::levels): the device kernels that stage tables in LDS (device/trace.h: stage_tables) read the
// first steps of the warp from there — the same floats, so the same sample. count == 0 (everywhere else): no such copy.
struct EnvTop { const float *p; uint32_t count, base; };
MIW_HD EnvTop env_top_none() { EnvTop t; t.p = nullptr; t.count = 0; t.base = 0; return t; }

// Hierarchical2D::sample, distr_2d.h:470-558 (Dimension = 0: no conditional parameters)
//
// The descent is a chain of DEPENDENT lookups — which 2x2 block of level l - 1 is read depends on the choice made at level l — and
// on the device every link below the levels held in LDS is a round trip through memory: six of them for the 1024 x 512 map of
// BASELINE config 4, the longest chain of the phase machine's shade body. Two facts shorten it without touching
// a single float of the result (MIW_ENV_PAIRED, round 5):
//   * a level l >= 2 holds the sums of the 2x2 blocks of level l - 1, computed by the constructor as d[0] + d[1] + d[2] + d[3] in
//     float32 (distr_2d.h:445-461, envmap_build.h) — so the four values of a block of level l can be RECOMPUTED from the four child
//     blocks of level l - 1 (a 4x4 region: two runs of eight consecutive floats in the 2x2-blocked layout), which are exactly what
//     the next step reads one of. One fetch of 16 floats serves levels l and l - 1;
//   * the step from level 1 (patch averages) to the bilinear patch of level 0 reads four texels of the data array around the
//     chosen patch: the 3 x 3 texels around the level-1 block cover all four candidates and are fetched WITH that block.
// Blocks beyond a level's (zero-padded, even) extent contribute zeros, as the constructor's loops leave them.
#ifndef MIW_ENV_PAIRED
#define MIW_ENV_PAIRED 1
#endif
template <bool Paired = (MIW_ENV_PAIRED != 0)>
MIW_HD V2 hier2d_sample(const EnvmapRec &e, V2 sample, float &pdf, EnvTop top = env_top_none()) {
    sample.x = clamp_(sample.x, 0.f, 1.f); sample.y = clamp_(sample.y, 0.f, 1.f);
    uint32_t ox = 0, oy = 0;
    auto step = [&](float v00, float v10, float v01, float v11, uint32_t &dx, uint32_t &dy) {   // one level's choice, :487-527
        sample.x = clamp_(sample.x, 0.f, 1.f); sample.y = clamp_(sample.y, 0.f, 1.f);
        // select the row
        float r0 = v00 + v10, r1 = v01 + v11;
        sample.y *= r0 + r1;
        bool mask = sample.y > r0;
        dy = mask ? 1u : 0u;
        if (mask) sample.y -= r0;
        sample.y /= mask ? r1 : r0;
        // select the column
        float c0 = mask ? v01 : v00, c1 = mask ? v11 : v10;
        sample.x *= c0 + c1;
        mask = sample.x > c0;
        if (mask) sample.x -= c0;
        sample.x /= mask ? c1 : c0;
        dx = mask ? 1u : 0u;
    };
    auto descend = [&](const float *lv, uint32_t width) {
        ox <<= 1; oy <<= 1;
        uint32_t i = hier2d_index(ox, oy, width), dx, dy;
        step(lv[i], lv[i + 1], lv[i + 2], lv[i + 3], dx, dy);
        ox += dx; oy += dy;
    };
    int l = (int) e.n_levels - 2;
    for (; l > 0 && l >= (int) e.n_levels - (int) top.count; --l) descend(top.p + (e.level_offset[l] - top.base), e.level_width[l]);
    const float *l0 = e.levels + e.level_offset[0];
    const uint32_t w = e.level_width[0];
    float v00 = 0.f, v10 = 0.f, v01 = 0.f, v11 = 0.f;
    bool have_corners = false;
    if (Paired) {
        for (; l > 2; l -= 2) {                                     // levels l and l - 1 (l - 1 >= 2) from one 4x4 region of level l - 1
            ox <<= 1; oy <<= 1;
            const float *lm = e.levels + e.level_offset[l - 1];
            const uint32_t wm = e.level_width[l - 1], hm = e.level_height[l - 1];
            const uint32_t cx = 2u * ox, cy = 2u * oy;             // the child region's origin at level l - 1: blocks (cx + 2 dx, cy + 2 dy)
            const bool in_x1 = cx + 2u < wm, in_y1 = cy + 2u < hm;  // (cx < wm and cy < hm always: the block at level l exists)
            // rows cy / cy + 1: eight consecutive floats = blocks dx = 0, 1; rows cy + 2 / cy + 3 likewise (addresses clamped into the level)
            const float *ra = lm + hier2d_index(cx, cy, wm), *rb = lm + hier2d_index(cx, in_y1 ? cy + 2u : cy, wm);
            const uint32_t o1 = in_x1 ? 4u : 0u;
            // (four 16-byte reads: the levels >= 1 start on 16-byte boundaries — envmap_build pads level 0 — and a block pair starts on a multiple of eight floats)
            const F4 qa0 = *reinterpret_cast<const F4 *>(ra), qa1 = *reinterpret_cast<const F4 *>(ra + o1),
                     qb0 = *reinterpret_cast<const F4 *>(rb), qb1 = *reinterpret_cast<const F4 *>(rb + o1);
            float a[8] = { qa0.x, qa0.y, qa0.z, qa0.w, qa1.x, qa1.y, qa1.z, qa1.w }, b[8] = { qb0.x, qb0.y, qb0.z, qb0.w, qb1.x, qb1.y, qb1.z, qb1.w };
            for (int k = 0; k < 4; ++k) { a[4 + k] = in_x1 ? a[4 + k] : 0.f; b[k] = in_y1 ? b[k] : 0.f; b[4 + k] = (in_x1 && in_y1) ? b[4 + k] : 0.f; }
            uint32_t dx, dy;
            step(((a[0] + a[1]) + a[2]) + a[3], ((a[4] + a[5]) + a[6]) + a[7], ((b[0] + b[1]) + b[2]) + b[3], ((b[4] + b[5]) + b[6]) + b[7], dx, dy);
            ox += dx; oy += dy;
            ox <<= 1; oy <<= 1;                                     // level l - 1: the chosen child block is in registers
            float c[4];
            for (int k = 0; k < 4; ++k) c[k] = dy ? (dx ? b[4 + k] : b[k]) : (dx ? a[4 + k] : a[k]);
            step(c[0], c[1], c[2], c[3], dx, dy);
            ox += dx; oy += dy;
        }
        if (l == 2) { descend(e.levels + e.level_offset[2], e.level_width[2]); l = 1; }
        if (l == 1) {                                                // level 1's block together with the 3 x 3 data texels under its four patches
            ox <<= 1; oy <<= 1;
            const float *l1 = e.levels + e.level_offset[1];
            const uint32_t i1 = hier2d_index(ox, oy, e.level_width[1]), h0 = e.level_height[0];
            float t[3][3];
            for (uint32_t r = 0; r < 3u; ++r) {
                const uint32_t yy = oy + r < h0 ? oy + r : h0 - 1u;  // (a patch on a padded row / column carries no mass and is never chosen)
                for (uint32_t q = 0; q < 3u; ++q) { const uint32_t xx = ox + q < w ? ox + q : w - 1u; t[r][q] = l0[xx + yy * w]; }
            }
            uint32_t dx, dy;
            step(l1[i1], l1[i1 + 1], l1[i1 + 2], l1[i1 + 3], dx, dy);
            ox += dx; oy += dy;
            v00 = dy ? (dx ? t[1][1] : t[1][0]) : (dx ? t[0][1] : t[0][0]); v10 = dy ? (dx ? t[1][2] : t[1][1]) : (dx ? t[0][2] : t[0][1]);
            v01 = dy ? (dx ? t[2][1] : t[2][0]) : (dx ? t[1][1] : t[1][0]); v11 = dy ? (dx ? t[2][2] : t[2][1]) : (dx ? t[1][2] : t[1][1]);
            l = 0; have_corners = true;
        }
    }
    for (; l > 0; --l) descend(e.levels + e.level_offset[l], e.level_width[l]);
    if (!have_corners) {                                            // the bilinear patch's corners were not fetched with level 1
        const uint32_t i = ox + oy * w;
        v00 = l0[i]; v10 = l0[i + 1]; v01 = l0[i + w]; v11 = l0[i + w + 1];
    }
    sample = square_to_bilinear(v00, v10, v01, v11, sample, pdf);
    return v2(((float) (int32_t) ox + sample.x) * e.patch_size[0],
              ((float) (int32_t) oy + sample.y) * e.patch_size[1]);
}

// Hierarchical2D::eval, distr_2d.h:650-680
MIW_HD float hier2d_eval(const EnvmapRec &e, V2 pos) {
    pos.x = clamp_(pos.x, 0.f, 1.f); pos.y = clamp_(pos.y, 0.f, 1.f);
    pos.x *= e.inv_patch_size[0]; pos.y *= e.inv_patch_size[1];
    uint32_t ox = (uint32_t) (int32_t) pos.x, oy = (uint32_t) (int32_t) pos.y;
    if (ox > e.max_patch_index[0]) ox = e.max_patch_index[0];
    if (oy > e.max_patch_index[1]) oy = e.max_patch_index[1];
    pos.x -= (float) (int32_t) ox; pos.y -= (float) (int32_t) oy;
    const float *l0 = e.levels + e.level_offset[0];
    uint32_t w = e.level_width[0], i = ox + oy * w;
    return square_to_bilinear_pdf(l0[i], l0[i + 1], l0[i + w], l0[i + w + 1], pos);
}

// eval_spectrum, envmap.cpp:269-320 (RGB branch :309-319)
MIW_HD V3 env_eval_uv(const EnvmapRec &e, V2 uv) {
    uv.x *= (float) (e.width - 1u); uv.y *= (float) (e.height - 1u);
    uint32_t px = (uint32_t) uv.x, py = (uint32_t) uv.y;
    if (px > e.width - 2u) px = e.width - 2u;
    if (py > e.height - 2u) py = e.height - 2u;
    float w1x = uv.x - (float) px, w1y = uv.y - (float) py, w0x = 1.f - w1x, w0y = 1.f - w1y;
    const float *p = e.data + 4 * ((size_t) px + (size_t) py * e.width);
    const float *q = p + 4 * (size_t) e.width;
    V3 v00 = ld3(p), v10 = ld3(p + 4), v01 = ld3(q), v11 = ld3(q + 4);
    V3 v0 = v3(fmadd(w0x, v00.x, w1x * v10.x), fmadd(w0x, v00.y, w1x * v10.y), fmadd(w0x, v00.z, w1x * v10.z)),
       v1 = v3(fmadd(w0x, v01.x, w1x * v11.x), fmadd(w0x, v01.y, w1x * v11.y), fmadd(w0x, v01.z, w1x * v11.z)),
       v  = v3(fmadd(w0y, v0.x, w1y * v1.x), fmadd(w0y, v0.y, w1y * v1.y), fmadd(w0y, v0.z, w1y * v1.z));
    return v * e.scale;
}

// direction (emitter-local) -> lat-long texture coordinates, envmap.cpp:140-143
MIW_HD V2 env_dir_to_uv(V3 v) {
    V2 uv = v2(atan2_(v.x, -v.z) * MIW_INV_TWO_PI, safe_acos(v.y) * MIW_INV_PI);
    uv.x -= __builtin_floorf(uv.x); uv.y -= __builtin_floorf(uv.y);
    return uv;
}

// EnvironmentMapEmitter::eval, envmap.cpp:134-146. `d_world` = -si.wi (= ray.d of the missing ray)
MIW_HD V3 env_eval(const EnvmapRec &e, V3 d_world) {
    return env_eval_uv(e, env_dir_to_uv(xf_vector(e.to_local, d_world)));
}

// safe_rsqrt(max(d.x^2 + d.z^2, eps^2)), envmap.cpp:171-172 / :201-202
MIW_HD float env_inv_sin_theta(V3 d) {
    float v = max_(sqr(d.x) + sqr(d.z), sqr(MIW_EPSILON));
    return rsqrt(max_(v, 0.f));
}

// EnvironmentMapEmitter::sample_direction, envmap.cpp:157-190. Returns radiance / pdf.
MIW_HD V3 env_sample_direction(const EnvmapRec &e, V3 ref_p, V2 sample, V3 &d_out, float &dist_out, float &pdf_out,
                               V3 &p_out, V3 &n_out, EnvTop top = env_top_none()) {
    float pdf;
    V2 uv = hier2d_sample(e, sample, pdf, top);
    float theta = uv.y * MIW_PI, phi = uv.x * (2.f * MIW_PI);
    float st, ct, sp, cp;
    sincos_(theta, st, ct); sincos_(phi, sp, cp);          // math::sphdir, math.h:48-57
    V3 d = v3(cp * st, sp * st, ct);
    d = v3(d.y, d.z, -d.x);
    float dist = 2.f * e.radius;
    float inv_sin_theta = env_inv_sin_theta(d);
    d = xf_vector(e.to_world, d);
    p_out = ref_p + d * dist;
    n_out = -d;
    float ds_pdf = pdf > 0.f ? pdf * inv_sin_theta * (1.f / (2.f * sqr(MIW_PI))) : 0.f;
    d_out = d; dist_out = dist; pdf_out = ds_pdf;
    return env_eval_uv(e, uv) / ds_pdf;
}

// EnvironmentMapEmitter::pdf_direction, envmap.cpp:192-208
MIW_HD float env_pdf_direction(const EnvmapRec &e, V3 d_world) {
    V3 d = xf_vector(e.to_local, d_world);
    V2 uv = env_dir_to_uv(d);
    float inv_sin_theta = env_inv_sin_theta(d);
    return hier2d_eval(e, uv) * inv_sin_theta * (1.f / (2.f * sqr(MIW_PI)));
}

} // namespace miw
