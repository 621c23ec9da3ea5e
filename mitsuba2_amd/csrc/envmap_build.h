// Host-side construction of the environment map tables (EnvironmentMapEmitter ctor,
// src/emitters/envmap.cpp:67-128, + Hierarchical2D ctor, include/mitsuba/core/distr_2d.h:372-462),
// shared by the device uploader and the CPU checker so both sample the very same floats.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include "miw/envmap.h"
#include "../../include/miwave.h"

namespace miw {

struct EnvmapTables {
    std::vector<float> data;        // rgba copy
    std::vector<float> levels;      // concatenated Hierarchical2D levels
    EnvmapRec rec{};                // data / levels pointers are filled by the caller
    bool ok = false;
};

// mitsuba::luminance(Color3f), include/mitsuba/core/spectrum.h (Rec. 709 weights)
inline float env_luminance(const float *c) { return c[0] * 0.212671f + c[1] * 0.715160f + c[2] * 0.072169f; }

// general 4x4 inverse of a column-major matrix (Transform4f::inverse: the reference keeps an
// analytically composed inverse; for the rigid transforms an envmap takes the two agree)
inline bool invert4(const float *m, float *out) {
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = m[c * 4 + r]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        for (int r = i + 1; r < 4; ++r) if (std::fabs(a[r][i]) > std::fabs(a[piv][i])) piv = r;
        if (a[piv][i] == 0.0) return false;
        for (int c = 0; c < 8; ++c) std::swap(a[i][c], a[piv][c]);
        double d = a[i][i];
        for (int c = 0; c < 8; ++c) a[i][c] /= d;
        for (int r = 0; r < 4; ++r) if (r != i) { double f = a[r][i]; for (int c = 0; c < 8; ++c) a[r][c] -= f * a[i][c]; }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[c * 4 + r] = (float) a[r][4 + c];
    return true;
}

// `density`: the W x H array the warp is built from; nullptr = luminance * sin(theta) of the bitmap
// (envmap.cpp:82-116). A caller-supplied density exists for the Hierarchical2D known-answer tests.
inline EnvmapTables envmap_build_from_density(const mi_envmap &e, const float *density) {
    EnvmapTables t;
    const uint32_t W = e.width, H = e.height;
    if (!e.rgba || W < 2 || H < 2) return t;
    t.data.assign(e.rgba, e.rgba + (size_t) W * H * 4);
    std::vector<float> lum((size_t) W * H);
    for (uint32_t y = 0; y < H; ++y) {
        float sin_theta = std::sin((float) y / (float) (H - 1) * 3.14159265358979323846f);
        for (uint32_t x = 0; x < W; ++x)
            lum[(size_t) y * W + x] = density ? density[(size_t) y * W + x]
                                              : env_luminance(&t.data[4 * ((size_t) y * W + x)]) * sin_theta;
    }
    // Hierarchical2D(data, size), normalize = true, enable_sampling = true
    EnvmapRec &r = t.rec;
    std::memset(&r, 0, sizeof r);
    const uint32_t npx = W - 1, npy = H - 1;                       // n_patches
    uint32_t hm = npx > npy ? npx : npy, max_level = 0;
    while ((1u << max_level) < hm) ++max_level;                    // math::log2i_ceil
    r.max_patch_index[0] = npx - 1; r.max_patch_index[1] = npy - 1;
    r.patch_size[0] = 1.f / (float) npx; r.patch_size[1] = 1.f / (float) npy;
    r.inv_patch_size[0] = (float) npx; r.inv_patch_size[1] = (float) npy;
    struct L { uint32_t w, h; };
    std::vector<L> dims; dims.push_back({ W, H });
    uint32_t lw = npx, lh = npy;
    for (int level = (int) max_level; level >= 0; --level) {
        lw += lw & 1u; lh += lh & 1u;                              // zero-pad
        dims.push_back({ lw, lh });
        lw >>= 1; lh >>= 1;
    }
    if (dims.size() > MIW_ENV_MAX_LEVELS) return t;
    r.n_levels = (uint32_t) dims.size();
    size_t total = 0;
    // (level 0, the W x H data array, is padded to a multiple of four floats so that every level >= 1 — even widths and heights —
    // starts on a 16-byte boundary: hier2d_sample reads block pairs of those levels as 16-byte words)
    for (size_t l = 0; l < dims.size(); ++l) {
        r.level_offset[l] = (uint32_t) total; r.level_width[l] = dims[l].w; r.level_height[l] = dims[l].h;
        total += ((size_t) dims[l].w * dims[l].h + 3u) / 4u * 4u;
    }
    t.levels.assign(total, 0.f);
    float *l0 = t.levels.data() + r.level_offset[0], *l1 = t.levels.data() + r.level_offset[1];
    // integrate the linear interpolant, distr_2d.h:424-436
    double sum = 0.0;
    const float *in = lum.data();
    for (uint32_t y = 0; y < npy; ++y) {
        for (uint32_t x = 0; x < npx; ++x) {
            float avg = (in[0] + in[1] + in[W] + in[W + 1]) * .25f;
            sum += (double) avg;
            l1[hier2d_index(x, y, dims[1].w)] = avg;
            ++in;
        }
        ++in;
    }
    if (!(sum > 0.0)) return t;
    float scale = (float) ((double) ((uint64_t) npx * npy) / sum);   // hprod(n_patches) / sum
    for (size_t i = 0; i < (size_t) W * H; ++i) l0[i] = lum[i] * scale;
    for (size_t i = 0; i < (size_t) dims[1].w * dims[1].h; ++i) l1[i] *= scale;
    // MIP hierarchy, :445-461
    uint32_t sx = npx, sy = npy;
    for (uint32_t level = 2; level <= max_level + 1; ++level) {
        const float *d0l = t.levels.data() + r.level_offset[level - 1];
        float *d1l = t.levels.data() + r.level_offset[level];
        sx = (sx + 1u) >> 1; sy = (sy + 1u) >> 1;
        for (uint32_t y = 0; y < sy; ++y)
            for (uint32_t x = 0; x < sx; ++x) {
                const float *d0 = d0l + hier2d_index(x * 2, y * 2, dims[level - 1].w);
                d1l[hier2d_index(x, y, dims[level].w)] = d0[0] + d0[1] + d0[2] + d0[3];
            }
    }
    r.width = W; r.height = H; r.scale = e.scale;
    std::memcpy(r.to_world, e.to_world, 64);
    if (!invert4(e.to_world, r.to_local)) return t;
    // set_scene, envmap.cpp:128-132
    float rad = e.bsphere_radius * (1.f + MIW_RAY_EPSILON);
    r.radius = MIW_RAY_EPSILON > rad ? MIW_RAY_EPSILON : rad;
    r.emitter_index = e.emitter_index;
    t.ok = true;
    return t;
}

// mi_envmap::density is the scalar_spectral library's input (there the texels are coefficients: include/miwave.h). The scalar_rgb
// library computes the array from `rgba` and does NOT read the field: a C caller compiled against the round-3 header (no such
// member) or with an uninitialised struct hands over whatever lies behind bsphere_radius / emitter_index.
inline EnvmapTables envmap_build(const mi_envmap &e) {
#if MIW_SPECTRAL
    return envmap_build_from_density(e, e.density);
#else
    return envmap_build_from_density(e, nullptr);
#endif
}

} // namespace miw
