// ImageBlock::put — filtered splat of one sample into the film accumulator.
//
// Follows src/librender/imageblock.cpp:79-172 with the block geometry of
// src/librender/integrator.cpp:114-130 (one bordered block per 32x32 spiral
// tile, border = filter border) and the block->film accumulation of
// imageblock.cpp:49-77 / bitmap.h:657-712 (out-of-film border texels clipped).
//
// The sample position is converted to *block-local* coordinates exactly as the
// reference does (float32, :114), so the discretised filter weights are the
// reference's bit for bit; the accumulation target is the global film.
// Accumulation happens in float64 (HBM atomics on the device): 1080p@512spp adds
// ~8000 terms per texel, and float32 accumulation order alone would cost
// ~2e-6 relative error — the f64 sum is order-independent at float32 precision.
#pragma once
#include "base.h"

namespace miw {

#define MIW_FILTER_RESOLUTION 31        /* include/mitsuba/core/rfilter.h: MTS_FILTER_RESOLUTION */
#define MIW_FILM_CHANNELS 5             /* X Y Z A W, integrator.cpp:71-72 */

struct FilmRec {
    int32_t crop_w, crop_h;             // film->crop_size()
    int32_t crop_x, crop_y;             // film->crop_offset()
    int32_t block_size;                 // m_block_size (integrator.cpp:88-97)
    int32_t border;                     // rfilter border_size (rfilter.cpp:19)
    float radius;                       // rfilter radius
    float scale_factor;                 // MTS_FILTER_RESOLUTION / radius (rfilter.cpp:18)
    float lut[MIW_FILTER_RESOLUTION + 1];
};

// rfilter.h:62-65
MIW_HD float filter_eval_discretized(const FilmRec &f, float x) {
    int index = (int) abs_(x * f.scale_factor);
    if (index > MIW_FILTER_RESOLUTION) index = MIW_FILTER_RESOLUTION;
    return f.lut[index];
}

MIW_HD int ceil2int(float x)  { return (int) __builtin_ceilf(x); }
MIW_HD int floor2int(float x) { return (int) __builtin_floorf(x); }

// The sample-validity test of imageblock.cpp:85-109 (invalid samples are
// dropped, not fatal).
MIW_HD bool sample_is_valid(const float *value) {
    bool ok = true;
    for (int k = 0; k < MIW_FILM_CHANNELS; ++k)
        ok = ok && (value[k] >= -1e-5f) && isfinite_(value[k]);
    return ok;
}

// Calls add(texel_index, channel, value) for every texel the sample touches,
// in the reference's loop order (:148-161). `px,py` = integer pixel the sample
// belongs to (selects the spiral block), `pos` = position_sample.
template <typename Add>
MIW_HD void film_splat(const FilmRec &f, int px, int py, V2 pos_, const float *value, Add add) {
    if (!sample_is_valid(value)) return;

    // block this pixel lives in (spiral.cpp:43-45): offset, clipped size
    int bx = ((px - f.crop_x) / f.block_size) * f.block_size,
        by = ((py - f.crop_y) / f.block_size) * f.block_size;
    int bw = f.crop_w - bx < f.block_size ? f.crop_w - bx : f.block_size,
        bh = f.crop_h - by < f.block_size ? f.crop_h - by : f.block_size;
    int off_x = bx + f.crop_x, off_y = by + f.crop_y;
    int size_x = bw + 2 * f.border, size_y = bh + 2 * f.border;

    // :114  pos = pos_ - (m_offset - m_border_size + .5f)
    float posx = pos_.x - ((float) (off_x - f.border) + .5f),
          posy = pos_.y - ((float) (off_y - f.border) + .5f);

    if (f.radius > 0.5f + MIW_RAY_EPSILON) {
        int lo_x = ceil2int(posx - f.radius), lo_y = ceil2int(posy - f.radius);
        if (lo_x < 0) lo_x = 0;
        if (lo_y < 0) lo_y = 0;
        int hi_x = floor2int(posx + f.radius), hi_y = floor2int(posy + f.radius);
        if (hi_x > size_x - 1) hi_x = size_x - 1;
        if (hi_y > size_y - 1) hi_y = size_y - 1;
        int n = ceil2int((f.radius - 2.f * MIW_RAY_EPSILON) * 2.f);
        if (n > 8) n = 8;
        float base_x = (float) lo_x - posx, base_y = (float) lo_y - posy;
        float wx[8], wy[8];
        for (int i = 0; i < n; ++i) {
            wx[i] = filter_eval_discretized(f, base_x + (float) i);
            wy[i] = filter_eval_discretized(f, base_y + (float) i);
        }
        for (int yr = 0; yr < n; ++yr) {
            int y = lo_y + yr;
            bool enabled = y <= hi_y;
            // block texel -> film texel (imageblock.cpp:49-77)
            int fy = y + by - f.border;
            for (int xr = 0; xr < n; ++xr) {
                int x = lo_x + xr;
                float weight = wy[yr] * wx[xr];
                enabled = enabled && x <= hi_x;
                int fx = x + bx - f.border;
                if (enabled && fx >= 0 && fx < f.crop_w && fy >= 0 && fy < f.crop_h) {
                    int texel = fy * f.crop_w + fx;
                    for (int k = 0; k < MIW_FILM_CHANNELS; ++k)
                        add(texel, k, value[k] * weight);
                }
            }
        }
    } else {                                             // box filter, :163-170
        int lo_x = ceil2int(posx - .5f), lo_y = ceil2int(posy - .5f);
        if (lo_x >= 0 && lo_y >= 0 && lo_x < size_x && lo_y < size_y) {
            int fx = lo_x + bx - f.border, fy = lo_y + by - f.border;
            if (fx >= 0 && fx < f.crop_w && fy >= 0 && fy < f.crop_h) {
                int texel = fy * f.crop_w + fx;
                for (int k = 0; k < MIW_FILM_CHANNELS; ++k)
                    add(texel, k, value[k]);
            }
        }
    }
}

} // namespace miw
