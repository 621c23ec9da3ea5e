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
    uint32_t warn_negative;             // ImageBlock::m_warn_negative = !has_aovs (integrator.cpp:110-113): 0 under the moment integrator
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
// `warn_negative` = ImageBlock::m_warn_negative: SamplingIntegrator::render builds its blocks with
// warn_negative = !has_aovs (integrator.cpp:110-113), so under the moment integrator only the isfinite test applies.
MIW_HD bool sample_is_valid(const float *value, bool warn_negative = true) {
    bool ok = true;
    for (int k = 0; k < MIW_FILM_CHANNELS; ++k)
        ok = ok && (!warn_negative || value[k] >= -1e-5f) && isfinite_(value[k]);
    return ok;
}

// ImageBlock::put into ONE bordered block (offset off_x/off_y, clipped size bw x bh):
// calls add(block_texel, channel, value) with block_texel = y * (bw + 2*border) + x,
// for every texel the sample touches, in the reference's loop order (:148-161).
// The caller has already validated the sample (sample_is_valid).
template <typename Add>
MIW_HD void block_splat(const FilmRec &f, int off_x, int off_y, int bw, int bh, V2 pos_, const float *value, Add add) {
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
            for (int xr = 0; xr < n; ++xr) {
                int x = lo_x + xr;
                float weight = wy[yr] * wx[xr];
                enabled = enabled && x <= hi_x;
                if (enabled) {
                    int texel = y * size_x + x;
                    for (int k = 0; k < MIW_FILM_CHANNELS; ++k)
                        add(texel, k, value[k] * weight);
                }
            }
        }
    } else {                                             // box filter, :163-170
        int lo_x = ceil2int(posx - .5f), lo_y = ceil2int(posy - .5f);
        if (lo_x >= 0 && lo_y >= 0 && lo_x < size_x && lo_y < size_y) {
            int texel = lo_y * size_x + lo_x;
            for (int k = 0; k < MIW_FILM_CHANNELS; ++k)
                add(texel, k, value[k]);
        }
    }
}

// ---- phase classes: what ImageBlock::put derives from a sample's position, in one byte per axis ----------------------
// Everything put() computes from the position of a sample (imageblock.cpp:114-146: lo, the footprint, the discretised
// weight indices) depends, per axis, only on the sample's PHASE phi = pos - (block-local texel coordinate of its pixel),
// a multiple of 2^-23 in [-.5, .5] — under the guards of film_classes.h (radius a multiple of 1/2, border >= reach: nothing
// is clipped, every subtraction on the way is exact). It is a step function of phi with ~70 steps (r = 2; each of the n
// weight indices moves through ~16 LUT bins as the sample crosses its pixel), so the host enumerates it once per filter
// (film_classes.h: all 2^23 + 1 phases through the reference's own float expressions) into
//   thr[256]     ascending phase thresholds: class c = [thr[c], thr[c + 1]); +inf beyond the last class (+ the search's bin table)
//   w[256][8]    w[c][a] = the weight put() gives texel (pixel's texel - reach + a), 0 outside the footprint;
//                rows >= count are 0: a rejected sample is logged with class `count`
// and the render kernels log 16 bytes per sample — X, Y, Z, class_x | class_y << 8 | alpha << 16 — instead of 8 bytes of
// position + 16 of value; the film replay looks the two weights up in its LDS copy of w. Same float32 products and
// sums as block_splat() above, texel for texel (tests/test_film_classes.py; every film parity test runs through it).
#define MIW_FC_CLASSES 256
#define MIW_FC_STRIDE 8
#define MIW_FC_REJECTED 255u            /* upper bound of the class count */
#define MIW_FC_BINS 256                 /* phase bins of the class search */
#define MIW_FC_PER_BIN 4                /* at most this many class boundaries inside one bin (checked by film_classes_build) */
#define MIW_FC_TABLE (MIW_FC_CLASSES + MIW_FC_PER_BIN + MIW_FC_BINS / 4)   /* floats: thresholds, +inf padding, the bins' first classes as bytes */
struct FilmClassView {
    const float *thr;                   // [MIW_FC_TABLE]: thr[0..255] class thresholds, thr[256..259] = +inf, then 256 bytes: first class of each phase bin
    const float *w;                     // [256][8]
    uint32_t count;                     // classes in use (<= 255)
    int32_t reach;                      // texels a footprint extends to the left of its pixel's texel (2 for r = 2, 1 for box)
};
// :114 for one axis, then the phase. pixel = film coordinate of the sample's pixel (crop offset included).
MIW_HD float film_phase(const FilmRec &f, float pos, int pixel, int crop_off) {
    const int local = pixel - crop_off, b0 = local & ~(f.block_size - 1);          // block_size is a power of two (mi_render checks)
    const float p = pos - ((float) (b0 + crop_off - f.border) + .5f);              // block-local position, as block_splat
    return p - (float) (local - b0 + f.border);                                    // exact: both multiples of ulp(p), |result| <= .5
}
// The class of a phase: the first class of the phase's bin (1 / 256 wide; a byte table behind the thresholds), then at most
// MIW_FC_PER_BIN boundaries further — two dependent reads (the second round's four are independent of each other) instead
// of an 8-step binary search: the sample-finish code runs under divergence almost every iteration of the render kernels
// (lanes finish their samples at different times), so every instruction in it is paid nearly once per path segment.
template <typename Thr>
MIW_HD uint32_t film_class_of(Thr thr, float phi) {
    int bin = (int) ((phi + .5f) * (float) MIW_FC_BINS);                           // exact: phi is a multiple of 2^-23 in [-.5, .5]
    bin = bin < 0 ? 0 : (bin > MIW_FC_BINS - 1 ? MIW_FC_BINS - 1 : bin);
    const uint32_t word = f2u(thr[MIW_FC_CLASSES + MIW_FC_PER_BIN + (bin >> 2)]);
    uint32_t c = (word >> (8 * (bin & 3))) & 255u;
    const float t1 = thr[c + 1], t2 = thr[c + 2], t3 = thr[c + 3], t4 = thr[c + 4];
    c += (phi >= t1 ? 1u : 0u) + (phi >= t2 ? 1u : 0u) + (phi >= t3 ? 1u : 0u) + (phi >= t4 ? 1u : 0u);
    return c;
}
// Where lane's j-th 16-byte record lives. il = 0: [lane][j]. il = 1 + log2(lanes per tile), chosen by mi_render when the film will be
// replayed by k_film_lanes (device/film_kernels.h: a wavefront = the same texel block of 64 consecutive tiles, each lane streaming its
// own tile's runs): [tile / 64][pixel of the tile][j][tile % 64] — the 64 records a replay load fetches are 1 KB of consecutive bytes
// instead of 64 cache lines 8 MB apart. For the kernel that writes the log the change is neutral: a pixel's consecutive records are
// 1 KB apart instead of adjacent, but they were never written together (a lane finishes a sample every few hundred microseconds).
MIW_HD size_t log_index(uint32_t il, uint32_t lane, uint32_t spp, uint32_t j) {
    if (!il) return (size_t) lane * spp + j;
    const uint32_t sh = il - 1u;
    const uint32_t hi = ((lane >> (sh + 6u)) << sh) | (lane & ((1u << sh) - 1u));       // (tile / 64, pixel)
    return (((size_t) hi * spp + j) << 6) | (size_t) ((lane >> sh) & 63u);              // ... record j, tile % 64
}
// records the log must hold for n_tiles tiles of 2^sh lanes (the interleaved layout pads to whole groups of 64 tiles)
MIW_HD size_t log_capacity(uint32_t il, uint32_t n_tiles, uint32_t lanes_per_tile, uint32_t spp) {
    return (size_t) (il ? (n_tiles + 63u) / 64u * 64u : n_tiles) * lanes_per_tile * spp;
}
MIW_HD uint32_t film_pack_meta(uint32_t cx, uint32_t cy, bool alpha) { return cx | (cy << 8) | (alpha ? 1u << 16 : 0u); }

// The spiral block a pixel belongs to (spiral.cpp:43-45): bordered-block origin and clipped size
MIW_HD void block_of_pixel(const FilmRec &f, int px, int py, int &bx, int &by, int &bw, int &bh) {
    bx = ((px - f.crop_x) / f.block_size) * f.block_size;
    by = ((py - f.crop_y) / f.block_size) * f.block_size;
    bw = f.crop_w - bx < f.block_size ? f.crop_w - bx : f.block_size;
    bh = f.crop_h - by < f.block_size ? f.crop_h - by : f.block_size;
}

// ImageBlock::put followed by Film::put (imageblock.cpp:49-77, out-of-film border texels
// clipped): add_xy(fx, fy, channel, value) with crop-relative film coordinates.
// `px,py` = the pixel the sample belongs to.
template <typename AddXY>
MIW_HD void film_splat_xy(const FilmRec &f, int px, int py, V2 pos_, const float *value, AddXY add_xy) {
    if (!sample_is_valid(value, f.warn_negative != 0)) return;
    int bx, by, bw, bh;
    block_of_pixel(f, px, py, bx, by, bw, bh);
    const int size_x = bw + 2 * f.border;
    block_splat(f, bx + f.crop_x, by + f.crop_y, bw, bh, pos_, value, [&](int texel, int k, float v) {
        int x = texel % size_x, y = texel / size_x;
        int fx = x + bx - f.border, fy = y + by - f.border;
        if (fx >= 0 && fx < f.crop_w && fy >= 0 && fy < f.crop_h) add_xy(fx, fy, k, v);
    });
}
// same, add(film_texel, channel, value)
template <typename Add>
MIW_HD void film_splat(const FilmRec &f, int px, int py, V2 pos_, const float *value, Add add) {
    const int w = f.crop_w;
    film_splat_xy(f, px, py, pos_, value, [&](int fx, int fy, int k, float v) { add(fy * w + fx, k, v); });
}

} // namespace miw
