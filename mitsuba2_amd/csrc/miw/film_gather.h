// Ordered film assembly from the per-lane sample log — ImageBlock::put +
// Film::put reproduced in the reference's float32 accumulation ORDER.
//
// scalar_rgb accumulates a texel of a bordered ImageBlock in float32 in this
// order: pixels of the 32x32 spiral block in Morton order, each pixel's samples
// back to back (src/librender/integrator.cpp:196-209 -> imageblock.cpp:148-161);
// the film then adds the (up to 4) block partials that cover a texel
// (imageblock.cpp:49-77). Atomics cannot reproduce that order, so the render
// passes only log what put() received (24 B/sample) and this gather — one work
// item per film texel, no atomics, no races — replays the additions in exactly
// that order: the result is bit-identical to the CPU restatement's float32 film
// (block partials merged in ascending block id, the restatement's deterministic
// stand-in for the reference's thread-timing-dependent merge order,
// src/samplers/independent.cpp:36-40).
//
// Cost: each logged sample is read by the <= (2*reach+1)^2 texels around it;
// the reads hit L1/L2 (neighbouring texels of a wavefront share source lanes).
#pragma once
#include "base.h"
#include "rng.h"
#include "film.h"

namespace miw {

struct GatherArgs {
    const F2 *log_pos; const F4 *log_val;   // [sample][lane]
    const U4 *st;                           // st[lane].w = samples finished by that lane
    uint32_t n_lanes;
    const uint32_t *block_ids;              // row-major block -> spiral id
    const int32_t *block_tile;              // row-major block -> tile index in this shard, or -1
    uint32_t blocks_x, blocks_y;
    uint32_t bs2_log2;                      // log2(block_size^2): lanes per tile
};

MIW_HD uint32_t morton_part1by1(uint32_t x) {
    x &= 0x0000ffffu;
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}
MIW_HD uint32_t morton_encode2(uint32_t x, uint32_t y) { return morton_part1by1(x) | (morton_part1by1(y) << 1); }

// Partial sum of one bordered block for the texel at block-local (tx, ty):
// imageblock.cpp:114-161 restricted to one texel, float32, reference order.
MIW_HD void gather_block_texel(const FilmRec &f, const GatherArgs &g, uint32_t tile,
                               int off_x, int off_y, int bw, int bh, int tx, int ty, float *partial) {
    const int size_x = bw + 2 * f.border, size_y = bh + 2 * f.border;
    const float kx = (float) (off_x - f.border) + .5f, ky = (float) (off_y - f.border) + .5f;
    const bool wide = f.radius > 0.5f + MIW_RAY_EPSILON;
    int n = ceil2int((f.radius - 2.f * MIW_RAY_EPSILON) * 2.f);
    if (n > 8) n = 8;
    const int reach = floor2int(f.radius + .5f);
    // source pixels (block-local) that can touch this texel
    int x0 = tx - f.border - reach, x1 = tx - f.border + reach,
        y0 = ty - f.border - reach, y1 = ty - f.border + reach;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > bw - 1) x1 = bw - 1;
    if (y1 > bh - 1) y1 = bh - 1;
    for (int k = 0; k < MIW_FILM_CHANNELS; ++k) partial[k] = 0.f;
    if (x0 > x1 || y0 > y1) return;

    // visit them in increasing Morton index (render_block's pixel order)
    int64_t last = -1;
    for (;;) {
        int64_t best = -1; int bx = 0, by = 0;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                int64_t code = (int64_t) morton_encode2((uint32_t) x, (uint32_t) y);
                if (code > last && (best < 0 || code < best)) { best = code; bx = x; by = y; }
            }
        if (best < 0) break;
        last = best;
        (void) bx; (void) by;
        const uint32_t lane = (tile << g.bs2_log2) + (uint32_t) best;
        const uint32_t count = g.st[lane].w;
        for (uint32_t j = 0; j < count; ++j) {           // this pixel's samples, back to back
            const size_t e = (size_t) j * g.n_lanes + lane;
            const F2 p = g.log_pos[e];
            if (!(p.x == p.x)) continue;                 // rejected sample (imageblock.cpp:98-108)
            const float posx = p.x - kx, posy = p.y - ky;   // :114
            float weight;
            if (wide) {
                int lo_x = ceil2int(posx - f.radius), lo_y = ceil2int(posy - f.radius);
                if (lo_x < 0) lo_x = 0;
                if (lo_y < 0) lo_y = 0;
                int hi_x = floor2int(posx + f.radius), hi_y = floor2int(posy + f.radius);
                if (hi_x > size_x - 1) hi_x = size_x - 1;
                if (hi_y > size_y - 1) hi_y = size_y - 1;
                const int xr = tx - lo_x, yr = ty - lo_y;
                if (xr < 0 || xr >= n || yr < 0 || yr >= n || tx > hi_x || ty > hi_y) continue;
                const float wx = filter_eval_discretized(f, ((float) lo_x - posx) + (float) xr),
                            wy = filter_eval_discretized(f, ((float) lo_y - posy) + (float) yr);
                weight = wy * wx;                        // :155
            } else {
                if (ceil2int(posx - .5f) != tx || ceil2int(posy - .5f) != ty) continue;   // :163-170
                weight = 1.f;
            }
            const F4 v = g.log_val[e];
            if (wide) {
                partial[0] += v.x * weight; partial[1] += v.y * weight; partial[2] += v.z * weight;
                partial[3] += v.w * weight; partial[4] += 1.f * weight;
            } else {
                partial[0] += v.x; partial[1] += v.y; partial[2] += v.z; partial[3] += v.w; partial[4] += 1.f;
            }
        }
    }
}

// Film texel (fx, fy) in crop-relative coordinates -> out[5] (X, Y, Z, A, W).
MIW_HD void film_gather_texel(const FilmRec &f, const GatherArgs &g, int fx, int fy, float *out) {
    const int bs = f.block_size;
    for (int k = 0; k < MIW_FILM_CHANNELS; ++k) out[k] = 0.f;
    // blocks whose bordered area contains the texel: at most 2 x 2
    int bx_lo = (fx - f.border) / bs, bx_hi = (fx + f.border) / bs,
        by_lo = (fy - f.border) / bs, by_hi = (fy + f.border) / bs;
    if (fx - f.border < 0) bx_lo = 0;
    if (fy - f.border < 0) by_lo = 0;
    if (bx_hi > (int) g.blocks_x - 1) bx_hi = (int) g.blocks_x - 1;
    if (by_hi > (int) g.blocks_y - 1) by_hi = (int) g.blocks_y - 1;
    uint32_t cand[4]; uint32_t cand_id[4]; int nc = 0;
    for (int by = by_lo; by <= by_hi && nc < 4; ++by)
        for (int bx = bx_lo; bx <= bx_hi && nc < 4; ++bx) {
            uint32_t b = (uint32_t) by * g.blocks_x + (uint32_t) bx;
            if (g.block_tile[b] < 0) continue;           // rendered by another rank
            cand[nc] = b; cand_id[nc] = g.block_ids[b]; ++nc;
        }
    // merge in ascending spiral id (film->put order of the restatement)
    for (int i = 1; i < nc; ++i)
        for (int j = i; j > 0 && cand_id[j - 1] > cand_id[j]; --j) {
            uint32_t t = cand_id[j]; cand_id[j] = cand_id[j - 1]; cand_id[j - 1] = t;
            t = cand[j]; cand[j] = cand[j - 1]; cand[j - 1] = t;
        }
    for (int i = 0; i < nc; ++i) {
        const uint32_t b = cand[i];
        const int bx = (int) (b % g.blocks_x), by = (int) (b / g.blocks_x);
        const int px0 = bx * bs, py0 = by * bs;
        const int bw = f.crop_w - px0 < bs ? f.crop_w - px0 : bs,
                  bh = f.crop_h - py0 < bs ? f.crop_h - py0 : bs;
        const int tx = fx - px0 + f.border, ty = fy - py0 + f.border;
        if (tx < 0 || ty < 0 || tx >= bw + 2 * f.border || ty >= bh + 2 * f.border) continue;
        float partial[MIW_FILM_CHANNELS];
        gather_block_texel(f, g, (uint32_t) g.block_tile[b], px0 + f.crop_x, py0 + f.crop_y, bw, bh, tx, ty, partial);
        for (int k = 0; k < MIW_FILM_CHANNELS; ++k) out[k] += partial[k];   // imageblock.cpp:49-77
    }
}

} // namespace miw
