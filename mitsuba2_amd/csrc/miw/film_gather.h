// Ordered film assembly from the per-lane sample log — ImageBlock::put +
// Film::put reproduced in the reference's float32 accumulation ORDER.
//
// scalar_rgb accumulates a texel of a bordered ImageBlock in float32 in this
// order: pixels of the 32x32 spiral block in Morton order, each pixel's samples
// back to back (src/librender/integrator.cpp:196-209 -> imageblock.cpp:148-161);
// the film then adds the (up to 4) block partials that cover a texel
// (imageblock.cpp:49-77). Atomics into a shared film cannot reproduce that
// order, so the render passes only LOG what put() received (24 B/sample,
// [lane][sample] = one contiguous run per pixel) and the film is assembled
// afterwards in two steps:
//   1. block replay: one bordered ImageBlock per spiral block, its pixels'
//      sample runs replayed in Morton order (on the device: one wavefront per
//      block, the block lives in LDS, each lane owns one (footprint texel,
//      channel) slot and issues in-order LDS float adds — see k_film_blocks);
//   2. merge: every film texel sums the block tiles that cover it in ascending
//      block id (the CPU restatement's deterministic stand-in for the reference's
//      thread-timing-dependent Film::put order, src/samplers/independent.cpp:36-40).
// The result is bit-identical to the CPU restatement's float32 film.
//
// This header holds the portable form of both steps (used as-is by the CPU
// checker; the device kernels re-express step 1 lane-parallel with the same
// arithmetic).
#pragma once
#include "base.h"
#include "rng.h"
#include "film.h"

namespace miw {

struct BlockReplayArgs {
    const F2 *log_pos; const F4 *log_val;   // [lane][sample], `spp` entries per lane (24-byte format), or
    const U4 *log_rec;                      // the 16-byte records (path.h: LogSink16) with their class tables in `cls`
    uint32_t log_il = 0;                    // path.h: log_index (0: [lane][sample]; else the tile-interleaved layout k_film_lanes reads)
    FilmClassView cls;
    const U4 *st;                           // st[lane].w = samples finished by that lane
    uint32_t spp;
    const uint32_t *block_ids;              // row-major block -> spiral id
    const int32_t *block_tile;              // row-major block -> tile index in this shard, or -1
    const uint32_t *tile_list;              // tile -> row-major block (nullptr: identity)
    uint32_t blocks_x, blocks_y;
    uint32_t bs2_log2;                      // log2(block_size^2): lanes per tile
    uint32_t tile_stride;                   // floats per block tile in `tiles`: (bs + 2*border)^2 * 5
};

// geometry of the bordered block of row-major block index b
struct BlockGeom { int px0, py0, bw, bh, size_x, size_y; };
MIW_HD BlockGeom block_geom(const FilmRec &f, uint32_t blocks_x, uint32_t b) {
    BlockGeom g;
    g.px0 = (int) (b % blocks_x) * f.block_size; g.py0 = (int) (b / blocks_x) * f.block_size;
    g.bw = f.crop_w - g.px0 < f.block_size ? f.crop_w - g.px0 : f.block_size;
    g.bh = f.crop_h - g.py0 < f.block_size ? f.crop_h - g.py0 : f.block_size;
    g.size_x = g.bw + 2 * f.border; g.size_y = g.bh + 2 * f.border;
    return g;
}

// Step 1, portable: replay one block into acc[size_y * size_x * 5] (zeroed by the caller).
MIW_HD void film_block_replay(const FilmRec &f, const BlockReplayArgs &a, uint32_t tile, float *acc) {
    const uint32_t b = a.tile_list ? a.tile_list[tile] : tile;
    const BlockGeom g = block_geom(f, a.blocks_x, b);
    const uint32_t bs2 = 1u << a.bs2_log2;
    for (uint32_t q = 0; q < bs2; ++q) {                  // render_block's pixel order, integrator.cpp:196-203
        uint32_t x, y;
        morton_decode2(q, x, y);
        if ((int) x >= g.bw || (int) y >= g.bh) continue;
        const uint32_t lane = (tile << a.bs2_log2) + q;
        const uint32_t count = a.st[lane].w;
        const F2 *lp = a.log_pos + (size_t) lane * a.spp; const F4 *lv = a.log_val + (size_t) lane * a.spp;
        for (uint32_t j = 0; j < count; ++j) {            // this pixel's samples, back to back
            const F2 p = lp[j];
            if (!(p.x == p.x)) continue;                  // rejected sample (imageblock.cpp:98-108)
            const F4 v = lv[j];
            const float value[5] = { v.x, v.y, v.z, v.w, 1.f };
            block_splat(f, g.px0 + f.crop_x, g.py0 + f.crop_y, g.bw, g.bh, v2(p.x, p.y), value,
                        [acc](int texel, int k, float term) { acc[texel * MIW_FILM_CHANNELS + k] += term; });
        }
    }
}

// Step 1 for the 16-byte records: a sample of pixel (x, y) adds value * (w[cy][ay] * w[cx][ax]) to every texel of the
// (2 reach + 1)^2 window around the pixel's texel that lies inside the bordered block — the texels outside its footprint
// get value * 0 = +-0, which leaves a float32 sum as it is (sums start at +0 and can never become -0), the texels inside
// get the product block_splat() forms, in the same pixel / sample order. The device's k_film_groups does exactly this.
MIW_HD void film_block_replay16(const FilmRec &f, const BlockReplayArgs &a, uint32_t tile, float *acc) {
    const uint32_t b = a.tile_list ? a.tile_list[tile] : tile;
    const BlockGeom g = block_geom(f, a.blocks_x, b);
    const uint32_t bs2 = 1u << a.bs2_log2;
    const int reach = a.cls.reach;
    for (uint32_t q = 0; q < bs2; ++q) {
        uint32_t x, y;
        morton_decode2(q, x, y);
        if ((int) x >= g.bw || (int) y >= g.bh) continue;
        const uint32_t lane = (tile << a.bs2_log2) + q;
        const uint32_t count = a.st[lane].w;
        const int ptx = (int) x + f.border, pty = (int) y + f.border;
        for (uint32_t j = 0; j < count; ++j) {
            const U4 r = a.log_rec[log_index(a.log_il, lane, a.spp, j)];
            const uint32_t cx = r.w & 255u, cy = (r.w >> 8) & 255u;
            const float value[5] = { u2f(r.x), u2f(r.y), u2f(r.z), (r.w >> 16) & 1u ? 1.f : 0.f, 1.f };
            for (int ay = 0; ay <= 2 * reach; ++ay)
                for (int ax = 0; ax <= 2 * reach; ++ax) {
                    const int tx = ptx - reach + ax, ty = pty - reach + ay;
                    if (tx < 0 || ty < 0 || tx >= g.size_x || ty >= g.size_y) continue;
                    const float w = a.cls.w[cy * MIW_FC_STRIDE + ay] * a.cls.w[cx * MIW_FC_STRIDE + ax];   // wy * wx, :155
                    float *dst = acc + ((size_t) ty * g.size_x + tx) * MIW_FILM_CHANNELS;
                    for (int k = 0; k < MIW_FILM_CHANNELS; ++k) dst[k] += value[k] * w;
                }
        }
    }
}

// Step 2: film texel (fx, fy), crop-relative -> out[5]; `tiles` = block tiles written by step 1.
// `accumulate`: out[] already holds the texel of earlier passes (their block ids are smaller) and is added onto.
MIW_HD void film_merge_texel(const FilmRec &f, const BlockReplayArgs &a, const float *tiles, int fx, int fy, float *out,
                             bool accumulate = false) {
    const int bs = f.block_size;
    if (!accumulate) for (int k = 0; k < MIW_FILM_CHANNELS; ++k) out[k] = 0.f;
    // blocks whose bordered area contains the texel: 2 x 2 at most for block_size >= 2 * border, more for the tiny blocks
    // a many-threaded render of a small frame uses (integrator.cpp:88-97 halves the block size down to 1)
    int bx_lo = (fx - f.border) / bs, bx_hi = (fx + f.border) / bs,
        by_lo = (fy - f.border) / bs, by_hi = (fy + f.border) / bs;
    if (fx - f.border < 0) bx_lo = 0;
    if (fy - f.border < 0) by_lo = 0;
    if (bx_hi > (int) a.blocks_x - 1) bx_hi = (int) a.blocks_x - 1;
    if (by_hi > (int) a.blocks_y - 1) by_hi = (int) a.blocks_y - 1;
    // ascending spiral id = the restatement's film->put order: repeatedly take the covering block with the smallest
    // id above the last one added (ids are unique; no candidate array, any number of covering blocks)
    uint64_t floor_id = 0;                                 // next id must be >= floor_id
    for (;;) {
        uint64_t best_id = ~0ull; uint32_t best_b = 0;
        for (int by = by_lo; by <= by_hi; ++by)
            for (int bx = bx_lo; bx <= bx_hi; ++bx) {
                const uint32_t b = (uint32_t) by * a.blocks_x + (uint32_t) bx;
                if (a.block_tile[b] < 0) continue;       // rendered by another rank
                const uint64_t id = a.block_ids[b];
                if (id >= floor_id && id < best_id) { best_id = id; best_b = b; }
            }
        if (best_id == ~0ull) break;
        floor_id = best_id + 1;
        const BlockGeom g = block_geom(f, a.blocks_x, best_b);
        const int tx = fx - g.px0 + f.border, ty = fy - g.py0 + f.border;
        if (tx < 0 || ty < 0 || tx >= g.size_x || ty >= g.size_y) continue;
        const float *src = tiles + (size_t) a.block_tile[best_b] * a.tile_stride + ((size_t) ty * g.size_x + tx) * MIW_FILM_CHANNELS;
        for (int k = 0; k < MIW_FILM_CHANNELS; ++k) out[k] += src[k];   // imageblock.cpp:49-77
    }
}

} // namespace miw
