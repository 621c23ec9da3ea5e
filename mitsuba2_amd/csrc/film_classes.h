// Phase classes of a reconstruction filter footprint (miw/film.h): the host-side enumeration.
//
// For one axis, ImageBlock::put (src/librender/imageblock.cpp:114-146) derives from a sample's block-local position
//   lo = ceil(pos - r),  hi = floor(pos + r),  n = ceil((r - 2 eps) * 2) texels,  base = lo - pos,
//   weight of texel lo + i  =  lut[min(int(|(base + i) * scale|), 31)]            (rfilter.h:62-65)
// and with pos = t + phi (t = the texel of the sample's pixel, an integer; phi = the sample's phase, a multiple of 2^-23 in
// [-.5, .5]) every one of these is a function of phi alone — provided nothing is clipped and nothing rounds on the way:
//   * the radius is a multiple of 1/2 (box .5, tent 1, gaussian / mitchell / catmullrom 2): pos - r and lo - pos are exact;
//   * border >= reach (the footprint of a pixel of the block never leaves the bordered block: lo >= 0, hi <= size - 1, so the
//     clamps of :118-127 never bind) — the reference's own border, ceil(r - .5), satisfies it for these radii. The box
//     filter (one texel, weight 1, dropped when it falls outside: :163-170) is the exception that needs no border: a texel
//     outside the block is a texel no lane of the replay owns;
//   * hi - lo + 1 >= n for every phase, so that the loop bounds of :148-161 never cut the footprint short (checked below).
// film_classes_build walks ALL 2^23 + 1 phases through exactly those float32 expressions and records where the answer
// changes: thr[c] = first phase of class c, w[c][a] = weight of texel t - reach + a (0 outside the footprint). A filter
// that breaks a guard, or needs more than 255 classes or 8 offsets, gets ok = false: the render then logs positions
// (24 bytes per sample) and the texel-patch replay (k_film_blocks) derives the weights per sample, as before.
// Host code, shared by libmiwave.so's launcher and the CPU emulator of the device stages (oracle/wavefront_emu.cpp).
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#include "miw/film.h"

namespace miw {

struct FilmClasses {
    bool ok = false;
    uint32_t count = 0;
    int32_t reach = 0, n = 0;
    std::vector<float> thr, w;          // [MIW_FC_TABLE] (thresholds, +inf padding, first class of every phase bin), [256 * 8]
    FilmClassView view() const { FilmClassView v; v.thr = thr.data(); v.w = w.data(); v.count = count; v.reach = reach; return v; }
};

inline FilmClasses film_classes_build(const FilmRec &f) {
    FilmClasses out;
    const float r = f.radius;
    const bool wide = r > 0.5f + MIW_RAY_EPSILON;
    if (!(r > 0.f) || r > 2.f || (wide && r * 2.f != std::floor(r * 2.f))) return out;   // (the box filter's own radius is .5 + eps: its branch never reads it)
    const int n = wide ? ceil2int((r - 2.f * MIW_RAY_EPSILON) * 2.f) : 1;
    if (n < 1 || n > 4) return out;
    const int reach = wide ? -(int) std::ceil(-.5 - (double) r) : 1;        // -min(lo - t) over the phases
    if (wide && f.border < reach) return out;
    if (reach + 1 + n - 1 >= MIW_FC_STRIDE) return out;                       // offsets a = (lo - t) + reach + i must stay below 8
    out.thr.assign(MIW_FC_TABLE, std::numeric_limits<float>::infinity());
    std::vector<int> bin_first(MIW_FC_BINS, -1), bin_last(MIW_FC_BINS, -1);          // first / last class seen in each phase bin
    out.w.assign((size_t) MIW_FC_CLASSES * MIW_FC_STRIDE, 0.f);
    uint32_t count = 0;
    int prev_lo = 0; int prev_ix[4] = { -1, -1, -1, -1 };
    const double q = 1.0 / 8388608.0;                                         // 2^-23
    for (int64_t k = 0; k <= 8388608; ++k) {
        const float phi = (float) ((double) k * q - 0.5);                     // exact
        int lo_rel, ix[4] = { 0, 0, 0, 0 };
        if (wide) {
            // pos = t + phi; pos - r and lo - pos are exact in the reference (multiples of ulp(pos), smaller magnitude), so the
            // integer part t drops out: evaluate them in double and narrow — exact for every phase a real pos can have
            lo_rel = (int) std::ceil((double) phi - (double) r);
            const int hi_rel = (int) std::floor((double) phi + (double) r);
            if (hi_rel - lo_rel + 1 < n) return out;
            const float base = (float) ((double) lo_rel - (double) phi);
            for (int i = 0; i < n; ++i) {
                int index = (int) abs_((base + (float) i) * f.scale_factor);  // filter_eval_discretized's own expression
                if (index > MIW_FILTER_RESOLUTION) index = MIW_FILTER_RESOLUTION;
                ix[i] = index;
            }
        } else lo_rel = (int) std::ceil((double) phi - 0.5);                  // :163: lo = ceil(pos - .5), weight 1
        const bool same = count > 0 && lo_rel == prev_lo && !std::memcmp(ix, prev_ix, sizeof ix);
        {   // the bin film_class_of() computes for this phase, and the class it belongs to
            int bin = (int) ((phi + .5f) * (float) MIW_FC_BINS);
            bin = bin < 0 ? 0 : (bin > MIW_FC_BINS - 1 ? MIW_FC_BINS - 1 : bin);
            const int cls = (int) (same ? count - 1 : count);
            if (bin_first[bin] < 0) bin_first[bin] = cls;
            bin_last[bin] = cls;
        }
        if (same) continue;
        if (count >= MIW_FC_REJECTED) return out;
        prev_lo = lo_rel; std::memcpy(prev_ix, ix, sizeof ix);
        out.thr[count] = count == 0 ? -std::numeric_limits<float>::infinity() : phi;
        for (int i = 0; i < n; ++i) {
            const int a = lo_rel + reach + i;
            if (a < 0 || a >= MIW_FC_STRIDE) return out;
            out.w[(size_t) count * MIW_FC_STRIDE + a] = wide ? f.lut[ix[i]] : 1.f;
        }
        ++count;
    }
    // the search table: a bin's first class as a byte; the search walks at most MIW_FC_PER_BIN boundaries from there
    for (int b = 0; b < MIW_FC_BINS; ++b) {
        if (bin_first[b] < 0 || bin_last[b] - bin_first[b] > MIW_FC_PER_BIN) return out;
        uint32_t word; std::memcpy(&word, &out.thr[MIW_FC_CLASSES + MIW_FC_PER_BIN + (b >> 2)], 4);
        if ((b & 3) == 0) word = 0;
        word |= (uint32_t) bin_first[b] << (8 * (b & 3));
        std::memcpy(&out.thr[MIW_FC_CLASSES + MIW_FC_PER_BIN + (b >> 2)], &word, 4);
    }
    out.count = count; out.reach = reach; out.n = n; out.ok = true;
    return out;
}

} // namespace miw
