// The binned-SAH builder of bvh_build.h restated level by level — the form the device runs (sah_device.h): mi_bvh_build quality 0
// builds THE SAME TREE as quality 1 (same nodes in the same breadth-first numbering, same leaves), on the GPU, in a few
// milliseconds instead of 0.6 s of host recursion at 0.9 M triangles. Counterpart in the reference: its GPU mode builds and compacts the acceleration
// structure on the device (include/mitsuba/render/optix/shapes.h:72-167); the CPU variants build the kd-tree
// (src/librender/scene_native.inl:3-10).
//
// Why the same tree and not a similar one: a node's split depends on the SET of its triangles only (box unions, 16-bin counts per
// axis, the surface-area sweep — all order-independent, all in the host builder's float32 expressions), so a breadth-first sweep
// that hands every node of a level to one wavefront reproduces the recursion's decisions exactly; the render kernels then run at
// the SAH tree's speed whichever side built it (measured alternatives over the Morton order: the radix tree of lbvh_device.h, 7 % /
// 16 % behind on the interior / the material balls; a PLOC build — surface-area clustering, Meister & Bittner 2018 — +1 % on the
// balls, -9 % on the interior and two levels too deep for the LDS stack: removed), and the CPU tier can hold the two builders against
// each other node for node (tests/test_sah_levels.py).
//
// A level = the CANDIDATES (first, count, parent, side) in parent order, left before right. Step A, one wavefront per candidate:
// padded box + centroid box of its range, 3 x 16 bins (box + count), the sweep, leaf or split — sah_decide() below is the host
// builder's own expressions. A prefix sum over the "split" flags numbers the level's inner nodes (breadth-first order = the host
// builder's output order). Step B, one wavefront per candidate: write the box and the child reference into the parent's BvhNode,
// partition the range (stable) into the next level's index array, append the two child candidates.
// What the host recursion does differently and the level sweep does not reproduce — the median split it falls back to when no
// axis separates the centroids or beyond depth 40 — raises `need_host`: mi_bvh_build then runs the host builder (coincident
// triangles; never on a scene worth building on the device).
#pragma once
#include <vector>
#include <cstring>
#include "bvh_build.h"

namespace miw {

#define MIW_SAH_BINS 16
struct SahCand { uint32_t first, count; int32_t parent; uint32_t side; };          // side 0 / 1: child0 / child1 of `parent` (-1: the root)
struct SahBox { float lo[3], hi[3]; };
struct SahDecision { SahBox box; uint32_t split, axis, bin, n_left; float clo, scale; };   // split 0: leaf; clo / scale: the bin mapping of `axis`
struct SahBins { SahBox box[3][MIW_SAH_BINS]; uint32_t cnt[3][MIW_SAH_BINS]; };

MIW_HD float sah_half_area(const SahBox &b) {                     // BuildBox::half_area
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}
MIW_HD void sah_box_reset(SahBox &b) { for (int a = 0; a < 3; ++a) { b.lo[a] = MIW_INFINITY; b.hi[a] = -MIW_INFINITY; } }
MIW_HD void sah_box_expand(SahBox &b, const SahBox &o) { for (int a = 0; a < 3; ++a) { b.lo[a] = min_(b.lo[a], o.lo[a]); b.hi[a] = max_(b.hi[a], o.hi[a]); } }
// the bin of a centroid coordinate (bvh_build.h: `std::min(NB - 1, std::max(0, (int) ((c - lo) * scale)))`)
MIW_HD int sah_bin(float c, float lo, float scale) {
    int b = (int) ((c - lo) * scale);
    b = b < 0 ? 0 : b;
    return b > MIW_SAH_BINS - 1 ? MIW_SAH_BINS - 1 : b;
}
// One axis of the sweep (bvh_build.h: the two passes over the bins): the cheapest split of this axis, first bin on ties.
// box_at(b) / cnt_at(b) hand out bin b and right_area / right_cnt hold the suffix pass (the host passes arrays; the device decodes
// its LDS bins in place and keeps the suffixes in LDS as well: a first version with thread-local arrays spent ~1 ms per LEVEL in
// three threads' scratch-memory accesses).
template <typename BoxAt, typename CntAt>
MIW_HD void sah_sweep_axis(BoxAt box_at, CntAt cnt_at, float *right_area, uint32_t *right_cnt, float &best_cost, int &best_bin) {
    SahBox acc; sah_box_reset(acc); uint32_t c = 0;
    for (int b = MIW_SAH_BINS - 1; b > 0; --b) { sah_box_expand(acc, box_at(b)); c += cnt_at(b); right_area[b] = sah_half_area(acc); right_cnt[b] = c; }
    sah_box_reset(acc); c = 0;
    best_cost = MIW_INFINITY; best_bin = -1;
    for (int b = 0; b < MIW_SAH_BINS - 1; ++b) {
        sah_box_expand(acc, box_at(b)); c += cnt_at(b);
        if (c == 0 || right_cnt[b + 1] == 0) continue;
        const float cost = sah_half_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
        if (cost < best_cost) { best_cost = cost; best_bin = b; }
    }
}
// Leaf or split (bvh_build.h: Builder::build between the binning and the partition), from the three axes' sweeps: cost[a] / bin[a] =
// sah_sweep_axis of axis a, MIW_INFINITY / -1 for an axis that was not swept (centroid extent 0, depth >= 40, count 1). The axes
// compete in order with a strict `<`, as the host builder's single running minimum does. Returns 0 leaf, 1 split (axis, bin, the
// bin mapping; n_left is the caller's: the counts of the bins up to `bin`), 2 "the host builder takes its median-split branch
// here" (need_host).
MIW_HD int sah_decide(const SahBox &box, const SahBox &cbox, uint32_t count, uint32_t max_leaf, const float cost[3], const int bin[3], SahDecision &d) {
    d.box = box; d.split = 0; d.axis = 0; d.bin = 0; d.n_left = 0; d.clo = 0.f; d.scale = 0.f;
    if (count == 1) return 0;
    float best_cost = MIW_INFINITY; int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; ++a)
        if (bin[a] >= 0 && cost[a] < best_cost) { best_cost = cost[a]; best_axis = a; best_bin = bin[a]; }
    const float leaf_cost = sah_half_area(box) * count;
    if (count <= max_leaf && (best_axis < 0 || leaf_cost <= best_cost + 1.2f * sah_half_area(box))) return 0;
    if (best_axis < 0) return 2;
    d.split = 1; d.axis = (uint32_t) best_axis; d.bin = (uint32_t) best_bin;
    d.clo = cbox.lo[best_axis]; d.scale = MIW_SAH_BINS / (cbox.hi[best_axis] - cbox.lo[best_axis]);
    return 1;
}
// whether axis a of a candidate is swept at all (bvh_build.h: `depth < 40`, `if (!(ext > 0.f)) continue`)
MIW_HD bool sah_axis_swept(const SahBox &cbox, int a, uint32_t count, uint32_t depth) { return count > 1 && depth < 40 && (cbox.hi[a] - cbox.lo[a]) > 0.f; }
// a triangle's padded box and box centre (bvh_build.h: Builder's constructor)
MIW_HD void sah_prim(const Tri &t, float pad, SahBox &box, float cen[3]) {
    for (int a = 0; a < 3; ++a) {
        const float lo = min_(t.p0[a], min_(t.p1[a], t.p2[a])), hi = max_(t.p0[a], max_(t.p1[a], t.p2[a]));
        cen[a] = 0.5f * (lo + hi);
        box.lo[a] = lo - pad; box.hi[a] = hi + pad;
    }
}
// what a decided candidate writes into its parent's record (and, when it splits, into its own)
MIW_HD void sah_link(BvhNode *nodes, const SahCand &c, const SahDecision &d, int32_t me /* inner index, when it splits */) {
    if (d.split) { nodes[me].parent = c.parent; nodes[me].pad = 0; }
    if (c.parent < 0) return;
    BvhNode &p = nodes[c.parent];
    const int32_t ref = d.split ? me : bvh_leaf_code(c.first, c.count);
    if (c.side == 0) { for (int a = 0; a < 3; ++a) { p.lo0[a] = d.box.lo[a]; p.hi0[a] = d.box.hi[a]; } p.child0 = ref; }
    else             { for (int a = 0; a < 3; ++a) { p.lo1[a] = d.box.lo[a]; p.hi1[a] = d.box.hi[a]; } p.child1 = ref; }
}

struct SahLevelsResult : BvhBuildResult { std::vector<uint32_t> height; std::vector<uint32_t> level_start; bool need_host = false; };

// The host run of the level sweep: the steps above in plain loops (what the kernels of sah_device.h do one wavefront per
// candidate). Same result as bvh_build_sah() — tests/test_sah_levels.py compares them node for node.
inline SahLevelsResult sah_build_levels_host(const std::vector<Tri> &tris, float pad = -1.f, uint32_t max_leaf = 4) {
    SahLevelsResult out;
    const uint32_t n = (uint32_t) tris.size();
    if (max_leaf > 16) max_leaf = 16;
    if (pad < 0.f) pad = 2.f * scene_pad_unit(tris);
    if (n < 2) { out.need_host = true; return out; }
    std::vector<SahBox> pbox(n); std::vector<float> cen(3 * (size_t) n);
    for (uint32_t i = 0; i < n; ++i) sah_prim(tris[i], pad, pbox[i], &cen[3 * (size_t) i]);
    std::vector<uint32_t> idx(n), idx_next(n);
    for (uint32_t i = 0; i < n; ++i) idx[i] = i;
    std::vector<SahCand> cand{ SahCand{ 0u, n, -1, 0u } }, cand_next;
    out.nodes.assign(n, BvhNode{});                               // at most n - 1 inner nodes
    uint32_t base = 0, level = 0;
    out.level_start.push_back(0);
    while (!cand.empty()) {
        std::vector<SahDecision> dec(cand.size()); std::vector<uint32_t> rank(cand.size());
        uint32_t inner = 0;
        for (size_t j = 0; j < cand.size(); ++j) {                // step A
            const SahCand &c = cand[j];
            SahBox box, cbox; sah_box_reset(box); sah_box_reset(cbox);
            for (uint32_t i = c.first; i < c.first + c.count; ++i) {
                sah_box_expand(box, pbox[idx[i]]);
                for (int a = 0; a < 3; ++a) { cbox.lo[a] = min_(cbox.lo[a], cen[3 * (size_t) idx[i] + a]); cbox.hi[a] = max_(cbox.hi[a], cen[3 * (size_t) idx[i] + a]); }
            }
            SahBins bins;
            for (int a = 0; a < 3; ++a) for (int b = 0; b < MIW_SAH_BINS; ++b) { sah_box_reset(bins.box[a][b]); bins.cnt[a][b] = 0; }
            float cost[3] = { MIW_INFINITY, MIW_INFINITY, MIW_INFINITY }; int bin[3] = { -1, -1, -1 };
            for (int a = 0; a < 3; ++a) {
                if (!sah_axis_swept(cbox, a, c.count, level)) continue;
                const float scale = MIW_SAH_BINS / (cbox.hi[a] - cbox.lo[a]);
                for (uint32_t i = c.first; i < c.first + c.count; ++i) {
                    const int b = sah_bin(cen[3 * (size_t) idx[i] + a], cbox.lo[a], scale);
                    sah_box_expand(bins.box[a][b], pbox[idx[i]]); bins.cnt[a][b]++;
                }
                float right_area[MIW_SAH_BINS]; uint32_t right_cnt[MIW_SAH_BINS];
                sah_sweep_axis([&](int b) -> const SahBox & { return bins.box[a][b]; }, [&](int b) { return bins.cnt[a][b]; }, right_area, right_cnt, cost[a], bin[a]);
            }
            const int r = sah_decide(box, cbox, c.count, max_leaf, cost, bin, dec[j]);
            if (r == 2 || (level == 0 && r == 0)) { out.need_host = true; return out; }   // (a single-leaf scene: the host builder wraps it)
            if (r == 1) for (uint32_t b = 0; b <= dec[j].bin; ++b) dec[j].n_left += bins.cnt[dec[j].axis][b];
            rank[j] = inner; inner += dec[j].split;
        }
        cand_next.clear();
        for (size_t j = 0; j < cand.size(); ++j) {                // step B
            const SahCand &c = cand[j]; const SahDecision &d = dec[j];
            const int32_t me = (int32_t) (base + rank[j]);
            sah_link(out.nodes.data(), c, d, me);
            if (!d.split) { for (uint32_t i = c.first; i < c.first + c.count; ++i) idx_next[i] = idx[i]; continue; }
            uint32_t l = c.first, r = c.first + d.n_left;
            for (uint32_t i = c.first; i < c.first + c.count; ++i) {
                const bool left = sah_bin(cen[3 * (size_t) idx[i] + d.axis], d.clo, d.scale) <= (int) d.bin;
                if (left) idx_next[l++] = idx[i]; else idx_next[r++] = idx[i];
            }
            cand_next.push_back(SahCand{ c.first, d.n_left, me, 0u });
            cand_next.push_back(SahCand{ c.first + d.n_left, c.count - d.n_left, me, 1u });
        }
        out.depth = level;
        base += inner; out.level_start.push_back(base);
        cand.swap(cand_next); idx.swap(idx_next); ++level;
    }
    out.nodes.resize(base);
    // BVH2 heights for the 4-wide collapse (bvh4_build.h), bottom-up level by level (a level's nodes are contiguous)
    out.height.assign(base, 0u);
    for (size_t L = out.level_start.size() - 1; L-- > 0;)
        for (uint32_t i = out.level_start[L]; i < out.level_start[L + 1]; ++i) {
            const BvhNode &nd = out.nodes[i];
            const uint32_t h0 = nd.child0 >= 0 ? out.height[nd.child0] : 0u, h1 = nd.child1 >= 0 ? out.height[nd.child1] : 0u;
            out.height[i] = 1u + (h0 > h1 ? h0 : h1);
        }
    out.tris.resize(n); out.order = idx;
    for (uint32_t i = 0; i < n; ++i) out.tris[i] = tris[idx[i]];
    return out;
}

} // namespace miw
