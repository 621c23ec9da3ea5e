// Host-side binned-SAH BVH2 builder (mi_bvh_build quality = 1).
//
// Stands where ShapeKDTree::build() stands in the reference
// (src/librender/scene_native.inl:3-10, include/mitsuba/render/kdtree.h): it
// produces the acceleration structure once per scene. Nothing of the kd-tree's
// structure is reproduced — only its observable result matters (bvh.h).
// Output: 64-byte nodes in breadth-first order (so "the first K nodes" are the
// top of the tree, which the trace kernels stage in LDS) and the triangle array
// permuted into leaf order.
#pragma once
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include "miw/scene.h"
#include "miw/bvh.h"

namespace miw {

struct BuildBox {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; } }
    void expand(const float *p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    void expand(const BuildBox &b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BvhBuildResult {
    std::vector<BvhNode> nodes;
    std::vector<Tri> tris;            // leaf order
    std::vector<uint32_t> order;      // order[i] = index into the input array of tris[i]
    uint32_t depth = 0;
};

namespace detail {

struct TmpNode { BuildBox box; int32_t left = -1, right = -1; uint32_t first = 0, count = 0; };

struct Builder {
    const std::vector<Tri> &in;
    std::vector<BuildBox> boxes;
    std::vector<float> cx[3];
    std::vector<uint32_t> idx;
    std::vector<TmpNode> tmp;
    uint32_t max_leaf, depth_seen = 0;

    Builder(const std::vector<Tri> &t, float pad, uint32_t max_leaf_) : in(t), max_leaf(max_leaf_) {
        size_t n = t.size();
        boxes.resize(n); idx.resize(n);
        for (int a = 0; a < 3; ++a) cx[a].resize(n);
        for (size_t i = 0; i < n; ++i) {
            BuildBox b; b.reset();
            b.expand(t[i].p0); b.expand(t[i].p1); b.expand(t[i].p2);
            for (int a = 0; a < 3; ++a) {
                cx[a][i] = 0.5f * (b.lo[a] + b.hi[a]);
                b.lo[a] -= pad; b.hi[a] += pad;
            }
            boxes[i] = b; idx[i] = (uint32_t) i;
        }
    }

    int32_t build(uint32_t first, uint32_t count, uint32_t depth) {
        depth_seen = std::max(depth_seen, depth);
        int32_t me = (int32_t) tmp.size();
        tmp.emplace_back();
        BuildBox box; box.reset();
        BuildBox cbox; cbox.reset();
        for (uint32_t i = first; i < first + count; ++i) {
            box.expand(boxes[idx[i]]);
            float c[3] = { cx[0][idx[i]], cx[1][idx[i]], cx[2][idx[i]] };
            cbox.expand(c);
        }
        tmp[me].box = box;
        auto make_leaf = [&]() { tmp[me].first = first; tmp[me].count = count; return me; };
        if (count == 1) return make_leaf();

        // binned SAH over the three axes
        const int NB = 16;
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        if (depth < 40) {
            for (int a = 0; a < 3; ++a) {
                float ext = cbox.hi[a] - cbox.lo[a];
                if (!(ext > 0.f)) continue;
                BuildBox bb[NB]; uint32_t bc[NB];
                for (int b = 0; b < NB; ++b) { bb[b].reset(); bc[b] = 0; }
                float scale = NB / ext;
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = std::min(NB - 1, std::max(0, (int) ((cx[a][idx[i]] - cbox.lo[a]) * scale)));
                    bb[b].expand(boxes[idx[i]]); bc[b]++;
                }
                float right_area[NB]; uint32_t right_cnt[NB];
                BuildBox acc; acc.reset(); uint32_t c = 0;
                for (int b = NB - 1; b > 0; --b) { acc.expand(bb[b]); c += bc[b]; right_area[b] = acc.half_area(); right_cnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    acc.expand(bb[b]); c += bc[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * c + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
        }
        float leaf_cost = box.half_area() * count;
        // intersection cost 1 per tri vs traversal cost 1.2 per node visit
        if (count <= max_leaf && (best_axis < 0 || leaf_cost <= best_cost + 1.2f * box.half_area()))
            return make_leaf();

        uint32_t mid;
        if (best_axis >= 0) {
            float ext = cbox.hi[best_axis] - cbox.lo[best_axis], scale = NB / ext, lo = cbox.lo[best_axis];
            const std::vector<float> &c = cx[best_axis];
            auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t i) {
                int b = std::min(NB - 1, std::max(0, (int) ((c[i] - lo) * scale)));
                return b <= best_bin;
            });
            mid = (uint32_t) (it - idx.begin());
            if (mid == first || mid == first + count) mid = first + count / 2;
        } else {
            // identical centroids or depth guard: split by index (bounded depth)
            int a = 0;
            float e0 = cbox.hi[0] - cbox.lo[0], e1 = cbox.hi[1] - cbox.lo[1], e2 = cbox.hi[2] - cbox.lo[2];
            if (e1 > e0 && e1 >= e2) a = 1; else if (e2 > e0 && e2 > e1) a = 2;
            mid = first + count / 2;
            const std::vector<float> &c = cx[a];
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                             [&](uint32_t x, uint32_t y) { return c[x] < c[y] || (c[x] == c[y] && x < y); });
        }
        int32_t l = build(first, mid - first, depth + 1);
        int32_t r = build(mid, first + count - mid, depth + 1);
        tmp[me].left = l; tmp[me].right = r;
        return me;
    }
};

} // namespace detail

// The scene's padding unit: 1e-5 x the largest |coordinate|. A triangle hit counts iff its point lies inside the
// triangle's bounds grown by ONE unit (shape.h: accept_pad); the BVH boxes are grown by TWO, so that such a hit is
// never culled whatever the rounding of the slab tests (~1e-7 x the coordinates).
inline float scene_pad_unit(const std::vector<Tri> &tris) {
    float m = 0.f;
    for (const Tri &t : tris)
        for (int k = 0; k < 3; ++k) {
            m = std::max(m, std::fabs(t.p0[k])); m = std::max(m, std::fabs(t.p1[k])); m = std::max(m, std::fabs(t.p2[k]));
        }
    return std::max(1e-5f * m, 1e-30f);
}

// `pad`: absolute box padding (see bvh.h); pass <0 to derive it from the scene extent (2 x scene_pad_unit).
inline BvhBuildResult bvh_build_sah(const std::vector<Tri> &tris, float pad = -1.f, uint32_t max_leaf = 4) {
    BvhBuildResult out;
    if (max_leaf > 16) max_leaf = 16;
    if (pad < 0.f) pad = 2.f * scene_pad_unit(tris);
    const float inf = std::numeric_limits<float>::infinity();
    auto set_empty = [&](float *lo, float *hi) { for (int a = 0; a < 3; ++a) { lo[a] = inf; hi[a] = -inf; } };

    if (tris.empty()) {
        BvhNode n; std::memset(&n, 0, sizeof n);
        set_empty(n.lo0, n.hi0); set_empty(n.lo1, n.hi1);
        n.child0 = n.child1 = -1; n.parent = -1;
        out.nodes.push_back(n);
        return out;
    }

    detail::Builder b(tris, pad, max_leaf);
    b.tmp.reserve(tris.size() * 2);
    int32_t root = b.build(0, (uint32_t) tris.size(), 0);
    out.depth = b.depth_seen;

    // leaf-order triangles
    out.tris.resize(tris.size()); out.order.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) { out.tris[i] = tris[b.idx[i]]; out.order[i] = b.idx[i]; }

    // breadth-first numbering of inner nodes
    struct Item { int32_t tmp; int32_t parent; };
    std::vector<Item> queue;
    std::vector<int32_t> inner_index(b.tmp.size(), -1);
    auto is_leaf = [&](int32_t t) { return b.tmp[t].left < 0; };
    if (is_leaf(root)) {
        // single leaf: wrap it in a root with one absent child
        BvhNode n; std::memset(&n, 0, sizeof n);
        std::memcpy(n.lo0, b.tmp[root].box.lo, 12); std::memcpy(n.hi0, b.tmp[root].box.hi, 12);
        set_empty(n.lo1, n.hi1);
        n.child0 = bvh_leaf_code(b.tmp[root].first, b.tmp[root].count); n.child1 = -1; n.parent = -1;
        out.nodes.push_back(n);
        out.depth = 1;
        return out;
    }
    queue.push_back({ root, -1 });
    for (size_t q = 0; q < queue.size(); ++q) {
        int32_t t = queue[q].tmp;
        inner_index[t] = (int32_t) q;
        if (!is_leaf(b.tmp[t].left))  queue.push_back({ b.tmp[t].left,  (int32_t) q });
        if (!is_leaf(b.tmp[t].right)) queue.push_back({ b.tmp[t].right, (int32_t) q });
    }
    out.nodes.resize(queue.size());
    for (size_t q = 0; q < queue.size(); ++q) {
        const detail::TmpNode &t = b.tmp[queue[q].tmp];
        const detail::TmpNode &l = b.tmp[t.left], &r = b.tmp[t.right];
        BvhNode n; std::memset(&n, 0, sizeof n);
        std::memcpy(n.lo0, l.box.lo, 12); std::memcpy(n.hi0, l.box.hi, 12);
        std::memcpy(n.lo1, r.box.lo, 12); std::memcpy(n.hi1, r.box.hi, 12);
        n.child0 = is_leaf(t.left)  ? bvh_leaf_code(l.first, l.count) : inner_index[t.left];
        n.child1 = is_leaf(t.right) ? bvh_leaf_code(r.first, r.count) : inner_index[t.right];
        n.parent = queue[q].parent;
        out.nodes[q] = n;
    }
    // children were numbered after their parents were emitted: fix forward refs
    for (size_t q = 0; q < queue.size(); ++q) {
        const detail::TmpNode &t = b.tmp[queue[q].tmp];
        if (!is_leaf(t.left))  out.nodes[q].child0 = inner_index[t.left];
        if (!is_leaf(t.right)) out.nodes[q].child1 = inner_index[t.right];
    }
    return out;
}

} // namespace miw
