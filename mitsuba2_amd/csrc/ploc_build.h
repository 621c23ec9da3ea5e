// PLOC — parallel locally-ordered clustering (Meister & Bittner 2018): the agglomerative BVH2 builder behind mi_bvh_build
// quality 0 since round 4. Stands where ShapeKDTree::build() stands in the reference (src/librender/scene_native.inl:3-10) and
// where its GPU mode builds and compacts on the device (include/mitsuba/render/optix/shapes.h:72-167); like every builder here
// it only has to deliver a tree whose traversal result equals brute force (bvh.h), so none of the kd-tree's structure is
// reproduced.
//
// Why not the LBVH of lbvh_device.h alone: a radix tree splits where Morton codes differ, whatever the boxes look like — 6 %
// behind the binned-SAH tree on the 0.9 M-triangle interior, 25 % on the material balls, whose few huge wall triangles Morton
// order serves badly. PLOC keeps the Morton ORDER (one radix sort) but decides the topology by surface area: every cluster
// looks R neighbours to either side for the partner whose union has the smallest area, mutual choices merge, the survivors are
// compacted in order, repeat (~log n rounds). Tree quality is that of a full-sweep SAH build within a few per cent.
//
// This header holds the per-element steps as host + device functions: the kernels of ploc_device.h run them one element per
// thread, ploc_build_host() below runs the same functions in plain loops — the CPU tier walks the host-built tree against brute
// force (oracle/wavefront_emu.cpp: emu_set_builder), the GPU tier renders through the device-built one.
// Output = the format of the other builders: 64-byte BVH2 nodes holding both child boxes + parent links, root = node 0, leaves
// of up to `max_leaf` triangles (a subtree that small becomes one leaf: its triangles are contiguous after the final
// reordering), triangles permuted into leaf order, and the BVH2 heights the 4-wide collapse asks for (bvh4_build.h).
#pragma once
#include <vector>
#include <algorithm>
#include <cstring>
#include "miw/scene.h"
#include "miw/bvh.h"
#include "bvh_build.h"

namespace miw {

struct PlocBox { float lo[3], hi[3]; };

MIW_HD float ploc_union_half_area(const PlocBox &a, const PlocBox &b) {
    const float dx = max_(a.hi[0], b.hi[0]) - min_(a.lo[0], b.lo[0]),
                dy = max_(a.hi[1], b.hi[1]) - min_(a.lo[1], b.lo[1]),
                dz = max_(a.hi[2], b.hi[2]) - min_(a.lo[2], b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

// Node ids: inner nodes 0 .. n - 2 (handed out downwards from n - 2: the root, created last, is node 0 and every child has a
// larger id than its parent), leaves n - 1 + i for the triangle at sorted position i. All arrays below are indexed by node id.
struct PlocTree {
    PlocBox *box;            // [2 n - 1]
    uint32_t *count;         // [2 n - 1] triangles below the node (leaves: 1)
    int32_t *left, *right;   // [n - 1]
    int32_t *parent;         // [2 n - 1] (-1: the root)
    uint32_t *height;        // [n - 1] pushes a fan-out-2 walk needs below the node, fat leaves counting as leaves (bvh4_build.h)
};

// Step 1 of a round: the partner of cluster i — the cluster j within `radius` positions of i whose union with i has the smallest
// half area. Ties go to the smaller pair (min(i, j), max(i, j)): the pairs are totally ordered, so the globally smallest pair
// chooses itself from both sides and every round merges at least one pair.
MIW_HD uint32_t ploc_partner(const uint32_t *cluster, const PlocBox *box, uint32_t m, uint32_t i, uint32_t radius) {
    const PlocBox bi = box[cluster[i]];
    const uint32_t lo = i > radius ? i - radius : 0u, hi = i + radius < m - 1u ? i + radius : m - 1u;
    float best = MIW_INFINITY; uint32_t best_j = i == lo ? hi : lo;        // (m >= 2: some j != i exists)
    for (uint32_t j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const float d = ploc_union_half_area(bi, box[cluster[j]]);
        bool take = d < best;
        if (d == best) {
            const uint32_t a0 = i < j ? i : j, a1 = i < j ? j : i, b0 = i < best_j ? i : best_j, b1 = i < best_j ? best_j : i;
            take = a0 < b0 || (a0 == b0 && a1 < b1);
        }
        if (take) { best = d; best_j = j; }
    }
    return best_j;
}

// Step 2: what becomes of position i — bit 0: it survives into the next round (as itself, or as the merged node when it is the
// smaller index of a mutual pair); bit 32: it creates a node. (Packed so that ONE exclusive prefix sum yields both the
// position in the next round's array and the rank among this round's new nodes.)
MIW_HD unsigned long long ploc_flags(const uint32_t *partner, uint32_t i) {
    const uint32_t j = partner[i];
    const bool mutual = partner[j] == i;
    return (mutual && j < i) ? 0ull : ((mutual ? 1ull << 32 : 0ull) | 1ull);
}

// Step 3: write position i's survivor into the next round's array; the smaller index of a mutual pair creates the node.
// `scan` = exclusive prefix sum of ploc_flags over the round; `created` = nodes created in earlier rounds; n = triangle count.
MIW_HD void ploc_apply(const uint32_t *cluster, const uint32_t *partner, const unsigned long long *scan, uint32_t i, uint32_t created,
                       uint32_t n, uint32_t max_leaf, const PlocTree &T, uint32_t *cluster_next) {
    const unsigned long long f = ploc_flags(partner, i);
    if (!(f & 1ull)) return;
    const uint32_t pos = (uint32_t) scan[i];
    if (!(f >> 32)) { cluster_next[pos] = cluster[i]; return; }
    const uint32_t id = (n - 2u) - (created + (uint32_t) (scan[i] >> 32));
    const int32_t l = (int32_t) cluster[i], r = (int32_t) cluster[partner[i]];
    const PlocBox a = T.box[l], b = T.box[r];
    PlocBox u;
    for (int k = 0; k < 3; ++k) { u.lo[k] = min_(a.lo[k], b.lo[k]); u.hi[k] = max_(a.hi[k], b.hi[k]); }
    T.box[id] = u;
    T.left[id] = l; T.right[id] = r;
    T.parent[l] = (int32_t) id; T.parent[r] = (int32_t) id;
    const uint32_t cl = T.count[l], cr = T.count[r];
    T.count[id] = cl + cr;
    const uint32_t hl = ((uint32_t) l < n - 1u && cl > max_leaf) ? T.height[l] : 0u, hr = ((uint32_t) r < n - 1u && cr > max_leaf) ? T.height[r] : 0u;
    T.height[id] = 1u + (hl > hr ? hl : hr);
    cluster_next[pos] = id;
}

// After the last round: where a node's triangles start in leaf order = the triangle counts of the left siblings on its way up.
MIW_HD uint32_t ploc_offset(const PlocTree &T, uint32_t v) {
    uint32_t off = 0;
    int32_t c = (int32_t) v;
    for (int32_t p = T.parent[c]; p >= 0; c = p, p = T.parent[c])
        if (T.right[p] == c) off += T.count[T.left[p]];
    return off;
}

// The BvhNode of inner node `id` (offset[] = ploc_offset of every node). A child of at most max_leaf triangles is a leaf.
MIW_HD BvhNode ploc_emit(const PlocTree &T, const uint32_t *offset, uint32_t id, uint32_t n, uint32_t max_leaf) {
    const int32_t l = T.left[id], r = T.right[id];
    const PlocBox a = T.box[l], b = T.box[r];
    BvhNode nd;
    for (int k = 0; k < 3; ++k) { nd.lo0[k] = a.lo[k]; nd.hi0[k] = a.hi[k]; nd.lo1[k] = b.lo[k]; nd.hi1[k] = b.hi[k]; }
    nd.child0 = T.count[l] <= max_leaf ? bvh_leaf_code(offset[l], T.count[l]) : l;
    nd.child1 = T.count[r] <= max_leaf ? bvh_leaf_code(offset[r], T.count[r]) : r;
    nd.parent = T.parent[id]; nd.pad = 0;
    (void) n;
    return nd;
}

// 30-bit Morton code of a triangle's box centre inside the scene bounds (lbvh_device.h: k_lbvh_morton computes the same)
MIW_HD uint32_t ploc_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

struct PlocBuildResult : BvhBuildResult { std::vector<uint32_t> height; uint32_t rounds = 0; };

// The host run of the builder: same steps, plain loops. `pad` < 0: 2 x scene_pad_unit (bvh_build.h).
inline PlocBuildResult ploc_build_host(const std::vector<Tri> &tris, float pad = -1.f, uint32_t max_leaf = 4, uint32_t radius = 16) {
    PlocBuildResult out;
    const uint32_t n = (uint32_t) tris.size();
    if (n < 2 || max_leaf >= n) { static_cast<BvhBuildResult &>(out) = bvh_build_sah(tris, pad, std::max<uint32_t>(max_leaf, 1u)); out.height.assign(out.nodes.size(), 1u); return out; }
    if (max_leaf > 16) max_leaf = 16;
    if (max_leaf < 1) max_leaf = 1;
    if (pad < 0.f) pad = 2.f * scene_pad_unit(tris);
    float lo[3] = { MIW_INFINITY, MIW_INFINITY, MIW_INFINITY }, hi[3] = { -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY };
    for (const Tri &t : tris) for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(lo[a], std::min(t.p0[a], std::min(t.p1[a], t.p2[a]))); hi[a] = std::max(hi[a], std::max(t.p0[a], std::max(t.p1[a], t.p2[a])));
    }
    std::vector<unsigned long long> keys(n);
    for (uint32_t i = 0; i < n; ++i) {
        const Tri &t = tris[i]; uint32_t code = 0;
        for (int a = 0; a < 3; ++a) {
            const float c = (std::min(t.p0[a], std::min(t.p1[a], t.p2[a])) + std::max(t.p0[a], std::max(t.p1[a], t.p2[a]))) * 0.5f, ext = hi[a] - lo[a];
            const float u = ext > 0.f ? (c - lo[a]) / ext : 0.f;
            code |= ploc_expand_bits((uint32_t) std::min(std::max(u * 1024.f, 0.f), 1023.f)) << (2 - a);
        }
        keys[i] = ((unsigned long long) code << 32) | i;
    }
    std::sort(keys.begin(), keys.end());
    std::vector<PlocBox> box(2 * (size_t) n - 1); std::vector<uint32_t> count(2 * (size_t) n - 1, 1u), height(n - 1, 0u);
    std::vector<int32_t> left(n - 1, -1), right(n - 1, -1), parent(2 * (size_t) n - 1, -1);
    for (uint32_t i = 0; i < n; ++i) {
        const Tri &t = tris[(uint32_t) keys[i]]; PlocBox b;
        for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(t.p0[a], std::min(t.p1[a], t.p2[a])) - pad; b.hi[a] = std::max(t.p0[a], std::max(t.p1[a], t.p2[a])) + pad; }
        box[(n - 1) + i] = b;
    }
    PlocTree T{ box.data(), count.data(), left.data(), right.data(), parent.data(), height.data() };
    std::vector<uint32_t> cluster(n), next(n), partner(n); std::vector<unsigned long long> scan(n);
    for (uint32_t i = 0; i < n; ++i) cluster[i] = (n - 1) + i;
    uint32_t m = n, created = 0;
    while (m > 1) {
        for (uint32_t i = 0; i < m; ++i) partner[i] = ploc_partner(cluster.data(), box.data(), m, i, radius);
        unsigned long long acc = 0;
        for (uint32_t i = 0; i < m; ++i) { scan[i] = acc; acc += ploc_flags(partner.data(), i); }
        for (uint32_t i = 0; i < m; ++i) ploc_apply(cluster.data(), partner.data(), scan.data(), i, created, n, max_leaf, T, next.data());
        m = (uint32_t) acc; created += (uint32_t) (acc >> 32);
        cluster.swap(next); ++out.rounds;
    }
    std::vector<uint32_t> offset(2 * (size_t) n - 1);
    for (uint32_t v = 0; v < 2 * n - 1; ++v) offset[v] = ploc_offset(T, v);
    out.nodes.resize(n - 1);
    for (uint32_t id = 0; id < n - 1; ++id) out.nodes[id] = ploc_emit(T, offset.data(), id, n, max_leaf);
    out.tris.resize(n); out.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) { const uint32_t src = (uint32_t) keys[i], at = offset[(n - 1) + i]; out.tris[at] = tris[src]; out.order[at] = src; }
    out.height = height;
    out.depth = height[0];
    return out;
}

// Surface-area cost of a BVH2 in this node format (traversal 1.2 per inner node, intersection 1 per triangle, relative to the
// root's area; nodes inside fat leaves are unreferenced and not counted): the figure the builders are compared by in the CPU tier.
inline double bvh2_sah_cost(const std::vector<BvhNode> &nodes) {
    if (nodes.empty()) return 0.0;
    auto area = [](const float *lo, const float *hi) -> double {
        const double dx = (double) hi[0] - lo[0], dy = (double) hi[1] - lo[1], dz = (double) hi[2] - lo[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : dx * dy + dy * dz + dz * dx;
    };
    float rlo[3], rhi[3];
    for (int a = 0; a < 3; ++a) { rlo[a] = std::min(nodes[0].lo0[a], nodes[0].lo1[a]); rhi[a] = std::max(nodes[0].hi0[a], nodes[0].hi1[a]); }
    const double root = area(rlo, rhi);
    double cost = 1.2 * root;
    std::vector<int32_t> stack{ 0 };
    while (!stack.empty()) {
        const BvhNode &n = nodes[stack.back()]; stack.pop_back();
        const int32_t ch[2] = { n.child0, n.child1 }; const float *lo[2] = { n.lo0, n.lo1 }, *hi[2] = { n.hi0, n.hi1 };
        for (int k = 0; k < 2; ++k) {
            if (!(lo[k][0] <= hi[k][0])) continue;                // absent child
            const double a = area(lo[k], hi[k]);
            if (ch[k] >= 0) { cost += 1.2 * a; stack.push_back(ch[k]); }
            else cost += a * (double) ((((uint32_t) ~ch[k]) & 15u) + 1u);
        }
    }
    return root > 0 ? cost / root : 0.0;
}

} // namespace miw
