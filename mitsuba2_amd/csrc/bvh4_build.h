// Host-side collapse of a BVH2 (bvh_build.h's binned SAH, or the device LBVH read back) into the 4-wide quantised tree
// of miw/bvh4.h. Every BVH4 child box is a BVH2 child box rounded OUTWARDS onto the node's 8-bit grid, every BVH2 leaf
// survives as it is (same triangle order, same leaf codes), so the set of triangles a ray is tested against only grows by
// what the coarser planes let through — the observable result stays the BVH2's (== brute force).
//
// Fan-out: a node starts from its two BVH2 children and keeps opening the inner child with the largest surface area
// (the one a ray most likely enters anyway) until it has four — as long as the traversal stack stays inside its budget:
// a node with k children pushes up to k - 1 entries before descending, so the deepest stack is the largest sum of
// (k - 1) along a root-to-leaf path. A subtree of BVH2 height h can always be finished with h entries (fan-out 2), so
// the collapse opens a child only while  (k - 1) + height2(child)  fits every child's share of `stack_budget`; the
// returned `stack_bound` is the exact worst case of the emitted tree (<= stack_budget whenever height2(root) is).
#pragma once
#include <vector>
#include <cmath>
#include <cstring>
#include <algorithm>
#include "miw/bvh4.h"

namespace miw {

struct Bvh4BuildResult {
    std::vector<Bvh4Node> nodes;      // breadth-first: node 0 is the root
    uint32_t stack_bound = 0;         // exact worst-case stack entries of bvh4_intersect / the device node body
    uint32_t depth = 0;
    bool ok = false;                  // false: height2(root) > stack_budget (the caller keeps walking the BVH2)
};

namespace detail4 {
struct Kid { float lo[3], hi[3]; int32_t ref; };          // ref >= 0: BVH2 inner node; < 0: leaf code
MIW_HD bool absent(const float *lo, const float *hi) { return !(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]); }
MIW_HD float half_area(const Kid &k) {
    const float dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
MIW_HD int kids_of(const BvhNode &n, Kid *out) {
    int m = 0;
    if (!absent(n.lo0, n.hi0)) { Kid k; for (int a = 0; a < 3; ++a) { k.lo[a] = n.lo0[a]; k.hi[a] = n.hi0[a]; } k.ref = n.child0; out[m++] = k; }
    if (!absent(n.lo1, n.hi1)) { Kid k; for (int a = 0; a < 3; ++a) { k.lo[a] = n.lo1[a]; k.hi[a] = n.hi1[a]; } k.ref = n.child1; out[m++] = k; }
    return m;
}
} // namespace detail4

// ONE node of the collapse, shared by the host builder below and the device builder (bvh4_device.h): the BVH4 node that
// stands for BVH2 node `ref` under a stack budget of `budget` entries. h2[i] = pushes a fan-out-2 walk needs below BVH2 node
// i (its height: leaves 0). Writes the quantised child boxes, the child count and the leaf codes into `n`; the BVH2 nodes
// the inner children stand for go to kid_ref[c] (>= 0; their n.child[c] is left for the caller, who knows where it puts
// them). Returns the child count, or -1 when a coordinate range cannot be quantised (the caller keeps the BVH2).
MIW_HD int bvh4_collapse_node(const BvhNode *n2, const uint32_t *h2, int32_t ref, uint32_t budget, int max_fan, Bvh4Node &n, int32_t kid_ref[4]) {
    using namespace detail4;
    Kid kids[4]; int nk = kids_of(n2[ref], kids);
    while (nk < max_fan) {
        // open the inner child with the largest area, if every child still fits its share of the stack afterwards
        int pick = -1; float best = -1.f;
        for (int i = 0; i < nk; ++i)
            if (kids[i].ref >= 0 && half_area(kids[i]) > best) { best = half_area(kids[i]); pick = i; }
        if (pick < 0) break;
        Kid grand[2]; const int ng = kids_of(n2[kids[pick].ref], grand);
        const uint32_t k_new = (uint32_t) (nk - 1 + ng);
        bool fits = true;
        for (int i = 0; i < nk && fits; ++i)
            if (i != pick && k_new - 1 + (kids[i].ref >= 0 ? h2[kids[i].ref] : 0u) > budget) fits = false;
        for (int g = 0; g < ng; ++g) if (k_new - 1 + (grand[g].ref >= 0 ? h2[grand[g].ref] : 0u) > budget) fits = false;
        if (!fits) break;          // (a smaller child might still fit; the largest one is the one worth opening)
        for (int i = pick; i + 1 < nk; ++i) kids[i] = kids[i + 1];      // erase(pick), then the grandchildren at the end
        --nk;
        for (int g = 0; g < ng; ++g) kids[nk++] = grand[g];
    }
    n.origin[0] = n.origin[1] = n.origin[2] = 0.f; n.pad[0] = n.pad[1] = 0u;
    const uint32_t k = (uint32_t) nk;
    const float inf = __builtin_inff();
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    for (int c = 0; c < nk; ++c) for (int a = 0; a < 3; ++a) { lo[a] = min_(lo[a], kids[c].lo[a]); hi[a] = max_(hi[a], kids[c].hi[a]); }
    if (k == 0) { lo[0] = lo[1] = lo[2] = 0.f; hi[0] = hi[1] = hi[2] = 0.f; }
    n.exps = k << 24;
    for (int a = 0; a < 3; ++a) {
        n.origin[a] = lo[a];
        // plane spacing: the smallest power of two s with origin + 255 s >= hi, then outward-rounded bytes, checked in
        // the arithmetic the walk uses (fma(q, s, origin) in float)
        int e = 0;
        const float ext = hi[a] - lo[a];
        if (ext > 0.f) { (void) __builtin_frexpf(ext / 255.f, &e); } else e = -125;
        e = e < -125 ? -125 : (e > 126 ? 126 : e);
        for (;; ++e) {
            const float s = u2f((uint32_t) (e + 127) << 23);              // 2^e, e in [-125, 126]
            bool fit = true;
            uint32_t wlo = 0, whi = 0;
            for (uint32_t c = 0; c < 4 && fit; ++c) {
                int ql = 255, qh = 0;                                 // absent slots: inverted (never read: child == ABSENT)
                if (c < k) {
                    double fl = __builtin_floor(((double) kids[c].lo[a] - (double) lo[a]) / (double) s),
                           ch = __builtin_ceil(((double) kids[c].hi[a] - (double) lo[a]) / (double) s);
                    if (!(ch <= 1e9)) { fit = false; break; }         // beyond any byte at this spacing (also keeps the casts defined)
                    ql = (int) fl; qh = (int) ch;
                    ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                    while (ql > 0 && __builtin_fmaf((float) ql, s, lo[a]) > kids[c].lo[a]) --ql;
                    qh = qh > ql ? qh : ql;
                    while (qh <= 255 && __builtin_fmaf((float) qh, s, lo[a]) < kids[c].hi[a]) ++qh;
                    if (qh > 255) { fit = false; break; }
                }
                wlo |= (uint32_t) ql << (8 * c); whi |= (uint32_t) qh << (8 * c);
            }
            if (fit) { n.qlo[a] = wlo; n.qhi[a] = whi; n.exps |= (uint32_t) (e + 127) << (8 * a); break; }
            if (e >= 126) return -1;                                  // coordinates beyond float range: keep the BVH2
        }
    }
    for (uint32_t c = 0; c < 4; ++c) { n.child[c] = MIW_BVH4_ABSENT; kid_ref[c] = -1; }
    for (uint32_t c = 0; c < k; ++c) {
        if (kids[c].ref < 0) n.child[c] = kids[c].ref;
        else kid_ref[c] = kids[c].ref;
    }
    return nk;
}

// BVH2 heights (pushes a fan-out-2 walk needs below a node): leaves 0; a node whose children are both leaves still pushes
// one entry (the far leaf), so height = (two children ? 1 : 0) + max(child heights).
inline std::vector<uint32_t> bvh2_heights(const std::vector<BvhNode> &n2) {
    using namespace detail4;
    std::vector<uint32_t> h2(n2.size(), 0);
    if (n2.empty()) return h2;
    std::vector<std::pair<int32_t, int>> st; st.push_back({ 0, 0 });
    while (!st.empty()) {
        const int32_t i = st.back().first;
        const BvhNode &n = n2[i];
        if (st.back().second == 0) {
            st.back().second = 1;
            if (n.child0 >= 0 && !absent(n.lo0, n.hi0)) st.push_back({ n.child0, 0 });
            if (n.child1 >= 0 && !absent(n.lo1, n.hi1)) st.push_back({ n.child1, 0 });
        } else {
            uint32_t h = 0;
            if (n.child0 >= 0 && !absent(n.lo0, n.hi0)) h = std::max(h, h2[n.child0]);
            if (n.child1 >= 0 && !absent(n.lo1, n.hi1)) h = std::max(h, h2[n.child1]);
            const bool two = !absent(n.lo0, n.hi0) && !absent(n.lo1, n.hi1);
            h2[i] = h + (two ? 1u : 0u);
            st.pop_back();
        }
    }
    return h2;
}

inline Bvh4BuildResult bvh4_collapse(const std::vector<BvhNode> &n2, uint32_t stack_budget, int max_fan = 4) {
    Bvh4BuildResult out;
    if (n2.empty()) return out;
    if (max_fan < 2) max_fan = 2;
    if (max_fan > 4) max_fan = 4;
    const std::vector<uint32_t> h2 = bvh2_heights(n2);
    if (h2[0] > stack_budget) return out;

    struct Item { int32_t ref; uint32_t budget, depth; };
    std::vector<Item> queue; queue.push_back({ 0, stack_budget, 1 });
    for (size_t q = 0; q < queue.size(); ++q) {                  // breadth first = level by level, as the device builder emits them
        const Item it = queue[q];
        out.depth = std::max(out.depth, it.depth);
        Bvh4Node n; int32_t kid_ref[4];
        const int k = bvh4_collapse_node(n2.data(), h2.data(), it.ref, it.budget, max_fan, n, kid_ref);
        if (k < 0) { out.nodes.clear(); return out; }
        for (int c = 0; c < k; ++c)
            if (kid_ref[c] >= 0) { n.child[c] = (int32_t) queue.size(); queue.push_back({ kid_ref[c], it.budget - (uint32_t) (k - 1), it.depth + 1 }); }
        out.nodes.push_back(n);
    }
    // exact stack bound, bottom-up (children have larger indices than their parents)
    std::vector<uint32_t> used_below(out.nodes.size(), 0);
    for (size_t i = out.nodes.size(); i-- > 0;) {
        const Bvh4Node &n = out.nodes[i];
        const uint32_t k = n.exps >> 24;
        uint32_t below = 0;
        for (uint32_t c = 0; c < k; ++c) if (n.child[c] >= 0) below = std::max(below, used_below[n.child[c]]);
        used_below[i] = (k ? k - 1 : 0) + below;
    }
    out.stack_bound = used_below[0];
    out.ok = out.stack_bound <= stack_budget;
    return out;
}

} // namespace miw
