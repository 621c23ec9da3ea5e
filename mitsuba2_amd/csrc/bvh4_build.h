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
inline bool absent(const float *lo, const float *hi) { return !(lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]); }
inline float half_area(const Kid &k) {
    const float dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
inline void kids_of(const BvhNode &n, std::vector<Kid> &out) {
    if (!absent(n.lo0, n.hi0)) { Kid k; memcpy(k.lo, n.lo0, 12); memcpy(k.hi, n.hi0, 12); k.ref = n.child0; out.push_back(k); }
    if (!absent(n.lo1, n.hi1)) { Kid k; memcpy(k.lo, n.lo1, 12); memcpy(k.hi, n.hi1, 12); k.ref = n.child1; out.push_back(k); }
}
} // namespace detail4

inline Bvh4BuildResult bvh4_collapse(const std::vector<BvhNode> &n2, uint32_t stack_budget, int max_fan = 4) {
    using namespace detail4;
    Bvh4BuildResult out;
    if (n2.empty()) return out;
    if (max_fan < 2) max_fan = 2;
    if (max_fan > 4) max_fan = 4;
    // BVH2 heights (pushes a fan-out-2 walk needs below a node): leaves 0, inner 1 + max over inner children ... a node
    // whose children are both leaves still pushes one entry (the far leaf), so height = 1 + max(child heights).
    std::vector<uint32_t> h2(n2.size(), 0);
    {
        std::vector<std::pair<int32_t, int>> st; st.push_back({ 0, 0 });
        while (!st.empty()) {
            auto &[i, phase] = st.back();
            const BvhNode &n = n2[i];
            if (phase == 0) {
                phase = 1;
                const int32_t me = i;                      // `i` dangles once the vector grows
                if (n2[me].child0 >= 0 && !absent(n2[me].lo0, n2[me].hi0)) st.push_back({ n2[me].child0, 0 });
                if (n2[me].child1 >= 0 && !absent(n2[me].lo1, n2[me].hi1)) st.push_back({ n2[me].child1, 0 });
            } else {
                uint32_t h = 0;
                if (n.child0 >= 0 && !absent(n.lo0, n.hi0)) h = std::max(h, h2[n.child0]);
                if (n.child1 >= 0 && !absent(n.lo1, n.hi1)) h = std::max(h, h2[n.child1]);
                const bool two = !absent(n.lo0, n.hi0) && !absent(n.lo1, n.hi1);
                h2[i] = h + (two ? 1u : 0u);
                st.pop_back();
            }
        }
    }
    if (h2[0] > stack_budget) return out;
    auto height = [&](const Kid &k) -> uint32_t { return k.ref >= 0 ? h2[k.ref] : 0u; };

    struct Item { int32_t ref; uint32_t budget, depth; };
    std::vector<Item> queue; queue.push_back({ 0, stack_budget, 1 });
    std::vector<Kid> kids; kids.reserve(8);
    std::vector<uint32_t> used_below;                        // per emitted node: filled bottom-up for stack_bound
    for (size_t q = 0; q < queue.size(); ++q) {
        const Item it = queue[q];
        out.depth = std::max(out.depth, it.depth);
        kids.clear();
        kids_of(n2[it.ref], kids);
        while ((int) kids.size() < max_fan) {
            // open the inner child with the largest area, if every child still fits its share of the stack afterwards
            int pick = -1; float best = -1.f;
            for (int i = 0; i < (int) kids.size(); ++i)
                if (kids[i].ref >= 0 && half_area(kids[i]) > best) { best = half_area(kids[i]); pick = i; }
            if (pick < 0) break;
            std::vector<Kid> grand; kids_of(n2[kids[pick].ref], grand);
            const uint32_t k_new = (uint32_t) (kids.size() - 1 + grand.size());
            bool fits = true;
            for (int i = 0; i < (int) kids.size() && fits; ++i)
                if (i != pick && k_new - 1 + height(kids[i]) > it.budget) fits = false;
            for (const Kid &g : grand) if (k_new - 1 + height(g) > it.budget) fits = false;
            if (!fits) break;      // (a smaller child might still fit; the largest one is the one worth opening)
            kids.erase(kids.begin() + pick);
            for (const Kid &g : grand) kids.push_back(g);
        }

        Bvh4Node n; memset(&n, 0, sizeof n);
        const uint32_t k = (uint32_t) kids.size();
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (const Kid &c : kids) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c.lo[a]); hi[a] = std::max(hi[a], c.hi[a]); }
        if (k == 0) { lo[0] = lo[1] = lo[2] = 0.f; hi[0] = hi[1] = hi[2] = 0.f; }
        n.exps = k << 24;
        for (int a = 0; a < 3; ++a) {
            n.origin[a] = lo[a];
            // plane spacing: the smallest power of two s with origin + 255 s >= hi, then outward-rounded bytes, checked in
            // the arithmetic the walk uses (fma(q, s, origin) in float)
            int e = 0;
            const float ext = hi[a] - lo[a];
            if (ext > 0.f) { std::frexp(ext / 255.f, &e); } else e = -125;
            e = std::max(-125, std::min(126, e));
            for (;; ++e) {
                const float s = std::ldexp(1.f, e);
                bool fit = true;
                uint32_t wlo = 0, whi = 0;
                for (uint32_t c = 0; c < 4 && fit; ++c) {
                    int ql = 255, qh = 0;                                 // absent slots: inverted (never read: child == ABSENT)
                    if (c < k) {
                        ql = (int) std::floor(((double) kids[c].lo[a] - (double) lo[a]) / (double) s);
                        qh = (int) std::ceil(((double) kids[c].hi[a] - (double) lo[a]) / (double) s);
                        ql = std::max(0, std::min(255, ql));
                        while (ql > 0 && std::fmaf((float) ql, s, lo[a]) > kids[c].lo[a]) --ql;
                        qh = std::max(qh, ql);
                        while (qh <= 255 && std::fmaf((float) qh, s, lo[a]) < kids[c].hi[a]) ++qh;
                        if (qh > 255) { fit = false; break; }
                    }
                    wlo |= (uint32_t) ql << (8 * c); whi |= (uint32_t) qh << (8 * c);
                }
                if (fit) { n.qlo[a] = wlo; n.qhi[a] = whi; n.exps |= (uint32_t) (e + 127) << (8 * a); break; }
                if (e >= 126) { out.nodes.clear(); return out; }          // coordinates beyond float range: keep the BVH2
            }
        }
        for (uint32_t c = 0; c < 4; ++c) n.child[c] = MIW_BVH4_ABSENT;
        for (uint32_t c = 0; c < k; ++c) {
            if (kids[c].ref < 0) n.child[c] = kids[c].ref;
            else { n.child[c] = (int32_t) queue.size(); queue.push_back({ kids[c].ref, it.budget - (k - 1), it.depth + 1 }); }
        }
        out.nodes.push_back(n);
    }
    // exact stack bound, bottom-up (children have larger indices than their parents)
    used_below.assign(out.nodes.size(), 0);
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
