// Collapse of a BVH2 (bvh_build.h's binned SAH on the host, sah_device.h's level sweep or the radix tree on the device) into the
// 8-wide quantised tree of miw/bvh8.h. Every child box is a BVH2 child box rounded OUTWARDS onto the node's 8-bit grid and every
// BVH2 leaf survives with its triangles, so the set of triangles a ray is tested against only grows by what the coarser planes
// let through: the observable result stays the BVH2's (== brute force).
//
// A node starts from the two children of its BVH2 node and keeps opening the inner child with the largest surface area until it
// has eight (or no inner child is left). The children then go to the slot whose sign pattern matches their position in the node
// (greedy assignment on  sum_a sign_s[a] * (centroid[a] - centre[a]) ), which is what lets the walk order hits by
// (slot XOR ray octant) instead of sorting distances. Inner children are numbered consecutively in slot order (child_base +
// rank), the triangles of the leaf slots are laid out consecutively in slot order in a triangle array of the tree's own
// (tri_base + offset): `perm[new position] = position in the BVH2's leaf order` is what the caller gathers the triangle
// records (and their vertex normals / texture coordinates) with.
// Stack need of the walk = one entry per level below the root, so a tree deeper than MIW_BVH8_STACK is refused (ok = false: the
// caller keeps the 4-wide tree); so is a tree with a leaf of more than MIW_BVH8_MAX_LEAF triangles.
#pragma once
#include <vector>
#include <cmath>
#include <cstring>
#include <algorithm>
#include "miw/bvh8.h"
#include "bvh4_build.h"

namespace miw {

struct Bvh8BuildResult {
    std::vector<Bvh8Node> nodes;      // breadth-first: node 0 is the root; the inner children of a node are consecutive
    std::vector<uint32_t> perm;       // perm[i] = position in the BVH2's triangle array of triangle i of this tree's order
    uint32_t depth = 0;               // levels of inner nodes = the walk's worst-case stack entries + 1
    bool ok = false;
};

// ONE node of the collapse, shared by the host builder below and the device builder (bvh8_device.h): the Bvh8Node that stands for
// BVH2 node `ref`. Fills everything but child_base / tri_base (the caller knows where it puts the children and the triangles);
// kid_ref[s] = the BVH2 node an inner slot stands for (-1 otherwise), leaf_code[s] = the BVH2 leaf code of a leaf slot (0
// otherwise; its run starts at offset (meta byte & 31)). Returns the number of triangles under the node's leaf slots (<= 32), or
// -1 when the node cannot be built (coordinates beyond the quantisation range, a leaf of more than MIW_BVH8_MAX_LEAF triangles).
// The topology comes from the dynamic programme of Ylitie et al. 2017 (section 4.1) run bottom-up over the BVH2 first: with
// C(n, i) = the smallest total surface area of 8-wide inner nodes that represents the subtree of BVH2 node n as at most i
// roots (a leaf costs nothing here: the BVH2's leaves survive as they are, so their cost is the same in every candidate),
//   C(n, 1) = area(n) + min_k [ C(left, k) + C(right, 8 - k) ]            (n becomes one 8-wide node)
//   C(n, i) = min( C(n, i - 1), min_k [ C(left, k) + C(right, i - k) ] )  (n dissolves: its two subtrees share the i roots)
// i.e. the expected number of node steps of a random ray. (The greedy rule "open the largest child" left 3.7 of 8 slots used on
// the 0.9 M-triangle interior and saved 26 % of the 4-wide walk's steps; the programme's trees save 40 - 45 %.)
// Bvh8Dp: cost[i - 1] = C(n, i); pick bits 3(i-2) .. 3(i-2)+2 (i = 2..7) = the k of the best split into i roots, 0 = "i - 1 roots
// are as good"; bits 18..20 = the k of C(n, 1)'s split into 8.
struct Bvh8Dp { float cost[7]; uint32_t pick; };
MIW_HD float bvh8_dp_cost(const Bvh8Dp *dp, int32_t ref, int i) { return ref < 0 ? 0.f : dp[ref].cost[i - 1]; }      // leaves: 0 for every i
// one BVH2 node of the programme (children first: the caller sweeps the levels bottom-up, or in reverse breadth-first order)
MIW_HD void bvh8_dp_node(const BvhNode *n2, Bvh8Dp *dp, int32_t ref) {
    using namespace detail4;
    const BvhNode &n = n2[ref];
    const bool has0 = !absent(n.lo0, n.hi0), has1 = !absent(n.lo1, n.hi1);
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = has0 ? (has1 ? min_(n.lo0[a], n.lo1[a]) : n.lo0[a]) : n.lo1[a];
        hi[a] = has0 ? (has1 ? max_(n.hi0[a], n.hi1[a]) : n.hi0[a]) : n.hi1[a];
    }
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const float area = (has0 || has1) ? dx * dy + dy * dz + dz * dx : 0.f;
    const int32_t c0 = has0 ? n.child0 : -1, c1 = has1 ? n.child1 : -1;     // (an absent child behaves like a leaf: cost 0)
    Bvh8Dp out; out.pick = 0u;
    // distribute(j) = min over k of C(c0, k) + C(c1, j - k), k = 1 .. j - 1, both sides capped at 7 roots
    float best8 = __builtin_inff(); uint32_t k8 = 1u;
    for (int k = 1; k <= 7; ++k) {
        const float v = bvh8_dp_cost(dp, c0, k) + bvh8_dp_cost(dp, c1, 8 - k);
        if (v < best8) { best8 = v; k8 = (uint32_t) k; }
    }
    out.cost[0] = area + best8; out.pick |= k8 << 18;
    for (int i = 2; i <= 7; ++i) {
        float best = out.cost[i - 2]; uint32_t kb = 0u;
        for (int k = 1; k < i; ++k) {
            const float v = bvh8_dp_cost(dp, c0, k) + bvh8_dp_cost(dp, c1, i - k);
            if (v < best) { best = v; kb = (uint32_t) k; }
        }
        out.cost[i - 1] = best; out.pick |= kb << (3 * (i - 2));
    }
    dp[ref] = out;
}
// the children of the 8-wide node that stands for BVH2 node `ref`, read off the programme's picks: expands (subtree, roots)
// pairs until every pair is one root — a BVH2 node that becomes an 8-wide node of its own, or a leaf. Returns their number (<= 8).
MIW_HD int bvh8_dp_kids(const BvhNode *n2, const Bvh8Dp *dp, int32_t ref, detail4::Kid *kids) {
    using namespace detail4;
    struct Todo { Kid kid; int roots; };
    Todo todo[8]; int nt = 0, nk = 0;
    {
        Kid two[2]; const int m = kids_of(n2[ref], two);
        const int k = (int) ((dp[ref].pick >> 18) & 7u);
        if (m == 2) { todo[nt++] = { two[0], k }; todo[nt++] = { two[1], 8 - k }; }
        else if (m == 1) todo[nt++] = { two[0], 7 };
    }
    while (nt > 0) {
        Todo t = todo[--nt];
        if (t.kid.ref < 0) { kids[nk++] = t.kid; continue; }             // a leaf is one root whatever it was offered
        int i = t.roots > 7 ? 7 : t.roots;
        uint32_t k = 0u;
        while (i >= 2 && (k = (dp[t.kid.ref].pick >> (3 * (i - 2))) & 7u) == 0u) --i;
        if (i < 2) { kids[nk++] = t.kid; continue; }                     // one root: an 8-wide node of its own
        Kid two[2]; const int m = kids_of(n2[t.kid.ref], two);
        if (m == 2) { todo[nt++] = { two[0], (int) k }; todo[nt++] = { two[1], i - (int) k }; }
        else if (m == 1) todo[nt++] = { two[0], i };
    }
    return nk;
}

MIW_HD int bvh8_collapse_node(const BvhNode *n2, const Bvh8Dp *dp, int32_t ref, int max_fan, Bvh8Node &n, int32_t kid_ref[8], int32_t leaf_code[8]) {
    using namespace detail4;
    Kid kids[8]; int nk;
    if (dp) nk = bvh8_dp_kids(n2, dp, ref, kids);
    else {
        nk = kids_of(n2[ref], kids);
        while (nk < max_fan) {                     // (greedy twin, A/B runs and fan-out caps: open the inner child with the largest area)
            int pick = -1; float best = -1.f;
            for (int i = 0; i < nk; ++i)
                if (kids[i].ref >= 0 && half_area(kids[i]) > best) { best = half_area(kids[i]); pick = i; }
            if (pick < 0) break;
            Kid grand[2]; const int ng = kids_of(n2[kids[pick].ref], grand);
            if (nk - 1 + ng > max_fan) break;
            for (int i = pick; i + 1 < nk; ++i) kids[i] = kids[i + 1];
            --nk;
            for (int g = 0; g < ng; ++g) kids[nk++] = grand[g];
        }
    }
    const float inf = __builtin_inff();
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    for (int c = 0; c < nk; ++c) for (int a = 0; a < 3; ++a) { lo[a] = min_(lo[a], kids[c].lo[a]); hi[a] = max_(hi[a], kids[c].hi[a]); }
    if (nk == 0) { lo[0] = lo[1] = lo[2] = 0.f; hi[0] = hi[1] = hi[2] = 0.f; }
    // slots: greedy assignment, largest  sum_a (slot bit a ? + : -) * (child centre - node centre)[a]  first
    int slot_of[8], kid_in[8];
    for (int i = 0; i < 8; ++i) { slot_of[i] = -1; kid_in[i] = -1; }
    float off[8][3];
    for (int c = 0; c < nk; ++c) for (int a = 0; a < 3; ++a) off[c][a] = (kids[c].lo[a] + kids[c].hi[a]) - (lo[a] + hi[a]);   // twice the offset: same order
    for (int round = 0; round < nk; ++round) {
        int bc = -1, bs = -1; float bv = -inf;
        for (int c = 0; c < nk; ++c) {
            if (slot_of[c] >= 0) continue;
            for (int s = 0; s < 8; ++s) {
                if (kid_in[s] >= 0) continue;
                const float v = ((s & 1) ? off[c][0] : -off[c][0]) + ((s & 2) ? off[c][1] : -off[c][1]) + ((s & 4) ? off[c][2] : -off[c][2]);
                if (v > bv) { bv = v; bc = c; bs = s; }
            }
        }
        if (bc < 0) {                              // (NaN offsets: infinite coordinates) first free pair
            for (int c = 0; c < nk && bc < 0; ++c) if (slot_of[c] < 0) bc = c;
            for (int s = 0; s < 8 && bs < 0; ++s) if (kid_in[s] < 0) bs = s;
        }
        slot_of[bc] = bs; kid_in[bs] = bc;
    }
    n.child_base = 0u; n.tri_base = 0u; n.meta[0] = n.meta[1] = 0u; n.exps = 0u;
    for (int a = 0; a < 3; ++a) {
        n.origin[a] = lo[a];
        uint32_t *q = a == 0 ? n.qx : (a == 1 ? n.qy : n.qz);
        int e = 0;
        const float ext = hi[a] - lo[a];
        if (ext > 0.f) { (void) __builtin_frexpf(ext / 255.f, &e); } else e = -125;
        e = e < -125 ? -125 : (e > 126 ? 126 : e);
        for (;; ++e) {
            const float sp = u2f((uint32_t) (e + 127) << 23);
            bool fit = true;
            uint32_t w[4] = { 0u, 0u, 0u, 0u };
            for (int s = 0; s < 8 && fit; ++s) {
                int ql = 255, qh = 0;                                  // absent slots: inverted
                const int c = kid_in[s];
                if (c >= 0) {
                    const double fl = __builtin_floor(((double) kids[c].lo[a] - (double) lo[a]) / (double) sp),
                                 ch = __builtin_ceil(((double) kids[c].hi[a] - (double) lo[a]) / (double) sp);
                    if (!(ch <= 1e9)) { fit = false; break; }
                    ql = (int) fl; qh = (int) ch;
                    ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                    while (ql > 0 && __builtin_fmaf((float) ql, sp, lo[a]) > kids[c].lo[a]) --ql;
                    qh = qh > ql ? qh : ql;
                    while (qh <= 255 && __builtin_fmaf((float) qh, sp, lo[a]) < kids[c].hi[a]) ++qh;
                    if (qh > 255) { fit = false; break; }
                }
                w[s >> 2] |= (uint32_t) ql << (8 * (s & 3)); w[2 + (s >> 2)] |= (uint32_t) qh << (8 * (s & 3));
            }
            if (fit) { q[0] = w[0]; q[1] = w[1]; q[2] = w[2]; q[3] = w[3]; n.exps |= (uint32_t) (e + 127) << (8 * a); break; }
            if (e >= 126) return -1;
        }
    }
    uint32_t imask = 0u, n_tris = 0u;
    for (int s = 0; s < 8; ++s) {
        kid_ref[s] = -1; leaf_code[s] = 0;
        const int c = kid_in[s];
        if (c < 0) continue;
        if (kids[c].ref >= 0) { imask |= 1u << s; kid_ref[s] = kids[c].ref; continue; }
        const uint32_t code = (uint32_t) ~kids[c].ref, count = (code & 15u) + 1u;
        if (count > MIW_BVH8_MAX_LEAF) return -1;
        leaf_code[s] = kids[c].ref;
        n.meta[s >> 2] |= ((count << 5) | n_tris) << (8 * (s & 3));
        n_tris += count;
    }
    n.exps |= imask << 24;
    return (int) n_tris;
}

inline Bvh8BuildResult bvh8_collapse(const std::vector<BvhNode> &n2, uint32_t tri_count, int max_fan = 8) {
    Bvh8BuildResult out;
    if (n2.empty()) return out;
    max_fan = std::min(8, std::max(2, max_fan));
    // the programme, children before parents: reverse order of a breadth-first (or any parent-first) enumeration
    std::vector<Bvh8Dp> dp;
    if (max_fan == 8) {
        dp.resize(n2.size());
        std::vector<int32_t> order; order.reserve(n2.size()); order.push_back(0);
        for (size_t q = 0; q < order.size(); ++q) {
            const BvhNode &n = n2[order[q]];
            if (n.child0 >= 0 && !detail4::absent(n.lo0, n.hi0)) order.push_back(n.child0);
            if (n.child1 >= 0 && !detail4::absent(n.lo1, n.hi1)) order.push_back(n.child1);
        }
        for (size_t q = order.size(); q-- > 0;) bvh8_dp_node(n2.data(), dp.data(), order[q]);
    }
    struct Item { int32_t ref; uint32_t depth; };
    std::vector<Item> queue; queue.push_back({ 0, 1 });
    out.perm.reserve(tri_count);
    for (size_t q = 0; q < queue.size(); ++q) {                  // breadth first: the children of a node are consecutive
        const Item it = queue[q];
        out.depth = std::max(out.depth, it.depth);
        Bvh8Node n; int32_t kid_ref[8], leaf_code[8];
        const int nt = bvh8_collapse_node(n2.data(), dp.empty() ? nullptr : dp.data(), it.ref, max_fan, n, kid_ref, leaf_code);
        if (nt < 0) { out.nodes.clear(); out.perm.clear(); return out; }
        n.child_base = (uint32_t) queue.size(); n.tri_base = (uint32_t) out.perm.size();
        for (int s = 0; s < 8; ++s) {
            if (kid_ref[s] >= 0) queue.push_back({ kid_ref[s], it.depth + 1 });
            else if (leaf_code[s]) {
                const uint32_t code = (uint32_t) ~leaf_code[s], first = code >> 4, count = (code & 15u) + 1u;
                for (uint32_t j = 0; j < count; ++j) out.perm.push_back(first + j);
            }
        }
        out.nodes.push_back(n);
    }
    out.ok = out.depth <= MIW_BVH8_STACK && out.perm.size() == tri_count;
    return out;
}

} // namespace miw
