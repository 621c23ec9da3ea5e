// 8-wide BVH with quantised child boxes in an 80-byte node — the tree the wave-level phase machine walks since round 5
// (device/phased_kernel.h, NodeKind = 8; the 4-wide tree of miw/bvh4.h stays as the A/B twin and as the fallback).
//
// Why (profiles/r04_c4_*, tools/ubench/node_fetch.hip): a node step of the 4-wide walk is four scattered 16-byte requests per
// lane and one dependent round trip, and on the 0.9 M-triangle interior the wave spends 47 % of its cycles waiting for those
// round trips. An 8-wide node (after Ylitie, Karras & Laine 2017, "Efficient incoherent ray traversal on GPUs through
// compressed wide BVHs") is five requests and replaces ~1.6 steps of the 4-wide walk: fewer dependent steps per ray AND
// fewer requests per ray. What makes the node fit 80 bytes:
//   * child boxes as 8-bit planes on the node's own grid (origin + per-axis power-of-two spacing), rounded outwards;
//   * no child pointers: the inner children of a node are numbered consecutively from `child_base` in slot order (the rank
//     of a slot among the inner slots = popcount of `imask` below it), and the triangles of its leaf children lie
//     consecutively from `tri_base` in a triangle array of the tree's own order (bvh8_build.h emits the permutation), a
//     leaf slot naming its run by (offset, count) in one byte;
//   * no distance sort: children sit in the slot whose sign pattern (bit a = "on the high side of axis a") matches where
//     they lie in the node, and a ray visits the hit slots in ascending (slot XOR ray octant) — near side first.
// The walk state is two GROUPS instead of a node and a leaf range: the node group (child_base, pending hit slots | imask)
// and the triangle group (tri_base, pending triangle bits); the per-lane stack holds node groups only, one 8-byte entry
// per level (at most one push per node step: stack need <= tree depth).
//
// Like every box test here the quantised test only has to be CONSERVATIVE; every hit is decided by the exact
// Moeller-Trumbore test (+ accept rule), ties in t go to the smaller primitive id, so the observable result is the BVH2's
// and brute force's whatever the visiting order (include/mitsuba/render/kdtree.h:2079-2171: closest t; any-hit: a bool).
#pragma once
#include "base.h"
#include "scene.h"
#include "bvh.h"
#include "bvh4.h"

namespace miw {

struct alignas(16) Bvh8Node {
    float origin[3];          // lo corner of the node's box
    uint32_t exps;            // bytes 0..2: biased exponent of the per-axis plane spacing; byte 3: imask (slots holding inner nodes)
    uint32_t child_base;      // index of the first inner child (the others follow in slot order)
    uint32_t tri_base;        // first triangle (tree order) of the first leaf slot
    uint32_t meta[2];         // byte s: leaf slot: count << 5 | offset (count 1..4, offset 0..28 from tri_base); inner / absent slot: 0
    uint32_t qx[4], qy[4], qz[4];   // per axis: [0], [1] = lo planes of slots 0-3, 4-7; [2], [3] = hi planes (absent slots: lo 255, hi 0)
};
static_assert(sizeof(Bvh8Node) == 80, "Bvh8Node must be five 16-byte requests");

#define MIW_BVH8_MAX_LEAF 4u        /* triangles per leaf slot (3 bits would hold 7; 8 slots x 4 = the 32 bits of a triangle group) */
#define MIW_BVH8_STACK 16           /* 8-byte entries per lane = the 128 bytes of the 4-wide walk's 32 x 4-byte column */

// slot s of an 8-bit slot mask moves to bit (s ^ oct): bit-index XOR = conditional swaps of neighbours, pairs and nibbles
MIW_HD uint32_t bvh8_permute(uint32_t x, uint32_t oct) {
    x = (oct & 1u) ? (((x & 0x55u) << 1) | ((x >> 1) & 0x55u)) : x;
    x = (oct & 2u) ? (((x & 0x33u) << 2) | ((x >> 2) & 0x33u)) : x;
    x = (oct & 4u) ? (((x & 0x0fu) << 4) | ((x >> 4) & 0x0fu)) : x;
    return x;
}
// the ray's octant: bit a set = the ray travels towards -axis a (slots on the high side of a come first)
template <typename Ray> MIW_HD uint32_t bvh8_octant(const Ray &r) {
    return (f2u(r.inv_d.x) >> 31) | ((f2u(r.inv_d.y) >> 31) << 1) | ((f2u(r.inv_d.z) >> 31) << 2);
}
MIW_HD uint32_t bvh8_ctz(uint32_t x) { return (uint32_t) __builtin_ctz(x); }        // x != 0 at every call

// Slab-tests the eight child boxes of `n`: `hits` = bit s set when the ray's [mint, tmax_wide] interval meets the box of slot s
// (real slot numbering; absent slots may come out either way — the caller only ever reads `hits & imask` and the triangle
// bits, and an absent slot contributes to neither), `tris` = the triangle bits (offsets from n.tri_base) of the leaf slots hit.
// Same arithmetic as bvh4_test: entry / exit planes picked by the ray's direction signs, t(q) = fma(q, a, b).
template <typename Ray>
MIW_HD void bvh8_test(const Bvh8Node &n, const Ray &r, float tmax_wide, uint32_t &hits, uint32_t &tris) {
    const float sx = u2f((n.exps & 0xffu) << 23), sy = u2f(((n.exps >> 8) & 0xffu) << 23), sz = u2f(((n.exps >> 16) & 0xffu) << 23);
    const float ax = sx * r.inv_d.x, ay = sy * r.inv_d.y, az = sz * r.inv_d.z;
    const float bx = __builtin_fmaf(n.origin[0], r.inv_d.x, r.neg_o_inv_d.x),
                by = __builtin_fmaf(n.origin[1], r.inv_d.y, r.neg_o_inv_d.y),
                bz = __builtin_fmaf(n.origin[2], r.inv_d.z, r.neg_o_inv_d.z);
    const bool px = r.inv_d.x >= 0.f, py = r.inv_d.y >= 0.f, pz = r.inv_d.z >= 0.f;
    uint32_t h = 0u, tm = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int w = 0; w < 2; ++w) {
        const uint32_t nx = px ? n.qx[w] : n.qx[2 + w], fx = px ? n.qx[2 + w] : n.qx[w],
                       ny = py ? n.qy[w] : n.qy[2 + w], fy = py ? n.qy[2 + w] : n.qy[w],
                       nz = pz ? n.qz[w] : n.qz[2 + w], fz = pz ? n.qz[2 + w] : n.qz[w];
        const uint32_t mw = n.meta[w];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int c = 0; c < 4; ++c) {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef float f2_ __attribute__((ext_vector_type(2)));
            const f2_ txx = __builtin_elementwise_fma((f2_) { bvh4_byte(nx, c), bvh4_byte(fx, c) }, (f2_) { ax, ax }, (f2_) { bx, bx }),
                      tyy = __builtin_elementwise_fma((f2_) { bvh4_byte(ny, c), bvh4_byte(fy, c) }, (f2_) { ay, ay }, (f2_) { by, by }),
                      tzz = __builtin_elementwise_fma((f2_) { bvh4_byte(nz, c), bvh4_byte(fz, c) }, (f2_) { az, az }, (f2_) { bz, bz });
            const float tnx = txx.x, tfx = txx.y, tny = tyy.x, tfy = tyy.y, tnz = tzz.x, tfz = tzz.y;
#else
            const float tnx = __builtin_fmaf(bvh4_byte(nx, c), ax, bx), tfx = __builtin_fmaf(bvh4_byte(fx, c), ax, bx),
                        tny = __builtin_fmaf(bvh4_byte(ny, c), ay, by), tfy = __builtin_fmaf(bvh4_byte(fy, c), ay, by),
                        tnz = __builtin_fmaf(bvh4_byte(nz, c), az, bz), tfz = __builtin_fmaf(bvh4_byte(fz, c), az, bz);
#endif
            const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, r.mint));
            float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
            tf = __builtin_fminf(__builtin_fmaf(abs_(tf), 2e-6f, tf), tmax_wide);
            const bool hit = tn <= tf;
            const uint32_t m = (mw >> (8 * c)) & 0xffu;
            const uint32_t run = ((1u << (m >> 5)) - 1u) << (m & 31u);           // v_bfm_b32: `count` ones at `offset` (count 0: none)
            h |= hit ? (1u << (4 * w + c)) : 0u;
            tm |= hit ? run : 0u;
        }
    }
    hits = h; tris = tm;
}

// ---- the two walk bodies of the phase machine over this tree (one definition for kernel and CPU checker, like walk4_*) ----
// One lane's walk: node group (gb, gm) — gb = child_base of the group; gm bits 0..7 = hit slots still to be visited, PERMUTED
// (bit k stands for slot k ^ octant: the lowest set bit is the next slot), bits 8..15 = the group's imask (real slots), bits
// 16..18 = the ray's octant, bits 24..28 = stack depth sp; triangle group (tb, tm): tm bit i = triangle tb + i still to be
// tested. Walk over: no pending slot, no pending triangle, sp == 0. `stack[i]` = the lane's i-th entry (U2: x = gb, y = low 16 bits of gm).
// A lane takes a node step only while its triangle group is empty (triangles first: a hit shrinks tmax before the walk descends).
// Invariant between steps: the node group is empty only if the stack is (the node step pops when its node had no inner hit).
// Spec (the speculating variant, like the 4-wide walk's): a lane holding an untested triangle group may keep descending; the
// triangles a further node hands it wait in a SECOND group (tb2, tm2) and the lane stalls only while both are full.
struct Walk8 { uint32_t gb, gm, tb, tm, tb2, tm2; };
#define MIW_W8_PENDING(gm_) ((gm_) & 0xffu)
#define MIW_W8_SP(gm_) (((gm_) >> 24) & 31u)
template <typename Ray> MIW_HD void walk8_begin(Walk8 &w, const Ray &r) {
    const uint32_t oct = bvh8_octant(r);
    // the root as a group of one: node 0 = child_base 0 + rank 0; pending bit 0 stands for slot `oct`, whose rank under imask = 1 << oct is 0
    w.gb = 0u; w.gm = 1u | ((1u << oct) << 8) | (oct << 16); w.tb = 0u; w.tm = 0u; w.tb2 = 0u; w.tm2 = 0u;
}
template <bool Spec = false> MIW_HD bool walk8_node_ready(const Walk8 &w) { return MIW_W8_PENDING(w.gm) != 0u && (Spec ? w.tm2 : w.tm) == 0u; }
MIW_HD bool walk8_tri_ready(const Walk8 &w) { return w.tm != 0u; }
MIW_HD bool walk8_over(const Walk8 &w) { return (w.gm & 0x1f0000ffu) == 0u && w.tm == 0u; }
// the node the next node step of `w` fetches (the kernel issues the five loads, then calls walk8_node_step with the record)
MIW_HD uint32_t walk8_next_node(const Walk8 &w) {
    const uint32_t oct = (w.gm >> 16) & 7u, s = bvh8_ctz(MIW_W8_PENDING(w.gm)) ^ oct, imask = (w.gm >> 8) & 0xffu;
    return w.gb + (uint32_t) __builtin_popcount(imask & ((1u << s) - 1u));
}
template <typename Stack> MIW_HD void walk8_pop(Walk8 &w, Stack stack) {
    uint32_t sp = MIW_W8_SP(w.gm);
    if (MIW_W8_PENDING(w.gm) == 0u && sp != 0u) {
        --sp;
        const U2 e = stack[(int32_t) sp];
        w.gb = e.x; w.gm = (w.gm & 0x00070000u) | (e.y & 0xffffu) | (sp << 24);
    }
}
template <bool Spec = false, typename Ray, typename Stack>
MIW_HD void walk8_node_step(const Bvh8Node &n, const Ray &r, float tmax_wide, Walk8 &w, Stack stack) {
    const uint32_t oct = (w.gm >> 16) & 7u;
    uint32_t sp = MIW_W8_SP(w.gm);
    uint32_t rest = w.gm & (w.gm - 1u) & 0xffffu;                    // the slot being visited leaves the group (pending != 0: the lowest set bit is in the low byte)
    // the siblings still to be visited wait on the stack (unconditional store, predicated depth — as the 4-wide step)
    U2 e; e.x = w.gb; e.y = rest;
    stack[(int32_t) sp] = e; sp += (rest & 0xffu) ? 1u : 0u;
    uint32_t hits, tris;
    bvh8_test(n, r, tmax_wide, hits, tris);
    const uint32_t imask = n.exps >> 24;
    const uint32_t inner = bvh8_permute(hits & imask, oct);
    if (Spec) {                                                       // (T2 non-empty implies T non-empty: the first group fills first)
        const bool first = w.tm == 0u;
        w.tb2 = first ? w.tb2 : n.tri_base; w.tm2 = first ? 0u : tris;
        w.tb = first ? n.tri_base : w.tb; w.tm = first ? tris : w.tm;
    } else { w.tb = n.tri_base; w.tm = tris; }
    w.gb = n.child_base; w.gm = (oct << 16) | (imask << 8) | inner | (sp << 24);
    walk8_pop(w, stack);
}
// Triangle step: the two lowest pending triangles of the group, both records fetched up front; otherwise walk4_tri_step's
// rules (tests in ascending order, ties to the smaller primitive id, any hit ends a shadow walk).
template <bool Analytic, bool Spec = false, typename TriAt>
MIW_HD void walk8_tri_step(TriAt tri_at, const PrimCtx &ctx, V3 o, V3 d, float mint, float maxt, bool any_hit, Hit &best, float &tmax,
                           bool &occluded, Walk8 &w) {
    const uint32_t i1 = bvh8_ctz(w.tm);
    uint32_t rest = w.tm & (w.tm - 1u);
    const bool two = rest != 0u;
    const uint32_t i2 = two ? bvh8_ctz(rest) : i1;
    rest = rest & (rest - 1u);                                        // (0 & anything = 0 when there was no second one)
    const uint32_t a1 = w.tb + i1, a2 = w.tb + i2;
    Tri tr, tr2;
    tri_fetch2(tri_at, a1, a2, tr, tr2);                              // (miw/bvh4.h; the device's accessor issues the six loads together)
    float t, u, v, t2, u2, v2;
    const bool hit1 = prim_intersect<Analytic>(tr, ctx, o, d, mint, maxt, t, u, v);
    const bool hit2 = prim_intersect<Analytic>(tr2, ctx, o, d, mint, maxt, t2, u2, v2) && two;
    const bool hit = hit1 | hit2, stop = any_hit & hit;
    occluded = occluded | stop;
    bool take1 = !any_hit & hit1 & (t < best.t);
    if (!any_hit & hit1 & (t == best.t)) take1 = best.tri == MIW_MISS || tr.prim < tri_at(best.tri).prim;
    best.t = take1 ? t : best.t; best.u = take1 ? u : best.u; best.v = take1 ? v : best.v; best.tri = take1 ? a1 : best.tri;
    tmax = take1 ? t : tmax;
    bool take2 = !any_hit & hit2 & (t2 < best.t);
    if (!any_hit & hit2 & (t2 == best.t)) take2 = best.tri == MIW_MISS || tr2.prim < tri_at(best.tri).prim;
    best.t = take2 ? t2 : best.t; best.u = take2 ? u2 : best.u; best.v = take2 ? v2 : best.v; best.tri = take2 ? a2 : best.tri;
    tmax = take2 ? t2 : tmax;
    w.tm = stop ? 0u : rest;
    if (Spec) {                                                       // a drained first group hands over to the second
        const bool drained = w.tm == 0u;
        w.tb = drained ? w.tb2 : w.tb; w.tm = drained ? (stop ? 0u : w.tm2) : w.tm; w.tm2 = drained ? 0u : w.tm2;
    }
    w.gm = stop ? (w.gm & 0x00070000u) : w.gm;                         // an occluded shadow walk is over: no pending slots, empty stack
}

// Reference walk (the definition the bodies above restate): same groups, a host array as the stack, triangles before nodes.
template <bool AnyHit, typename TriAt>
MIW_HD bool bvh8_intersect(const Bvh8Node *nodes, TriAt tri_at, V3 o, V3 d, float mint, float maxt, Hit &best, PrimCtx ctx,
                           uint32_t *max_stack_seen = nullptr, uint64_t *steps = nullptr) {
    best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    const SlabRay r = slab_ray_host(o, d, mint);
    const uint32_t oct = bvh8_octant(r);
    float tmax = maxt;
    uint32_t gb = 0u, pending = 1u, imask = 1u << oct;              // the root as a group of one
    uint32_t sb[64], sm[64]; uint32_t sp = 0;
    for (;;) {
        if (pending == 0u) {
            if (sp == 0u) return best.tri != MIW_MISS;
            --sp; gb = sb[sp]; pending = sm[sp] & 0xffu; imask = sm[sp] >> 8;
        }
        const uint32_t s = bvh8_ctz(pending) ^ oct;
        pending &= pending - 1u;
        const Bvh8Node &n = nodes[gb + (uint32_t) __builtin_popcount(imask & ((1u << s) - 1u))];
        if (pending) { sb[sp] = gb; sm[sp] = pending | (imask << 8); ++sp; }
        if (max_stack_seen && sp > *max_stack_seen) *max_stack_seen = sp;
        uint32_t hits, tris;
        bvh8_test(n, r, __builtin_fmaf(abs_(tmax), 2e-6f, tmax), hits, tris);
        if (steps) { steps[0]++; steps[1] += (uint64_t) __builtin_popcount(tris); }
        while (tris) {
            const uint32_t i = n.tri_base + bvh8_ctz(tris);
            tris &= tris - 1u;
            const Tri &tr = tri_at(i);
            float t, u, v;
            if (prim_intersect(tr, ctx, o, d, mint, maxt, t, u, v)) {
                if (AnyHit) { best.t = 0.f; best.tri = i; best.prim = tr.prim; return true; }
                if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                    best.t = t; best.u = u; best.v = v; best.tri = i; best.prim = tr.prim;
                    tmax = t;
                }
            }
        }
        gb = n.child_base; imask = n.exps >> 24; pending = bvh8_permute(hits & imask, oct);
    }
}

} // namespace miw
