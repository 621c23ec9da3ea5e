// Stackless BVH2 traversal — the replacement for ShapeKDTree::ray_intersect_scalar
// (include/mitsuba/render/kdtree.h:2079-2171) + intersect_prim (:2362-2391).
//
// Only the kd-tree's *observable* result is reproduced: over all triangles that
// pass Mesh::ray_intersect_triangle (mesh.h:194-226) inside [mint, maxt], the
// closest hit; ties in t are broken by the smaller global primitive id (the
// reference's tie order depends on kd-tree leaf order and is not reproducible by
// any other structure; the CPU checker uses the same rule). Any-hit = "some
// triangle passes".
//
// Traversal keeps no stack: a 64-bit trail holds one bit per inner node on the
// root-to-current path ("far child still pending"); backtracking follows parent
// links and re-derives near/far from the node (order depends only on the box
// entry distances, never on the shrinking maxt). State per lane: node index +
// trail = 3 VGPRs, no LDS/scratch stack, depth <= 62.
//
// Boxes are padded at build time and the slab test is widened, so a triangle the
// Moeller-Trumbore test accepts is never culled (result == brute force).
#pragma once
#include "base.h"
#include "scene.h"

namespace miw {

#define MIW_BVH_MAX_DEPTH 62

MIW_HD int32_t bvh_leaf_code(uint32_t first, uint32_t count) { return ~(int32_t) ((first << 4) | (count - 1)); }

struct RayPrep { V3 o, d, inv_d; float mint, maxt; };

MIW_HD RayPrep ray_prepare(V3 o, V3 d, float mint, float maxt) {
    RayPrep r; r.o = o; r.d = d; r.mint = mint; r.maxt = maxt;
    // never feed inf/NaN to the slab test: a zero component becomes +-1e-30
    float dx = abs_(d.x) < 1e-30f ? mulsign(1e-30f, d.x) : d.x,
          dy = abs_(d.y) < 1e-30f ? mulsign(1e-30f, d.y) : d.y,
          dz = abs_(d.z) < 1e-30f ? mulsign(1e-30f, d.z) : d.z;
    r.inv_d = v3(1.f / dx, 1.f / dy, 1.f / dz);
    return r;
}

// Conservative slab test; tnear is clamped by mint only (order key), the
// accept test uses the caller's current tmax.
MIW_HD bool box_test(const float *lo, const float *hi, const RayPrep &r, float tmax, float &tnear) {
    float t0x = (lo[0] - r.o.x) * r.inv_d.x, t1x = (hi[0] - r.o.x) * r.inv_d.x,
          t0y = (lo[1] - r.o.y) * r.inv_d.y, t1y = (hi[1] - r.o.y) * r.inv_d.y,
          t0z = (lo[2] - r.o.z) * r.inv_d.z, t1z = (hi[2] - r.o.z) * r.inv_d.z;
    float tn = max_(max_(min_(t0x, t1x), min_(t0y, t1y)), max_(min_(t0z, t1z), r.mint));
    float tf = min_(min_(max_(t0x, t1x), max_(t0y, t1y)), max_(t0z, t1z));
    tnear = tn;
    // widen: far side by 4 ulp, and let near exceed tmax by 4 ulp (tie candidates)
    return tn <= tf * 1.0000005f && tn <= tmax * 1.0000005f;
}

struct Hit { float t, u, v; uint32_t tri; uint32_t prim; };

// Accessors: NodeAt(i) -> const BvhNode&, TriAt(i) -> const Tri&. The device
// kernels pass LDS-staged accessors; the CPU checker passes plain arrays.
template <bool AnyHit, typename NodeAt, typename TriAt>
MIW_HD bool bvh_intersect(NodeAt node_at, TriAt tri_at, const RayPrep &r, Hit &best, PrimCtx ctx) {
    best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    float tmax = r.maxt;           // shrinks to the best t (closest-hit only)
    int32_t cur = 0;
    uint64_t trail = 2;            // bit1 = sentinel, bit0 = root's pending flag
    bool fresh = true;

    for (;;) {
        int32_t visit = 0; bool have_visit = false;

        if (fresh) {
            const BvhNode &n = node_at(cur);
            float tn0, tn1;
            bool h0 = box_test(n.lo0, n.hi0, r, tmax, tn0),
                 h1 = box_test(n.lo1, n.hi1, r, tmax, tn1);
            if (h0 && h1) {
                bool second_first = tn1 < tn0;
                trail |= 1ull;
                visit = second_first ? n.child1 : n.child0; have_visit = true;
            } else if (h0 || h1) {
                visit = h0 ? n.child0 : n.child1; have_visit = true;
            }
        }

        // backtrack until a pending far child is found (or the trail is empty)
        while (!have_visit) {
            if (trail & 1ull) {
                trail &= ~1ull;
                const BvhNode &n = node_at(cur);
                float tn0, tn1;
                bool h0 = box_test(n.lo0, n.hi0, r, tmax, tn0),
                     h1 = box_test(n.lo1, n.hi1, r, tmax, tn1);
                bool second_first = tn1 < tn0;       // same key as on the way down
                bool far_hit = second_first ? h0 : h1;
                if (far_hit) { visit = second_first ? n.child0 : n.child1; have_visit = true; }
            } else {
                trail >>= 1;
                if (trail == 1ull) return best.tri != MIW_MISS;
                cur = node_at(cur).parent;
            }
        }

        if (visit < 0) {                               // leaf: test its triangles
            uint32_t code = (uint32_t) ~visit, first = code >> 4, count = (code & 15u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const Tri &tr = tri_at(first + i);
                float t, u, v;
                if (prim_intersect(tr, ctx, r.o, r.d, r.mint, r.maxt, t, u, v)) {
                    if (AnyHit) { best.t = 0.f; best.tri = first + i; best.prim = tr.prim; return true; }
                    if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                        best.t = t; best.u = u; best.v = v; best.tri = first + i; best.prim = tr.prim;
                        tmax = t;
                    }
                }
            }
            fresh = false;                             // stay at `cur`, go backtrack
        } else {
            cur = visit; trail <<= 1; fresh = true;
        }
    }
}

// Brute force over all triangles: the definition the BVH must reproduce.
template <bool AnyHit, typename TriAt>
MIW_HD bool brute_intersect(TriAt tri_at, uint32_t tri_count, V3 o, V3 d, float mint, float maxt, Hit &best, PrimCtx ctx) {
    best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    for (uint32_t i = 0; i < tri_count; ++i) {
        const Tri &tr = tri_at(i);
        float t, u, v;
        if (prim_intersect(tr, ctx, o, d, mint, maxt, t, u, v)) {
            if (AnyHit) { best.t = 0.f; best.tri = i; best.prim = tr.prim; return true; }
            if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                best.t = t; best.u = u; best.v = v; best.tri = i; best.prim = tr.prim;
            }
        }
    }
    return best.tri != MIW_MISS;
}

} // namespace miw
