// 4-wide BVH with quantised child boxes — the tree the wave-level phase machine walks (device/phased_kernel.h).
//
// Why: the tree kernels are bound by dependent fetches (a node step is one L2 round trip per lane, ~170 cycles of issue in
// ~1 100 - 1 700 wall cycles per wave; the 0.9 M-triangle interior moves 4 TB/s through the fabric, profiles/r02_*). A
// 64-byte node that holds FOUR child boxes (8-bit planes relative to the node's own origin and per-axis power-of-two
// scale, after Ylitie, Karras & Laine 2017) halves the round trips and the bytes per ray against the BVH2's two full
// boxes per 64 bytes; the extra slab tests are VALU work, of which the kernels use 35 - 40 % of the issue slots.
//
// Like every box test here the quantised test only has to be CONSERVATIVE (planes are rounded outwards; the BVH2 boxes
// it is built from are already padded by 2e-5 x the scene extent, orders of magnitude above the float rounding of
// origin + q * scale): every hit is decided by the exact Moeller-Trumbore test, so the observable result is the
// BVH2's and brute force's — closest t, ties to the smaller primitive id; any-hit = some triangle passes.
#pragma once
#include "base.h"
#include "scene.h"
#include "bvh.h"

namespace miw {

#define MIW_BVH4_ABSENT ((int32_t) 0x80000000)      /* = MIW_WALK_DONE: never a leaf code, as leaf codes address fewer than 2^27 triangles (mi_scene_upload refuses more) */

struct alignas(64) Bvh4Node {
    float origin[3];          // lo corner of the node's box
    uint32_t exps;            // bytes 0..2: biased exponent e of the per-axis plane spacing 2^(e - 127); byte 3: child count
    uint32_t qlo[3];          // per axis: byte c = quantised lo plane of child c (rounded down)
    uint32_t qhi[3];          // per axis: byte c = quantised hi plane of child c (rounded up)
    int32_t child[4];         // >= 0: Bvh4Node index; < 0: leaf, ~child = (first_tri << 4) | (count - 1); absent: MIW_BVH4_ABSENT
    uint32_t pad[2];
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node must be one 64-byte line");

// The ray as the slab test wants it: t(plane) = fma(plane, inv_d, neg_o_inv_d). (FastRay of device/trace.h has the same
// fields; the host fills them with IEEE divisions — the test is conservative, not bit-reproducible, on both sides.)
struct SlabRay { V3 inv_d, neg_o_inv_d; float mint; };
MIW_HD SlabRay slab_ray_host(V3 o, V3 d, float mint) {
    SlabRay r;
    float dx = abs_(d.x) < 1e-30f ? mulsign(1e-30f, d.x) : d.x,
          dy = abs_(d.y) < 1e-30f ? mulsign(1e-30f, d.y) : d.y,
          dz = abs_(d.z) < 1e-30f ? mulsign(1e-30f, d.z) : d.z;
    r.inv_d = v3(1.f / dx, 1.f / dy, 1.f / dz);
    r.neg_o_inv_d = v3(-(o.x * r.inv_d.x), -(o.y * r.inv_d.y), -(o.z * r.inv_d.z));
    r.mint = mint;
    return r;
}

MIW_HD float bvh4_byte(uint32_t w, int c) { return (float) ((w >> (8 * c)) & 0xffu); }   // v_cvt_f32_ubyte{c} on gfx950

// Slab-tests the four child boxes of `n`. keys[c] = entry distance of child c (bits of a float >= +0: ordered like unsigned
// integers) with the child's slot in the two low mantissa bits, 0x7f800000 | slot (> every hit key) for a miss or an absent
// child. `tmax_wide` = the caller's current tmax, widened (2e-6 relative) like every fast slab test.
// The ray's octant says through which plane of an axis it enters a box (lo if it travels in +axis, hi otherwise; t(q) =
// fma(q, a, b) is monotone in q), so the entry / exit bytes are selected once per node — three word swaps — instead of a
// min / max pair per child and axis; the distances are the same floats either way.
template <typename Ray>
MIW_HD void bvh4_test(const Bvh4Node &n, const Ray &r, float tmax_wide, uint32_t keys[4]) {
    // t(q) = (origin + q * s - o) / d = q * (s * inv_d) + (origin * inv_d - o * inv_d)
    const float sx = u2f((n.exps & 0xffu) << 23), sy = u2f(((n.exps >> 8) & 0xffu) << 23), sz = u2f(((n.exps >> 16) & 0xffu) << 23);
    const float ax = sx * r.inv_d.x, ay = sy * r.inv_d.y, az = sz * r.inv_d.z;
    const float bx = __builtin_fmaf(n.origin[0], r.inv_d.x, r.neg_o_inv_d.x),
                by = __builtin_fmaf(n.origin[1], r.inv_d.y, r.neg_o_inv_d.y),
                bz = __builtin_fmaf(n.origin[2], r.inv_d.z, r.neg_o_inv_d.z);
    const bool px = r.inv_d.x >= 0.f, py = r.inv_d.y >= 0.f, pz = r.inv_d.z >= 0.f;
    const uint32_t nx = px ? n.qlo[0] : n.qhi[0], fx = px ? n.qhi[0] : n.qlo[0],
                   ny = py ? n.qlo[1] : n.qhi[1], fy = py ? n.qhi[1] : n.qlo[1],
                   nz = pz ? n.qlo[2] : n.qhi[2], fz = pz ? n.qhi[2] : n.qlo[2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int c = 0; c < 4; ++c) {
#if defined(__HIP_DEVICE_COMPILE__)
        // entry and exit distance of an axis as ONE packed fma (v_pk_fma_f32: two IEEE fmas per lane in one issue slot)
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
        const bool hit = n.child[c] != MIW_BVH4_ABSENT && tn <= tf;
        keys[c] = hit ? ((f2u(tn) & 0x7ffffffcu) | (uint32_t) c) : (0x7f800000u | (uint32_t) c);     // (the mask also clears the sign of a -0)
    }
}
// ascending keys, the child references riding along: nearest child first (5 compare-exchanges)
MIW_HD void bvh4_sort(uint32_t k[4], int32_t ch[4]) {
#define MIW_CE(a, b) do { const bool sw_ = k[b] < k[a]; const uint32_t ka_ = sw_ ? k[b] : k[a], kb_ = sw_ ? k[a] : k[b]; \
                          const int32_t ca_ = sw_ ? ch[b] : ch[a], cb_ = sw_ ? ch[a] : ch[b]; k[a] = ka_; k[b] = kb_; ch[a] = ca_; ch[b] = cb_; } while (0)
    MIW_CE(0, 1); MIW_CE(2, 3); MIW_CE(0, 2); MIW_CE(1, 3); MIW_CE(1, 2);
#undef MIW_CE
}
MIW_HD bool bvh4_key_hit(uint32_t key) { return key < 0x7f800000u; }

// ---- the two walk bodies of the wave-level phase machine (device/phased_kernel.h) ---------------------------------------
// One lane's walk is (cur, sp, leaf range [tri_i, tri_end), best hit, tmax): `cur` >= 0 is a node to visit, < 0 a leaf code still
// to be taken, MIW_BVH4_ABSENT (= the kernels' MIW_WALK_DONE) nothing. The kernel runs these two functions under its votes; the
// CPU checker runs the same two under arbitrary schedules (oracle/wavefront_emu.cpp: emu_walk4), so what the device executes per
// lane is what the CPU tier tests. `stack[i]` is the lane's i-th stack slot (the device: an LDS column, entry-major).
//
// Node step: four slab tests, the 5-exchange sort, far ... near hits onto the stack, the nearest becomes the next node. The
// stores are unconditional and only `sp` is predicated (a store past the last hit is overwritten by the next push): the lane
// writes up to slot sp + 2, which is why the collapse budgets one entry less than the column holds. A leaf becomes the lane's
// triangle range if it holds none (Spec: otherwise it stays in `cur` until the range is drained), and the next stack entry the
// current node.
// (A macro, not a function: inlined as a function the same statements cost the 128-VGPR kernel eight more spilled registers and
// the 0.9 M-triangle interior 1.5 % — round-3 session J; the function below, which the CPU checker calls, expands it as well.)
#define MIW_WALK4_NODE_STEP(SPEC, n_, r_, tmax_wide_, cur_, sp_, tri_i_, tri_end_, stack_) do {                           \
        int32_t next_ = MIW_BVH4_ABSENT;                                                                                 \
        uint32_t k_[4];                                                                                                  \
        int32_t ch_[4] = { (n_).child[0], (n_).child[1], (n_).child[2], (n_).child[3] };                                 \
        bvh4_test((n_), (r_), (tmax_wide_), k_);                                                                         \
        bvh4_sort(k_, ch_);                                                                                              \
        (stack_)[sp_] = ch_[3]; sp_ += bvh4_key_hit(k_[3]) ? 1 : 0;                                                     \
        (stack_)[sp_] = ch_[2]; sp_ += bvh4_key_hit(k_[2]) ? 1 : 0;                                                     \
        (stack_)[sp_] = ch_[1]; sp_ += bvh4_key_hit(k_[1]) ? 1 : 0;                                                     \
        if (bvh4_key_hit(k_[0])) next_ = ch_[0];                                                                         \
        else if (sp_ != 0) { --sp_; next_ = (stack_)[sp_]; }                                                             \
        if (next_ < 0 && next_ != MIW_BVH4_ABSENT && (!(SPEC) || tri_i_ >= tri_end_)) {                                  \
            const uint32_t code_ = (uint32_t) ~next_;                                                                    \
            tri_i_ = code_ >> 4; tri_end_ = tri_i_ + (code_ & 15u) + 1u;                                                 \
            next_ = MIW_BVH4_ABSENT;                                                                                     \
            if (sp_ != 0) { --sp_; next_ = (stack_)[sp_]; }                                                              \
        }                                                                                                                \
        cur_ = next_;                                                                                                    \
    } while (0)
template <bool Spec, typename Ray, typename Stack>
MIW_HD void walk4_node_step(const Bvh4Node &n, const Ray &r, float tmax_wide, int32_t &cur, int32_t &sp, uint32_t &tri_i, uint32_t &tri_end, Stack stack) {
    MIW_WALK4_NODE_STEP(Spec, n, r, tmax_wide, cur, sp, tri_i, tri_end, stack);
}

// Triangle step: two triangles of the lane's range per call — both records are fetched before either test runs (the second
// address is clamped into the range, its test predicated), so a multi-triangle leaf costs one round trip per pair; the tests run
// in range order, which is all the closest-hit / any-hit rules ask for. An any-hit walk ends at its first hit (`occluded`); a
// drained range takes over the leaf the stack handed to `cur`, if any.
// the two records of a triangle step. (The device's accessor over global memory has its own overload, device/phased_kernel.h:
// written as two plain reads the compiler — minimising live registers — issues the second record's loads only after it has
// waited for the first's, one more round trip per step, round-5 ISA reading; the overload issues all six loads, then waits once.)
template <typename TriAt> MIW_HD void tri_fetch2(TriAt tri_at, uint32_t a, uint32_t b, Tri &ta, Tri &tb) { ta = tri_at(a); tb = tri_at(b); }

template <bool Analytic, typename TriAt, typename Stack>
MIW_HD void walk4_tri_step(TriAt tri_at, const PrimCtx &ctx, V3 o, V3 d, float mint, float maxt, bool any_hit, Hit &best, float &tmax,
                           bool &occluded, int32_t &cur, int32_t &sp, uint32_t &tri_i, uint32_t &tri_end, Stack stack) {
    const bool two = tri_i + 1u < tri_end;
    Tri tr, tr2;
    tri_fetch2(tri_at, tri_i, two ? tri_i + 1u : tri_i, tr, tr2);
    float t, u, v, t2, u2, v2;
    const bool hit1 = prim_intersect<Analytic>(tr, ctx, o, d, mint, maxt, t, u, v);
    const bool hit2 = prim_intersect<Analytic>(tr2, ctx, o, d, mint, maxt, t2, u2, v2) && two;
    // The updates are selects, not branches (the lanes of a wavefront never agree on them); the only branch left is the tie, which
    // looks the best hit's primitive id up instead of carrying it through every body of the phase machine (about never taken).
    const bool hit = hit1 | hit2, stop = any_hit & hit;              // any hit ends a shadow walk
    occluded = occluded | stop;
    bool take1 = !any_hit & hit1 & (t < best.t);
    if (!any_hit & hit1 & (t == best.t)) take1 = best.tri == MIW_MISS || tr.prim < tri_at(best.tri).prim;
    best.t = take1 ? t : best.t; best.u = take1 ? u : best.u; best.v = take1 ? v : best.v; best.tri = take1 ? tri_i : best.tri;
    tmax = take1 ? t : tmax;
    bool take2 = !any_hit & hit2 & (t2 < best.t);
    if (!any_hit & hit2 & (t2 == best.t)) take2 = best.tri == MIW_MISS || tr2.prim < tri_at(best.tri).prim;
    best.t = take2 ? t2 : best.t; best.u = take2 ? u2 : best.u; best.v = take2 ? v2 : best.v; best.tri = take2 ? tri_i + 1u : best.tri;
    tmax = take2 ? t2 : tmax;
    tri_end = stop ? 0u : tri_end; cur = stop ? MIW_BVH4_ABSENT : cur; sp = stop ? 0 : sp;
    tri_i += two ? 2u : 1u;
    if (tri_i >= tri_end && cur < 0 && cur != MIW_BVH4_ABSENT) {   // range drained and the stack handed over another leaf
        const uint32_t code = (uint32_t) ~cur;
        tri_i = code >> 4; tri_end = tri_i + (code & 15u) + 1u;
        cur = MIW_BVH4_ABSENT;
        if (sp != 0) { --sp; cur = stack[sp]; }
    }
}

// Reference walk (host array stack): the definition the device bodies of phased_kernel.h restate, run by the CPU checker
// against brute force. Same observable result as bvh_intersect / brute_intersect.
template <bool AnyHit, typename TriAt>
MIW_HD bool bvh4_intersect(const Bvh4Node *nodes, TriAt tri_at, V3 o, V3 d, float mint, float maxt, Hit &best, PrimCtx ctx,
                           uint32_t *max_stack_seen = nullptr, uint64_t *steps = nullptr) {
    best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    const SlabRay r = slab_ray_host(o, d, mint);
    float tmax = maxt;
    int32_t stack[128]; int sp = 0;
    int32_t cur = 0;
    for (;;) {
        if (cur >= 0) {
            const Bvh4Node &n = nodes[cur];
            uint32_t k[4];
            int32_t ch[4] = { n.child[0], n.child[1], n.child[2], n.child[3] };
            bvh4_test(n, r, __builtin_fmaf(abs_(tmax), 2e-6f, tmax), k);
            bvh4_sort(k, ch);
            if (steps) steps[0]++;
            for (int i = 3; i >= 1; --i) if (bvh4_key_hit(k[i])) stack[sp++] = ch[i];                    // far ... near
            if (max_stack_seen && (uint32_t) sp > *max_stack_seen) *max_stack_seen = (uint32_t) sp;
            if (bvh4_key_hit(k[0])) { cur = ch[0]; continue; }
        } else {
            const uint32_t code = (uint32_t) ~cur, first = code >> 4, count = (code & 15u) + 1u;
            if (steps) steps[1] += count;
            for (uint32_t i = 0; i < count; ++i) {
                const Tri &tr = tri_at(first + i);
                float t, u, v;
                if (prim_intersect(tr, ctx, o, d, mint, maxt, t, u, v)) {
                    if (AnyHit) { best.t = 0.f; best.tri = first + i; best.prim = tr.prim; return true; }
                    if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                        best.t = t; best.u = u; best.v = v; best.tri = first + i; best.prim = tr.prim;
                        tmax = t;
                    }
                }
            }
        }
        if (sp == 0) return best.tri != MIW_MISS;
        cur = stack[--sp];
    }
}

} // namespace miw
