// CHECKER — TEST INFRASTRUCTURE ONLY (same rules as miw_oracle.cpp).
//
// Runs the product's per-lane wavefront stages (mitsuba2_amd/csrc/miw/path.h,
// bvh.h, bvh_build.h — the very functions the gfx950 kernels wrap) in plain
// loops on the CPU. It exists so that the wavefront re-ordering of
// PathIntegrator::sample, the lane <-> pixel/seed mapping, the shadow-queue
// protocol and the stackless BVH can be compared with the scalar oracle without
// a GPU; it is NOT a product fallback and nothing in the package loads it.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <vector>
#include <xmmintrin.h>
#include <pmmintrin.h>

#include "../include/miwave.h"
#include "../mitsuba2_amd/csrc/miw/path.h"
#include "../mitsuba2_amd/csrc/miw/direct.h"
#include "../mitsuba2_amd/csrc/miw/film_gather.h"
#include "../mitsuba2_amd/csrc/film_classes.h"
#include "../mitsuba2_amd/csrc/miw/bvh.h"
#include "../mitsuba2_amd/csrc/bvh_build.h"
#include "../mitsuba2_amd/csrc/bvh4_build.h"
#include "../mitsuba2_amd/csrc/bvh8_build.h"
#include "../mitsuba2_amd/csrc/sah_levels.h"
#include "../mitsuba2_amd/csrc/envmap_build.h"
#include "../mitsuba2_amd/csrc/rect_build.h"
#include "../mitsuba2_amd/csrc/texture_build.h"

using namespace miw;

namespace {
struct EmuScene {
    std::vector<Tri> tris_in; std::vector<float> vn_in;
    std::vector<ShapeRec> shapes; std::vector<BsdfRec> bsdfs; std::vector<EmitterRec> emitters; std::vector<AnalyticRec> rects;
    std::vector<float> emit_tri, emit_vnorm, emit_pmf, emit_cdf;
    BvhBuildResult bvh; std::vector<float> vn_leaf, tri_uv; std::vector<BitmapRec> bitmaps;
    EnvmapTables env;
    SceneView view{};
};

bool emu_build(const mi_scene_desc *s, EmuScene &o, int max_leaf) {
    o.tris_in.assign(s->face_count, Tri{});
    o.shapes.resize(s->shape_count);
    bool any_normals = false;
    for (uint32_t i = 0; i < s->shape_count; ++i) {
        const mi_shape &sh = s->shapes[i];
        int32_t emitter_id = sh.emitter;
        if (emitter_id >= 0 && s->envmap && (uint32_t) emitter_id >= s->envmap->emitter_index) emitter_id += 1;
        o.shapes[i] = ShapeRec{ sh.bsdf, emitter_id, sh.flags & (SHAPE_HAS_NORMALS | SHAPE_HAS_TEXCOORDS), 0 };
        any_normals = any_normals || (sh.flags & 1u);
        for (uint32_t f = sh.first_face; f < sh.first_face + sh.face_count; ++f) o.tris_in[f].shape = i;
    }
    // analytic rectangles: record + two bounding triangles (the second one behind the faces), as mi_scene_upload
    o.rects.clear();
    o.tris_in.resize((size_t) s->face_count + s->rectangle_count + s->sphere_count);
    for (uint32_t k = 0; k < s->rectangle_count; ++k) {
        const mi_rectangle &q = s->rectangles[k];
        const uint32_t f = s->shapes[q.shape].first_face;
        o.rects.push_back(rect_record(q.to_world, q.to_object, q.shape, f));
        Tri two[2];
        rect_bounding_tris(o.rects.back(), k, two);
        o.tris_in[f] = two[0]; o.tris_in[(size_t) s->face_count + k] = two[1];
    }
    for (uint32_t k = 0; k < s->sphere_count; ++k) {
        const mi_sphere &q = s->spheres[k];
        const uint32_t f = s->shapes[q.shape].first_face, idx = (uint32_t) o.rects.size();
        o.rects.push_back(sphere_record(q.center, q.radius, q.flip_normals != 0, q.to_world, q.to_object, q.shape, f));
        Tri two[2];
        sphere_bounding_tris(o.rects.back(), idx, two);
        o.tris_in[f] = two[0]; o.tris_in[(size_t) s->face_count + idx] = two[1];
    }
    if (any_normals) o.vn_in.assign(o.tris_in.size() * 9, 0.f);
    for (uint32_t f = 0; f < s->face_count; ++f) {
        Tri &t = o.tris_in[f];
        if (t.pad) continue;
        for (int k = 0; k < 3; ++k) {
            uint32_t vi = s->faces[3 * f + k];
            float *dst = k == 0 ? t.p0 : (k == 1 ? t.p1 : t.p2);
            std::memcpy(dst, s->vertex_positions + 3 * (size_t) vi, 12);
            if (any_normals && (o.shapes[t.shape].flags & 1u))
                std::memcpy(&o.vn_in[(size_t) f * 9 + 3 * k], s->vertex_normals + 3 * (size_t) vi, 12);
        }
        t.prim = f; t.pad = 0;
    }
    o.bsdfs.resize(s->bsdf_count);
    for (uint32_t i = 0; i < s->bsdf_count; ++i) {
        int slot = 0;
        if (bsdf_record_from_abi(s->bsdfs[i], s->bitmap_count, s->bsdf_table_floats, o.bsdfs[i], &slot)) return false;
    }
    { uint32_t bad = 0; if (build_bitmap_table(s, o.bitmaps, &bad)) return false; }
    bool emit_normals = false;
    auto push_env = [&]() { EmitterRec r; std::memset(&r, 0, sizeof r); r.type = EMITTER_ENVMAP; r.shape = 0xffffffffu; o.emitters.push_back(r); };
    for (uint32_t i = 0; i < s->emitter_count; ++i) {
        if (s->envmap && s->envmap->emitter_index == i) push_env();
        const mi_emitter &e = s->emitters[i];
        const mi_shape &sh = s->shapes[e.shape];
        EmitterRec r; std::memset(&r, 0, sizeof r);
#if MIW_SPECTRAL
        std::memcpy(&r.radiance, &e.radiance_tex, sizeof(TexRec));
#else
        r.radiance.type = TEX_RGB; std::memcpy(r.radiance.v, e.radiance, 12);
#endif
        if (sh.flags & (MI_SHAPE_RECTANGLE | MI_SHAPE_SPHERE)) {
            const uint32_t k = o.tris_in[sh.first_face].pad - 1u;
            r.shape = e.shape; r.tri_first = k; r.tri_count = 0; r.flags = 2u;
            r.normalization = o.rects[k].inv_area; r.sum = rcp(o.rects[k].inv_area);
            o.emitters.push_back(r);
            continue;
        }
        r.shape = e.shape; r.tri_first = (uint32_t) o.emit_pmf.size(); r.tri_count = sh.face_count; r.flags = sh.flags & 1u;
        emit_normals = emit_normals || r.flags;
        double sum = 0.0; uint32_t vlo = 0xffffffffu, vhi = 0;
        for (uint32_t k = 0; k < sh.face_count; ++k) {
            const Tri &t = o.tris_in[sh.first_face + k];
            float area = face_area(ld3(t.p0), ld3(t.p1), ld3(t.p2));
            o.emit_pmf.push_back(area); sum += (double) area; o.emit_cdf.push_back((float) sum);
            if (area > 0.f) { if (vlo == 0xffffffffu) vlo = k; vhi = k; }
            o.emit_tri.insert(o.emit_tri.end(), t.p0, t.p0 + 3);
            o.emit_tri.insert(o.emit_tri.end(), t.p1, t.p1 + 3);
            o.emit_tri.insert(o.emit_tri.end(), t.p2, t.p2 + 3);
            for (int q = 0; q < 9; ++q) o.emit_vnorm.push_back(r.flags ? o.vn_in[(size_t) (sh.first_face + k) * 9 + q] : 0.f);
        }
        if (vlo == 0xffffffffu) return false;
        r.valid_lo = vlo; r.valid_hi = vhi; r.sum = (float) sum; r.normalization = (float) (1.0 / sum);
        o.emitters.push_back(r);
    }
    if (s->envmap && s->envmap->emitter_index >= s->emitter_count) push_env();
    if (s->envmap) {
        o.env = envmap_build(*s->envmap);
        if (!o.env.ok) return false;
        o.env.rec.data = o.env.data.data(); o.env.rec.levels = o.env.levels.data();
    }
    o.bvh = bvh_build_sah(o.tris_in, -1.f, (uint32_t) max_leaf);
    if (!o.vn_in.empty()) {
        o.vn_leaf.resize(o.vn_in.size());
        for (size_t i = 0; i < o.bvh.order.size(); ++i) std::memcpy(&o.vn_leaf[i * 9], &o.vn_in[(size_t) o.bvh.order[i] * 9], 36);
    }
    SceneView &v = o.view;
    v.nodes = o.bvh.nodes.data(); v.node_count = (uint32_t) o.bvh.nodes.size();
    v.tris = o.bvh.tris.data(); v.tri_count = (uint32_t) o.bvh.tris.size();
    v.tri_vn = o.vn_leaf.empty() ? nullptr : o.vn_leaf.data();
    if (!build_face_texcoords(s, o.tri_uv)) return false;
    v.tri_uv = o.tri_uv.empty() ? nullptr : o.tri_uv.data();
    v.bitmaps = o.bitmaps.empty() ? nullptr : o.bitmaps.data();
    v.bsdf_tables = s->bsdf_table_floats ? s->bsdf_tables : nullptr;   // the caller's array
    v.shapes = o.shapes.data(); v.shape_count = (uint32_t) o.shapes.size();
    v.bsdfs = o.bsdfs.data(); v.bsdf_count = (uint32_t) o.bsdfs.size();
    v.emitters = o.emitters.data(); v.emitter_count = (uint32_t) o.emitters.size();
    scene_view_prepare(v);
    v.emit_tri = o.emit_tri.data(); v.emit_vnorm = emit_normals ? o.emit_vnorm.data() : nullptr;
    v.emit_pmf = o.emit_pmf.data(); v.emit_cdf = o.emit_cdf.data();
    v.env = s->envmap ? &o.env.rec : nullptr; v.leaf_boxes = nullptr; v.nodes4 = nullptr; v.nodes8 = nullptr;
    v.rects = o.rects.empty() ? nullptr : o.rects.data(); v.rect_count = (uint32_t) o.rects.size();
    v.accept_pad = scene_pad_unit(o.tris_in); v.tri_bounds = nullptr;
    return true;
}
// plain IEEE float environment (denormals preserved), see miw_oracle.cpp FtzScope
struct Ftz { unsigned csr; Ftz() { csr = _mm_getcsr(); _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_OFF); _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_OFF); } ~Ftz() { _mm_setcsr(csr); } };
// One ray through the per-lane bodies of the wave-level phase machine — walk4_node_step / walk4_tri_step of csrc/miw/bvh4.h, the
// two functions k_path_phased (csrc/device/phased_kernel.h) runs under its votes — with the vote replaced by a coin: whenever a
// lane could take either body (Spec: it holds an untested leaf range AND a node to descend into) `coin()` picks one, so every
// interleaving the device can produce is reachable. The stack is the lane's column: `cap` slots, every access checked (the node
// step stores up to two slots past `sp`; the collapse budgets one slot less than the column holds, csrc/miwave.hip).
// Returns hit / occluded; *bad is set when a slot outside the column was touched.
struct EmuColumn {
    int32_t *p; int32_t cap; bool *bad; uint32_t *deepest;
    int32_t &operator[](int32_t i) const {
        if (i < 0 || i >= cap) { *bad = true; return p[cap]; }                 // p holds cap + 1 entries: the last one absorbs strays
        if ((uint32_t) i + 1u > *deepest) *deepest = (uint32_t) i + 1u;
        return p[i];
    }
};
template <bool Spec, typename TriAt, typename Coin>
bool emu_walk4(const Bvh4Node *nodes4, TriAt tri_at, const PrimCtx &ctx, V3 o, V3 d, float mint, float maxt, bool any_hit, Hit &best,
               Coin coin, int32_t cap, bool *bad, uint32_t *deepest) {
    const SlabRay r = slab_ray_host(o, d, mint);
    int32_t column[129];                                        // cap <= 128 (emu_trace4 refuses budgets above 127)
    const EmuColumn stack{ column, cap, bad, deepest };
    int32_t cur = 0, sp = 0;
    uint32_t tri_i = 0, tri_end = 0;
    float tmax = maxt;
    bool occluded = false;
    if (!any_hit) { best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu; }   // (an S walk never writes `best`)
    for (;;) {
        const bool has_range = tri_i < tri_end;
        const bool e_node = cur >= 0 && (Spec || !has_range), e_leaf = has_range;       // the predicates of the kernel's vote
        if (!e_node && !e_leaf) break;                                                   // walk_over
        if (e_node && (!e_leaf || coin()))
            walk4_node_step<Spec>(nodes4[cur], r, __builtin_fmaf(abs_(tmax), 2e-6f, tmax), cur, sp, tri_i, tri_end, stack);
        else
            walk4_tri_step<true>(tri_at, ctx, o, d, mint, maxt, any_hit, best, tmax, occluded, cur, sp, tri_i, tri_end, stack);
    }
    if (!any_hit && best.tri != MIW_MISS) best.prim = tri_at(best.tri).prim;          // (the triangle step does not carry it)
    return any_hit ? occluded : best.tri != MIW_MISS;
}
struct EmuCoin {                                                // xorshift32: the schedule of one test run
    uint32_t s;
    bool operator()() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (s & 1u) != 0; }
};
// The same for the 8-wide tree: walk8_node_step / walk8_tri_step of csrc/miw/bvh8.h, the bodies k_path_phased<NodeKind 8> runs. A lane
// is ready for exactly one body at a time (triangles before nodes), so there is no coin to toss inside one walk; the column
// (MIW_BVH8_STACK entries of 8 bytes) is checked on every access like the 4-wide one.
struct EmuColumn8 {
    U2 *p; int32_t cap; bool *bad; uint32_t *deepest;
    U2 &operator[](int32_t i) const {
        if (i < 0 || i >= cap) { *bad = true; return p[cap]; }
        if ((uint32_t) i + 1u > *deepest) *deepest = (uint32_t) i + 1u;
        return p[i];
    }
};
template <bool Spec = false, typename TriAt>
bool emu_walk8(const Bvh8Node *nodes8, TriAt tri_at, const PrimCtx &ctx, V3 o, V3 d, float mint, float maxt, bool any_hit, Hit &best,
               int32_t cap, bool *bad, uint32_t *deepest, uint64_t *steps, EmuCoin *coin = nullptr) {
    const SlabRay r = slab_ray_host(o, d, mint);
    U2 column[MIW_BVH8_STACK + 1];
    const EmuColumn8 stack{ column, cap, bad, deepest };
    Walk8 w; walk8_begin(w, r);
    float tmax = maxt;
    bool occluded = false;
    if (!any_hit) { best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu; }
    while (!walk8_over(w)) {
        // (Spec: a lane may be ready for both bodies — the coin decides, so every interleaving the wave votes can produce is reachable)
        if (walk8_node_ready<Spec>(w) && (!walk8_tri_ready(w) || !coin || (*coin)())) {
            const Bvh8Node &n = nodes8[walk8_next_node(w)];
            walk8_node_step<Spec>(n, r, __builtin_fmaf(abs_(tmax), 2e-6f, tmax), w, stack);
            if (steps) steps[0]++;
        } else if (walk8_tri_ready(w)) {
            walk8_tri_step<true, Spec>(tri_at, ctx, o, d, mint, maxt, any_hit, best, tmax, occluded, w);
            if (steps) steps[2]++;
        } else { *bad = true; break; }                          // neither over nor ready: a state the bodies must not produce
    }
    if (!any_hit && best.tri != MIW_MISS) best.prim = tri_at(best.tri).prim;
    return any_hit ? occluded : best.tri != MIW_MISS;
}
}

extern "C" {

// The level-by-level restatement of the binned-SAH builder (csrc/sah_levels.h: what the device builder runs) against the recursive
// host builder (csrc/bvh_build.h) on the scene's triangles. stats6 = inner nodes (recursive), inner nodes (levels), node records that
// differ (must be 0), depth (recursive), depth (levels), leaf ranges whose triangle SETS differ (must be 0); returns 1 when the level
// sweep asks for the host builder (need_host), 0 otherwise.
int emu_sah_levels_check(const mi_scene_desc *scene, int max_leaf, uint32_t *stats6) {
    EmuScene sc; if (!emu_build(scene, sc, max_leaf)) return -1;
    const SahLevelsResult lv = sah_build_levels_host(sc.tris_in, -1.f, (uint32_t) max_leaf);
    const BvhBuildResult &ref = sc.bvh;
    uint32_t diff = 0, leaf_diff = 0;
    if (!lv.need_host) {
        const size_t m = std::min(ref.nodes.size(), lv.nodes.size());
        for (size_t i = 0; i < m; ++i) diff += std::memcmp(&ref.nodes[i], &lv.nodes[i], sizeof(BvhNode)) != 0;
        diff += (uint32_t) (std::max(ref.nodes.size(), lv.nodes.size()) - m);
        for (size_t i = 0; i < m; ++i) {
            const int32_t ch[2] = { ref.nodes[i].child0, ref.nodes[i].child1 };
            for (int k = 0; k < 2; ++k) {
                if (ch[k] >= 0) continue;
                const uint32_t code = (uint32_t) ~ch[k], first = code >> 4, count = (code & 15u) + 1u;
                std::vector<uint32_t> a(ref.order.begin() + first, ref.order.begin() + first + count), b(lv.order.begin() + first, lv.order.begin() + first + count);
                std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
                leaf_diff += a != b;
            }
        }
        // the heights the 4-wide collapse takes from the builder == the ones it computes itself from the nodes
        const std::vector<uint32_t> h2 = bvh2_heights(lv.nodes);
        for (size_t i = 0; i < lv.nodes.size(); ++i) diff += h2[i] != lv.height[i];
    }
    if (stats6) { stats6[0] = (uint32_t) ref.nodes.size(); stats6[1] = (uint32_t) lv.nodes.size(); stats6[2] = diff; stats6[3] = ref.depth; stats6[4] = lv.depth; stats6[5] = leaf_diff; }
    return lv.need_host ? 1 : 0;
}

// stackless BVH (host SAH build, `max_leaf` triangles per leaf) over caller rays
int emu_trace(const mi_scene_desc *scene, const mi_rays_soa *r, const mi_hits_soa *h, uint64_t n, int any_hit, int max_leaf,
              uint32_t *stats3 /* nodes, tris, depth */) {
    EmuScene sc; if (!emu_build(scene, sc, max_leaf)) return -1;
    Ftz ftz;
    if (stats3) { stats3[0] = sc.view.node_count; stats3[1] = sc.view.tri_count; stats3[2] = sc.bvh.depth; }
    const BvhNode *nodes = sc.view.nodes; const Tri *tris = sc.view.tris; const PrimCtx rects = prim_ctx(sc.view);
    auto node_at = [nodes](int32_t i) -> const BvhNode & { return nodes[i]; };
    auto tri_at = [tris](uint32_t i) -> const Tri & { return tris[i]; };
    for (uint64_t i = 0; i < n; ++i) {
        RayPrep rp = ray_prepare(v3(r->ox[i], r->oy[i], r->oz[i]), v3(r->dx[i], r->dy[i], r->dz[i]), r->mint[i], r->maxt[i]);
        Hit hit; bool ok;
        if (any_hit) ok = bvh_intersect<true>(node_at, tri_at, rp, hit, rects); else ok = bvh_intersect<false>(node_at, tri_at, rp, hit, rects);
        h->t[i] = ok ? hit.t : MIW_INFINITY;
        if (h->u) h->u[i] = hit.u;
        if (h->v) h->v[i] = hit.v;
        if (h->prim) h->prim[i] = ok ? hit.prim : 0xffffffffu;
        if (h->shape) h->shape[i] = ok ? tris[hit.tri].shape : 0xffffffffu;
    }
    return 0;
}

// the 4-wide quantised tree of miw/bvh4.h (collapsed from the same SAH build) over caller rays.
// stats6: BVH2 nodes, BVH4 nodes, BVH4 depth, exact stack bound, deepest stack any of the rays reached, ok flag
// schedule: 0 = bvh4_intersect (the reference walk of miw/bvh4.h); > 0 = the phase machine's bodies (emu_walk4) with the
// schedule seeded by this number, bit 31 clear = Spec (the device default), set = the non-speculating variant; stats6[4] is then
// the deepest column slot any ray touched, and a touch outside the column (stack_budget + 1 slots, as on the device) returns 2.
int emu_trace4(const mi_scene_desc *scene, const mi_rays_soa *r, const mi_hits_soa *h, uint64_t n, int any_hit, int max_leaf,
               int stack_budget, int max_fan, uint32_t *stats6, uint32_t schedule) {
    EmuScene sc; if (!emu_build(scene, sc, max_leaf)) return -1;
    Ftz ftz;
    if (stack_budget < 1 || stack_budget > 127) return -3;                  // bvh4_intersect's host stack holds 128 entries
    const Bvh4BuildResult b4 = bvh4_collapse(sc.bvh.nodes, (uint32_t) stack_budget, max_fan);
    uint32_t seen = 0;
    if (stats6) { stats6[0] = sc.view.node_count; stats6[1] = (uint32_t) b4.nodes.size(); stats6[2] = b4.depth; stats6[3] = b4.stack_bound; stats6[4] = 0; stats6[5] = b4.ok ? 1u : 0u; }
    if (!b4.ok) return 1;
    const Tri *tris = sc.view.tris; const PrimCtx rects = prim_ctx(sc.view);
    auto tri_at = [tris](uint32_t i) -> const Tri & { return tris[i]; };
    EmuCoin coin{ schedule | 1u }; bool bad = false;
    for (uint64_t i = 0; i < n; ++i) {
        const V3 o = v3(r->ox[i], r->oy[i], r->oz[i]), d = v3(r->dx[i], r->dy[i], r->dz[i]);
        Hit hit; bool ok;
        if (schedule) {
            const bool spec = !(schedule & 0x80000000u);
            hit.t = MIW_INFINITY; hit.u = hit.v = 0.f; hit.tri = MIW_MISS; hit.prim = 0xffffffffu;
            ok = spec ? emu_walk4<true>(b4.nodes.data(), tri_at, rects, o, d, r->mint[i], r->maxt[i], any_hit != 0, hit, std::ref(coin), stack_budget + 1, &bad, &seen)
                      : emu_walk4<false>(b4.nodes.data(), tri_at, rects, o, d, r->mint[i], r->maxt[i], any_hit != 0, hit, std::ref(coin), stack_budget + 1, &bad, &seen);
            if (any_hit && ok) { hit.t = 0.f; hit.tri = 0; hit.prim = 0; }           // bvh4_intersect<true>'s convention (which triangle: not part of the contract)
        }
        else if (any_hit) ok = bvh4_intersect<true>(b4.nodes.data(), tri_at, o, d, r->mint[i], r->maxt[i], hit, rects, &seen);
        else ok = bvh4_intersect<false>(b4.nodes.data(), tri_at, o, d, r->mint[i], r->maxt[i], hit, rects, &seen);
        h->t[i] = ok ? hit.t : MIW_INFINITY;
        if (h->u) h->u[i] = hit.u;
        if (h->v) h->v[i] = hit.v;
        if (h->prim) h->prim[i] = ok ? hit.prim : 0xffffffffu;
        if (h->shape) h->shape[i] = ok ? tris[hit.tri].shape : 0xffffffffu;
    }
    if (stats6) stats6[4] = seen;
    return bad ? 2 : 0;
}

// the 8-wide quantised tree of miw/bvh8.h (collapsed from the same SAH build; triangles in the tree's own order) over caller rays.
// stats8: BVH2 nodes, BVH8 nodes, BVH8 depth, deepest stack any ray reached, ok flag, then as 64-bit pairs (lo, hi): node steps,
// triangles handed to the triangle test, triangle-pair steps (schedule > 0 only).
// schedule: 0 = bvh8_intersect (the reference walk); > 0 = the phase machine's bodies (emu_walk8), every column access checked (2 = a stray
// access); bit 31 set = the speculating variant (MIW_W8_SPEC: a second pending triangle group) under a pseudo-random body schedule.
// compare4 != 0: the 4-wide reference walk runs over the same rays as well and its node / triangle counts go to stats8[11..14].
int emu_trace8(const mi_scene_desc *scene, const mi_rays_soa *r, const mi_hits_soa *h, uint64_t n, int any_hit, int max_leaf,
               int max_fan, uint32_t *stats8, uint32_t schedule, int compare4) {
    EmuScene sc; if (!emu_build(scene, sc, max_leaf)) return -1;
    Ftz ftz;
    const Bvh8BuildResult b8 = bvh8_collapse(sc.bvh.nodes, sc.view.tri_count, max_fan);
    uint32_t seen = 0; uint64_t steps[3] = { 0, 0, 0 }, steps4[2] = { 0, 0 };
    if (stats8) { std::memset(stats8, 0, 15 * sizeof(uint32_t)); stats8[0] = sc.view.node_count; stats8[1] = (uint32_t) b8.nodes.size(); stats8[2] = b8.depth; stats8[4] = b8.ok ? 1u : 0u; }
    if (!b8.ok) return 1;
    std::vector<Tri> tris8(b8.perm.size());
    for (size_t i = 0; i < tris8.size(); ++i) tris8[i] = sc.view.tris[b8.perm[i]];
    const Tri *tris = tris8.data(); const PrimCtx rects = prim_ctx(sc.view);
    auto tri_at = [tris](uint32_t i) -> const Tri & { return tris[i]; };
    Bvh4BuildResult b4; const Tri *tris2 = sc.view.tris;
    auto tri_at2 = [tris2](uint32_t i) -> const Tri & { return tris2[i]; };
    if (compare4) b4 = bvh4_collapse(sc.bvh.nodes, 64u, 4);
    bool bad = false;
    const int32_t cap8 = (int32_t) std::max<uint32_t>(b8.depth, 2u);      // the column mi_render gives the walk: one entry per level of the tree
    for (uint64_t i = 0; i < n; ++i) {
        const V3 o = v3(r->ox[i], r->oy[i], r->oz[i]), d = v3(r->dx[i], r->dy[i], r->dz[i]);
        Hit hit; bool ok;
        if (schedule) {
            hit.t = MIW_INFINITY; hit.u = hit.v = 0.f; hit.tri = MIW_MISS; hit.prim = 0xffffffffu;
            EmuCoin coin{ schedule | 1u };
            ok = (schedule & 0x80000000u) ? emu_walk8<true>(b8.nodes.data(), tri_at, rects, o, d, r->mint[i], r->maxt[i], any_hit != 0, hit, cap8, &bad, &seen, steps, &coin)
                                          : emu_walk8<false>(b8.nodes.data(), tri_at, rects, o, d, r->mint[i], r->maxt[i], any_hit != 0, hit, cap8, &bad, &seen, steps);
            if (any_hit && ok) { hit.t = 0.f; hit.tri = 0; hit.prim = 0; }
        }
        else if (any_hit) ok = bvh8_intersect<true>(b8.nodes.data(), tri_at, o, d, r->mint[i], r->maxt[i], hit, rects, &seen, steps);
        else ok = bvh8_intersect<false>(b8.nodes.data(), tri_at, o, d, r->mint[i], r->maxt[i], hit, rects, &seen, steps);
        if (compare4 && b4.ok) {
            Hit h4;
            if (any_hit) bvh4_intersect<true>(b4.nodes.data(), tri_at2, o, d, r->mint[i], r->maxt[i], h4, rects, nullptr, steps4);
            else bvh4_intersect<false>(b4.nodes.data(), tri_at2, o, d, r->mint[i], r->maxt[i], h4, rects, nullptr, steps4);
        }
        h->t[i] = ok ? hit.t : MIW_INFINITY;
        if (h->u) h->u[i] = hit.u;
        if (h->v) h->v[i] = hit.v;
        if (h->prim) h->prim[i] = ok ? hit.prim : 0xffffffffu;
        if (h->shape) h->shape[i] = ok ? tris[hit.tri].shape : 0xffffffffu;
    }
    if (stats8) {
        stats8[3] = seen;
        for (int k = 0; k < 3; ++k) { stats8[5 + 2 * k] = (uint32_t) steps[k]; stats8[6 + 2 * k] = (uint32_t) (steps[k] >> 32); }
        for (int k = 0; k < 2; ++k) { stats8[11 + 2 * k] = (uint32_t) steps4[k]; stats8[12 + 2 * k] = (uint32_t) (steps4[k] >> 32); }
    }
    return bad ? 2 : 0;
}

// Phase classes (csrc/miw/film.h, csrc/film_classes.h) against ImageBlock::put itself: for `n` samples (position x, y and the
// film coordinates of the pixel each one belongs to) the per-texel weights of the 8 x 8 window around the pixel's texel
// (window texel (a, b) = block texel (t_x - reach + a, t_y - reach + b)), once as the 16-byte record + class tables give
// them (what LogSink16 / film_block_replay16 / k_film_groups use) and once from block_splat() with the value 1 in the
// weight channel. out_*: n x 64 floats. Returns the class count, or -1 when the filter has no class tables.
int emu_film_weights(const mi_render_cfg *cfg, int n, const float *pos_xy, const int32_t *pixel_xy, float *out_classes, float *out_direct, int32_t *out_reach) {
    FilmRec F; std::memset(&F, 0, sizeof F);
    F.crop_w = cfg->crop_w; F.crop_h = cfg->crop_h; F.crop_x = cfg->crop_x; F.crop_y = cfg->crop_y;
    F.block_size = cfg->block_size; F.border = cfg->filter_border; F.radius = cfg->filter_radius;
    F.scale_factor = (float) MIW_FILTER_RESOLUTION / cfg->filter_radius;
    std::memcpy(F.lut, cfg->filter_lut, sizeof F.lut);
    F.warn_negative = 1u;
    const FilmClasses C = film_classes_build(F);
    if (!C.ok) return -1;
    if (out_reach) *out_reach = C.reach;
    Ftz ftz;
    for (int s = 0; s < n; ++s) {
        const int px = pixel_xy[2 * s], py = pixel_xy[2 * s + 1];
        const V2 pos = v2(pos_xy[2 * s], pos_xy[2 * s + 1]);
        float *oc = out_classes + (size_t) s * 64, *od = out_direct + (size_t) s * 64;
        for (int i = 0; i < 64; ++i) oc[i] = od[i] = 0.f;
        const uint32_t cx = film_class_of(C.thr.data(), film_phase(F, pos.x, px, F.crop_x)),
                       cy = film_class_of(C.thr.data(), film_phase(F, pos.y, py, F.crop_y));
        int bx, by, bw, bh;
        block_of_pixel(F, px, py, bx, by, bw, bh);
        const int size_x = bw + 2 * F.border, size_y = bh + 2 * F.border;
        const int ptx = px - F.crop_x - bx + F.border, pty = py - F.crop_y - by + F.border;
        for (int b = 0; b < 8; ++b)
            for (int a = 0; a < 8; ++a) {
                const int tx = ptx - C.reach + a, ty = pty - C.reach + b;
                if (tx < 0 || ty < 0 || tx >= size_x || ty >= size_y) continue;          // a texel no lane of the replay owns
                oc[b * 8 + a] = C.w[cy * MIW_FC_STRIDE + b] * C.w[cx * MIW_FC_STRIDE + a];
            }
        const float value[5] = { 0.f, 0.f, 0.f, 0.f, 1.f };
        block_splat(F, bx + F.crop_x, by + F.crop_y, bw, bh, pos, value, [&](int texel, int k, float term) {
            if (k != 4) return;
            const int tx = texel % size_x, ty = texel / size_x, a = tx - (ptx - C.reach), b = ty - (pty - C.reach);
            if (a < 0 || b < 0 || a >= 8 || b >= 8) { od[63] = -1.f; return; }            // outside the window: the test fails on it
            od[b * 8 + a] += term;
        });
    }
    return (int) C.count;
}

// the wavefront render loop of mi_render, stage by stage, on the CPU.
// film64: crop_w*crop_h*5 doubles (film_mode 2: immediate splat, exact sum);
// film32 (may be NULL): crop_w*crop_h*5 floats (film_mode 1: sample log + ordered gather).
int emu_render(const mi_scene_desc *scene, const mi_render_cfg *cfg, double *film64, float *film32,
               uint64_t *stats4 /* samples, segments, shadow, iterations */) {
    EmuScene sc; if (!emu_build(scene, sc, 4)) return -1;
    Ftz ftz;
    RenderParams P; std::memset(&P, 0, sizeof P);
    std::memcpy(P.sensor.sample_to_camera, cfg->sample_to_camera, 64);
    std::memcpy(P.sensor.to_world, cfg->to_world, 64);
    P.sensor.near_clip = cfg->near_clip; P.sensor.far_clip = cfg->far_clip;
    P.sensor.pp_offset[0] = cfg->principal_point_offset[0]; P.sensor.pp_offset[1] = cfg->principal_point_offset[1];
    P.film.crop_w = cfg->crop_w; P.film.crop_h = cfg->crop_h; P.film.crop_x = cfg->crop_x; P.film.crop_y = cfg->crop_y;
    P.film.block_size = cfg->block_size; P.film.border = cfg->filter_border; P.film.radius = cfg->filter_radius;
    P.film.scale_factor = (float) MIW_FILTER_RESOLUTION / cfg->filter_radius;
    std::memcpy(P.film.lut, cfg->filter_lut, sizeof P.film.lut);
    P.film.warn_negative = cfg->moment_pass ? 0u : 1u;
    P.spp = cfg->spp; P.max_depth = cfg->max_depth; P.rr_depth = cfg->rr_depth;
    P.moment_pass = (uint32_t) cfg->moment_pass;
    render_params_prepare(P);
    const bool direct = cfg->integrator == MI_INTEGRATOR_DIRECT;
    if (direct) {                                              // fill_params (miwave.hip), direct.cpp:82-103
        const uint32_t ne = cfg->emitter_samples, nb = cfg->bsdf_samples;
        if (ne + nb == 0 || cfg->plan == 1) return -3;
        P.integrator = INTEG_DIRECT;
        P.direct.emitter_samples = ne; P.direct.bsdf_samples = nb; P.direct.hide_emitters = cfg->hide_emitters ? 1u : 0u;
        P.direct.weight_bsdf = 1.f / (float) nb; P.direct.weight_lum = 1.f / (float) ne;
        P.direct.frac_bsdf = (float) nb / (float) (ne + nb); P.direct.frac_lum = (float) ne / (float) (ne + nb);
    }

    const uint32_t bs = (uint32_t) cfg->block_size, bs2 = bs * bs;
    const uint32_t blocks_x = (cfg->crop_w + bs - 1) / bs;
    uint32_t n_tiles = cfg->tile_list ? cfg->tile_count : cfg->block_count;
    uint32_t n_lanes = n_tiles * bs2;
    P.n_lanes = n_lanes;
    std::vector<F4> tp(n_lanes), res(n_lanes), ray_o(n_lanes), ray_d(n_lanes), hit(n_lanes), sh_d(n_lanes), sh_c(n_lanes);
    std::vector<U4> st(n_lanes); std::vector<F2> pos(n_lanes); std::vector<uint32_t> pixel(n_lanes), sh_vis(n_lanes);
    LaneQueues Q; Q.tp = tp.data(); Q.res = res.data(); Q.st = st.data(); Q.pos = pos.data(); Q.pixel = pixel.data();
    Q.ray_o = ray_o.data(); Q.ray_d = ray_d.data(); Q.hit = hit.data(); Q.sh_d = sh_d.data(); Q.sh_c = sh_c.data(); Q.sh_vis = sh_vis.data();
    // the sample log, in the format the device would pick for this filter (miwave.hip: mi_render): 16-byte records with phase
    // classes where film_classes_build covers the filter, positions + values (24 bytes) otherwise
    std::vector<F2> log_pos; std::vector<F4> log_val; std::vector<U4> log_rec;
    const FilmClasses classes = film_classes_build(P.film);
    const bool rec16 = classes.ok && !getenv("MIW_FILM_LEGACY");
    // MIW_EMU_LOG_IL=1: the tile-interleaved log (miw/film.h: log_index) the device writes for k_film_lanes — same film
    const uint32_t log_il = rec16 && getenv("MIW_EMU_LOG_IL") && atoi(getenv("MIW_EMU_LOG_IL")) ? [&] { uint32_t l = 0; while ((1u << l) < bs2) ++l; return l + 1u; }() : 0u;
    if (film32) {
        if (rec16) log_rec.resize(log_capacity(log_il, n_tiles, bs2, cfg->spp));
        else { log_pos.resize((size_t) n_lanes * cfg->spp); log_val.resize((size_t) n_lanes * cfg->spp); }
    }
    Q.log_pos = log_pos.data(); Q.log_val = log_val.data(); Q.log_rec = rec16 ? log_rec.data() : nullptr; Q.log_thr = classes.thr.data(); Q.log_rej = classes.count;
    size_t film_n = (size_t) cfg->crop_w * cfg->crop_h * 5;
    if (!cfg->accumulate) std::memset(film64, 0, film_n * sizeof(double));

    for (uint32_t lane = 0; lane < n_lanes; ++lane) {          // k_init_lanes
        uint32_t tile = lane / bs2, i = lane % bs2;
        uint32_t b = cfg->tile_list ? cfg->tile_list[tile] : tile;
        uint32_t bx = b % blocks_x, by = b / blocks_x, x, y;
        morton_decode2(i, x, y);
        int bw = std::min<int>(bs, cfg->crop_w - (int) (bx * bs)), bh = std::min<int>(bs, cfg->crop_h - (int) (by * bs));
#if MIW_SPECTRAL   // spectral builds run the resident plan only: a lane is its st word (k_init_pixels)
        if ((int) x >= bw || (int) y >= bh) { pixel[lane] = 0; U4 d; d.x = d.y = d.w = 0; d.z = LF_DONE; st[lane] = d; continue; }
        uint32_t px = (uint32_t) cfg->crop_x + bx * bs + x, py = (uint32_t) cfg->crop_y + by * bs + y;
        pixel[lane] = px | (py << 16);
        st[lane] = lane_seed_state(cfg->base_seed + (uint64_t) cfg->block_ids[b] * bs2 + i);
#else
        if ((int) x >= bw || (int) y >= bh) { pixel[lane] = 0; lane_init_unused(Q, lane); continue; }
        uint32_t px = (uint32_t) cfg->crop_x + bx * bs + x, py = (uint32_t) cfg->crop_y + by * bs + y;
        pixel[lane] = px | (py << 16);
        lane_init(P, Q, lane, pixel[lane], cfg->base_seed + (uint64_t) cfg->block_ids[b] * bs2 + i);
#endif
    }
    const BvhNode *nodes = sc.view.nodes; const Tri *tris = sc.view.tris; const PrimCtx rects = prim_ctx(sc.view);
    auto node_at = [nodes](int32_t i) -> const BvhNode & { return nodes[i]; };
    auto tri_at = [tris](uint32_t i) -> const Tri & { return tris[i]; };
    auto add = [film64](int texel, int k, float v) { film64[(size_t) texel * 5 + k] += (double) v; };
    Counters cnt; std::memset(&cnt, 0, sizeof cnt);
    uint64_t iterations = 0;
    if (cfg->plan == 2 || direct) {
        // the resident plan (k_init_pixels + k_path_resident): every pixel advanced
        // `samples_per_launch` samples per pass, state between passes = st word only
        for (uint32_t lane = 0; lane < n_lanes; ++lane) {
            if (st[lane].z & LF_DONE) { st[lane].x = st[lane].y = st[lane].w = 0; continue; }
            uint32_t tile = lane / bs2, i = lane % bs2;
            uint32_t b = cfg->tile_list ? cfg->tile_list[tile] : tile;
            st[lane] = lane_seed_state(cfg->base_seed + (uint64_t) cfg->block_ids[b] * bs2 + i);
        }
        const uint32_t per_launch = cfg->samples_per_launch > 0 ? (uint32_t) cfg->samples_per_launch : 128u;
        // Scenes the device walks with k_path_phased (more triangles than the packet kernels take, a 4-wide collapse that fits the
        // lane stack) go through the phase machine's own per-lane bodies here too (emu_walk4: an E walk, then the S walk that leaves
        // `best` alone, each under a pseudo-random body schedule); MIW_EMU_WALK=bvh2 keeps the stackless BVH2 walk for all scenes.
        // Since round 5 the device's default tree is the 8-wide one (miw/bvh8.h: walk8_node_step / walk8_tri_step, triangles and vertex
        // normals in that tree's order): emu_walk8 over a view with the permuted arrays; MIW_EMU_WALK=bvh4 keeps the 4-wide bodies.
        Bvh4BuildResult b4; Bvh8BuildResult b8;
        const char *walk_env = getenv("MIW_EMU_WALK");
        const bool want_tree = sc.view.tri_count > 64u && !(walk_env && !strcmp(walk_env, "bvh2"));
        if (want_tree && !(walk_env && !strcmp(walk_env, "bvh4"))) b8 = bvh8_collapse(sc.bvh.nodes, sc.view.tri_count);
        const bool phased8 = b8.ok && !b8.nodes.empty();
        if (want_tree && !phased8) b4 = bvh4_collapse(sc.bvh.nodes, 31u, 4);
        const bool phased = b4.ok && !b4.nodes.empty();
        std::vector<Tri> tris8; std::vector<float> vn8;
        SceneView view8 = sc.view;
        if (phased8) {
            tris8.resize(b8.perm.size());
            for (size_t i = 0; i < tris8.size(); ++i) tris8[i] = sc.view.tris[b8.perm[i]];
            view8.tris = tris8.data();
            if (sc.view.tri_vn) {
                vn8.resize(tris8.size() * 9);
                for (size_t i = 0; i < tris8.size(); ++i) std::memcpy(&vn8[i * 9], sc.view.tri_vn + (size_t) b8.perm[i] * 9, 36);
                view8.tri_vn = vn8.data();
            }
        }
        const SceneView &V = phased8 ? view8 : sc.view;
        const Tri *tris_w = V.tris;
        auto tri_at_w = [tris_w](uint32_t i) -> const Tri & { return tris_w[i]; };
        EmuCoin coin{ 0x9e3779b9u }; bool bad_slot = false; uint32_t deepest = 0;
        auto trace2 = [&](V3 o, float mint, V3 dE, float maxtE, bool hasE, V3 dS, float maxtS, bool hasS, F4 &hE, bool &occS) {
            Hit h; h.t = MIW_INFINITY; h.u = h.v = 0.f; h.tri = MIW_MISS; h.prim = 0xffffffffu;
            occS = false;
            if (phased8) {
                // (the speculating bodies under a pseudo-random schedule — the device default, MIW_W8_SPEC — in a column of exactly `depth`
                // entries, as mi_render sizes it)
                if (hasE) emu_walk8<true>(b8.nodes.data(), tri_at_w, rects, o, dE, mint, maxtE, false, h, (int32_t) std::max<uint32_t>(b8.depth, 2u), &bad_slot, &deepest, nullptr, &coin);
                if (hasS) occS = emu_walk8<true>(b8.nodes.data(), tri_at_w, rects, o, dS, mint, maxtS, true, h, (int32_t) std::max<uint32_t>(b8.depth, 2u), &bad_slot, &deepest, nullptr, &coin);
            } else if (phased) {
                if (hasE) emu_walk4<true>(b4.nodes.data(), tri_at, rects, o, dE, mint, maxtE, false, h, std::ref(coin), 32, &bad_slot, &deepest);
                if (hasS) occS = emu_walk4<true>(b4.nodes.data(), tri_at, rects, o, dS, mint, maxtS, true, h, std::ref(coin), 32, &bad_slot, &deepest);
            } else {
                if (hasE) { RayPrep rp = ray_prepare(o, dE, mint, maxtE); bvh_intersect<false>(node_at, tri_at, rp, h, rects); }
                if (hasS) { Hit hs; RayPrep rp = ray_prepare(o, dS, mint, maxtS); occS = bvh_intersect<true>(node_at, tri_at, rp, hs, rects); }
            }
            hE.x = h.t; hE.y = h.u; hE.z = h.v; hE.w = u2f(h.tri);
        };
        for (uint32_t done = 0; done < cfg->spp; ) {
            const uint32_t end = cfg->spp - done < per_launch ? cfg->spp : done + per_launch;
            for (uint32_t lane = 0; lane < n_lanes; ++lane) {
                if (st[lane].z & LF_DONE) continue;
                SplatSink<decltype(add)> splat{ &P.film, add };
                LogSink log{ Q.log_pos, Q.log_val, lane, cfg->spp, P.film.warn_negative };
                LogSink16<const float *> log16{ Q.log_rec, Q.log_thr, &P.film, lane, cfg->spp, Q.log_rej, log_il };
                bool do_log = film32 != nullptr;
                auto sink = [&](uint32_t pixel, uint32_t sample_idx, V2 pos, const float *aovs) {
                    splat(pixel, sample_idx, pos, aovs);
                    if (do_log) { if (rec16) log16(pixel, sample_idx, pos, aovs); else log(pixel, sample_idx, pos, aovs); }
                };
                st[lane] = direct ? pixel_render<INTEG_DIRECT>(P, V, pixel[lane], st[lane], end, trace2, sink, &cnt)
                                  : pixel_render(P, V, pixel[lane], st[lane], end, trace2, sink, &cnt);
            }
            done = end; ++iterations;
        }
        if (bad_slot) return -4;                               // a walk touched a slot outside its 32-entry column
    } else
#if MIW_SPECTRAL
    return -2;                                                 // the HBM-queue plan carries RGB path state
#else
    for (;;) {
        for (uint32_t lane = 0; lane < n_lanes; ++lane) {      // k_trace<any>
            F4 d = sh_d[lane]; if (d.w < 0.f) continue;
            F4 o = ray_o[lane]; Hit h;
            RayPrep rp = ray_prepare(v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w, d.w);
            sh_vis[lane] = bvh_intersect<true>(node_at, tri_at, rp, h, rects) ? 0u : 1u;
        }
        for (uint32_t lane = 0; lane < n_lanes; ++lane) {      // k_trace<closest>
            F4 d = ray_d[lane]; if (d.w < 0.f) continue;
            F4 o = ray_o[lane]; Hit h;
            RayPrep rp = ray_prepare(v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w, d.w);
            bvh_intersect<false>(node_at, tri_at, rp, h, rects);
            F4 r; r.x = h.t; r.y = h.u; r.z = h.v; r.w = u2f(h.tri); hit[lane] = r;
        }
        uint64_t active = 0;
        for (uint32_t lane = 0; lane < n_lanes; ++lane) {      // k_shade (both film modes at once)
            SplatSink<decltype(add)> splat{ &P.film, add };
            LogSink log{ Q.log_pos, Q.log_val, lane, cfg->spp, P.film.warn_negative };
            LogSink16<const float *> log16{ Q.log_rec, Q.log_thr, &P.film, lane, cfg->spp, Q.log_rej, log_il };
            bool do_log = film32 != nullptr;
            auto sink = [&](uint32_t pixel, uint32_t sample_idx, V2 pos, const float *aovs) {
                splat(pixel, sample_idx, pos, aovs);
                if (do_log) { if (rec16) log16(pixel, sample_idx, pos, aovs); else log(pixel, sample_idx, pos, aovs); }
            };
            active += (lane_shade(P, sc.view, Q, lane, &cnt, sink) & LF_DONE) ? 0 : 1;
        }
        ++iterations;
        if (active == 0) break;
    }
#endif
    if (film32) {                                              // k_film_blocks + k_film_merge
        std::vector<int32_t> block_tile(cfg->block_count, -1);
        for (uint32_t t = 0; t < n_tiles; ++t) block_tile[cfg->tile_list ? cfg->tile_list[t] : t] = (int32_t) t;
        BlockReplayArgs A; A.log_pos = log_pos.data(); A.log_val = log_val.data(); A.st = st.data(); A.spp = cfg->spp;
        A.log_rec = rec16 ? log_rec.data() : nullptr; A.log_il = log_il; A.cls = classes.view();
        A.block_ids = cfg->block_ids; A.block_tile = block_tile.data(); A.tile_list = cfg->tile_list;
        A.blocks_x = blocks_x; A.blocks_y = (cfg->crop_h + bs - 1) / bs;
        uint32_t l2 = 0; while ((1u << l2) < bs2) ++l2;
        A.bs2_log2 = l2;
        uint32_t side = bs + 2u * (uint32_t) cfg->filter_border;
        A.tile_stride = side * side * MIW_FILM_CHANNELS;
        std::vector<float> tiles((size_t) n_tiles * A.tile_stride, 0.f);
        for (uint32_t t = 0; t < n_tiles; ++t) {
            if (rec16) film_block_replay16(P.film, A, t, tiles.data() + (size_t) t * A.tile_stride);       // k_film_groups
            else film_block_replay(P.film, A, t, tiles.data() + (size_t) t * A.tile_stride);               // k_film_blocks
        }
        for (int fy = 0; fy < cfg->crop_h; ++fy)
            for (int fx = 0; fx < cfg->crop_w; ++fx)
                film_merge_texel(P.film, A, tiles.data(), fx, fy, film32 + ((size_t) fy * cfg->crop_w + fx) * 5, cfg->accumulate != 0);
    }
    if (stats4) { stats4[0] = cnt.samples; stats4[1] = cnt.segments; stats4[2] = cnt.shadow_rays; stats4[3] = iterations; }
    return 0;
}

} // extern "C"
