// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked, imported or executed by the
// product (libmiwave.so / libmiwave_host.so); only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may call it.
//
// A CPU restatement of Mitsuba 2's `scalar_rgb` path-tracing hot path: the
// control flow below follows the reference files line by line (citations on
// every function); the float32 leaf arithmetic (vector ops, PCG32, warps,
// triangle test, BSDFs, emitter sampling, filter splat) comes from the product's
// shared `miw/*.h` headers BY DESIGN: bit-parity between
// x86-64 and gfx950 is then a property of the two compilers, and those leaf
// functions are pinned by the reference's own known-answer tests
// (tests/test_oracle_kat.py: TEA, PCG32 demo vector, dielectric, Fresnel,
// microfacet, mesh intersection, ImageBlock::put, Spiral, diffuse).
//
// What is independent of the product: the render loop (render / render_block /
// render_sample), the scalar PathIntegrator::sample loop with its carried
// state, the spiral, brute-force Scene::ray_intersect / ray_test, bordered
// float32 ImageBlocks merged into the film — i.e. everything the wavefront
// decomposition re-orders.
//
// Parity status: PINNED against every known-answer vector the reference's tests
// hold for this path; the whole-image result is NOT pinned by the reference (its
// stored references are statistical and its data submodule is absent; the real
// binary cannot be built here: ext/enoki, ext/tbb ... are empty) — see DESIGN.md.
// The estimators as a whole (path, direct: also restated here, direct.cpp:105-198; the moment wrapper) are checked
// against closed forms that share no code with them (tests/test_furnace.py) and by the statistical tier of the
// reference's own tests (tests/test_chi2.py, tests/test_moment.py).
//
// Denormals: preserved (plain IEEE) on both sides — see FtzScope below.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include <xmmintrin.h>
#include <pmmintrin.h>

#include "../include/miwave.h"
#include "../mitsuba2_amd/csrc/miw/base.h"
#include "../mitsuba2_amd/csrc/miw/rng.h"
#include "../mitsuba2_amd/csrc/miw/warp.h"
#include "../mitsuba2_amd/csrc/miw/special.h"
#include "../mitsuba2_amd/csrc/miw/shape.h"
#include "../mitsuba2_amd/csrc/miw/bsdf.h"
#include "../mitsuba2_amd/csrc/miw/scene.h"
#include "../mitsuba2_amd/csrc/miw/film.h"
#include "../mitsuba2_amd/csrc/envmap_build.h"
#include "../mitsuba2_amd/csrc/rect_build.h"
#include "../mitsuba2_amd/csrc/texture_build.h"
#include "../mitsuba2_amd/csrc/bvh_build.h"     // scene_pad_unit only: the oracle has no acceleration structure

using namespace miw;

namespace {

// Float environment of every oracle thread: IEEE-754 round-to-nearest with
// denormals PRESERVED (FTZ/DAZ forced off, whatever the caller's MXCSR says).
// The reference flushes denormals on the CPU (scoped_flush_denormals,
// integrator.cpp:117) as a speed measure; x86 FTZ/DAZ and gfx950's flush mode do
// not agree on denormal pass-through (min/max/select), so both sides of this
// code base run plain IEEE instead — results differ from a flushing build only
// for |x| < 1.2e-38.
struct FtzScope {
    unsigned csr;
    FtzScope() { csr = _mm_getcsr(); _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_OFF); _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_OFF); }
    ~FtzScope() { _mm_setcsr(csr); }
};

// ---- scene in scene order. Scene queries are brute force by definition; orc_set_accel(1) routes them through the
// checker's OWN spatial index below (OAccel), which exists only so that the configured sizes of BASELINE configs 3 / 4 (a
// 128x128 window at 1024 spp over 41 k triangles, 2048 spp over 0.9 M) finish in minutes on the host. It shares nothing with
// the product's builders or walks (csrc/bvh_build.h, miw/bvh.h, miw/bvh4.h): median splits, float64 slab arithmetic, and the
// SAME leaf test as the brute-force loop (prim_intersect, min t, ties to the smaller primitive id). It must return brute
// force's answer bit for bit: tests/test_oracle_accel.py (random, axis-parallel, in-plane and edge-grazing rays; films).
struct OAccelNode { double lo[3], hi[3]; uint32_t left, right, first, count; };   // count > 0: leaf over prims[first, first + count)
struct OAccel {
    std::vector<OAccelNode> nodes; std::vector<uint32_t> prims; std::vector<uint32_t> always;   // always: analytic primitives (tested on every query)
    bool on = false;
};
std::atomic<int> g_accel{0};

struct OScene {
    std::vector<Tri> tris;              // face order == global primitive id; pad = k + 1: the slot of analytic rectangle k
    std::vector<AnalyticRec> rects;         // analytic rectangles (src/shapes/rectangle.cpp)
    std::vector<float> tri_vn;          // 9 per face or empty
    std::vector<float> tri_uv;          // 6 per face (Mesh::vertex_texcoord of its three vertices) or empty
    std::vector<BitmapRec> bitmaps;     // bitmap textures (data = the caller's arrays)
    std::vector<ShapeRec> shapes;
    std::vector<BsdfRec> bsdfs;
    std::vector<EmitterRec> emitters;
    std::vector<float> emit_tri, emit_vnorm, emit_pmf, emit_cdf;
    EnvmapTables env;                   // environment emitter (scene.cpp:47-51) or !ok
    SceneView view{};
    OAccel accel;
};

// Why the index cannot lose a hit brute force counts: a triangle hit counts only if fl(o + t d) lies inside the triangle's
// vertex bounds grown by accept_pad (shape.h, the accept rule); the exact point o + t d is within a few ulps of the scene
// extent of that float point, i.e. well inside the bounds grown by 2 x accept_pad — the boxes used here. The slab interval
// of such a box, computed in float64 from the float inputs (relative error 1e-16, slack 1e-9 below), therefore contains t,
// and t also lies in [mint, min(maxt, best t so far)]: a node is skipped only when its interval misses that range.
// (-DMIW_ACCEPT_RULE=0 builds have no such bound: there the index is refused, orc_render falls back to brute force.)
void accel_build(OScene &o) {
    OAccel &A = o.accel;
    A.nodes.clear(); A.prims.clear(); A.always.clear(); A.on = false;
#if !MIW_ACCEPT_RULE
    return;
#endif
    const double pad = 2.0 * (double) o.view.accept_pad;
    const size_t n = o.tris.size();
    std::vector<double> blo(3 * n), bhi(3 * n), cen(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const Tri &t = o.tris[i];
        if (t.pad) { A.always.push_back((uint32_t) i); continue; }
        for (int a = 0; a < 3; ++a) {
            const double x0 = t.p0[a], x1 = t.p1[a], x2 = t.p2[a];
            blo[3 * i + a] = std::min(x0, std::min(x1, x2)) - pad; bhi[3 * i + a] = std::max(x0, std::max(x1, x2)) + pad;
            cen[3 * i + a] = (x0 + x1 + x2) / 3.0;
        }
        A.prims.push_back((uint32_t) i);
    }
    if (A.prims.empty()) { A.on = true; return; }
    struct Job { uint32_t node, first, count; };
    std::vector<Job> todo;
    A.nodes.push_back(OAccelNode{});
    todo.push_back({ 0u, 0u, (uint32_t) A.prims.size() });
    while (!todo.empty()) {
        const Job j = todo.back(); todo.pop_back();
        OAccelNode nd{};
        double clo[3] = { 1e300, 1e300, 1e300 }, chi[3] = { -1e300, -1e300, -1e300 };
        for (int a = 0; a < 3; ++a) { nd.lo[a] = 1e300; nd.hi[a] = -1e300; }
        for (uint32_t k = j.first; k < j.first + j.count; ++k) {
            const uint32_t i = A.prims[k];
            for (int a = 0; a < 3; ++a) {
                nd.lo[a] = std::min(nd.lo[a], blo[3 * i + a]); nd.hi[a] = std::max(nd.hi[a], bhi[3 * i + a]);
                clo[a] = std::min(clo[a], cen[3 * i + a]); chi[a] = std::max(chi[a], cen[3 * i + a]);
            }
        }
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
        if (j.count <= 4u || !(chi[axis] > clo[axis])) {         // few primitives, or all centroids coincide: a leaf
            nd.first = j.first; nd.count = j.count;
            A.nodes[j.node] = nd;
            continue;
        }
        const uint32_t half = j.count / 2u;
        std::nth_element(A.prims.begin() + j.first, A.prims.begin() + j.first + half, A.prims.begin() + j.first + j.count,
                         [&](uint32_t x, uint32_t y) { const double cx = cen[3 * x + axis], cy = cen[3 * y + axis]; return cx < cy || (cx == cy && x < y); });
        nd.count = 0; nd.left = (uint32_t) A.nodes.size(); nd.right = nd.left + 1u;
        A.nodes.push_back(OAccelNode{}); A.nodes.push_back(OAccelNode{});
        A.nodes[j.node] = nd;
        todo.push_back({ nd.left, j.first, half });
        todo.push_back({ nd.right, j.first + half, j.count - half });
    }
    A.on = true;
}
// slab interval of a box for the ray (float inputs, float64 arithmetic); false: the ray misses the box
inline bool accel_slab(const OAccelNode &nd, const double o[3], const double d[3], double &tn, double &tf) {
    tn = -1e300; tf = 1e300;
    for (int a = 0; a < 3; ++a) {
        if (d[a] == 0.0) { if (o[a] < nd.lo[a] || o[a] > nd.hi[a]) return false; continue; }
        double t0 = (nd.lo[a] - o[a]) / d[a], t1 = (nd.hi[a] - o[a]) / d[a];
        if (t0 > t1) std::swap(t0, t1);
        tn = std::max(tn, t0); tf = std::min(tf, t1);
    }
    return true;
}

bool build_scene(const mi_scene_desc *s, OScene &o) {
    o.tris.assign(s->face_count, Tri{});
    o.shapes.resize(s->shape_count);
    bool any_normals = false;
    for (uint32_t i = 0; i < s->shape_count; ++i) {
        const mi_shape &sh = s->shapes[i];
        int32_t emitter_id = sh.emitter;            // Scene::m_emitters order: the envmap sits at envmap->emitter_index
        if (emitter_id >= 0 && s->envmap && (uint32_t) emitter_id >= s->envmap->emitter_index) emitter_id += 1;
        o.shapes[i] = ShapeRec{ sh.bsdf, emitter_id, sh.flags & (SHAPE_HAS_NORMALS | SHAPE_HAS_TEXCOORDS), 0 };
        any_normals = any_normals || (sh.flags & 1u);
        for (uint32_t f = sh.first_face; f < sh.first_face + sh.face_count; ++f) o.tris[f].shape = i;
    }
    if (any_normals) o.tri_vn.assign((size_t) s->face_count * 9, 0.f);
    if (!build_face_texcoords(s, o.tri_uv)) return false;
    o.rects.clear();
    for (uint32_t k = 0; k < s->rectangle_count; ++k) {        // Rectangle(props) + update(), rectangle.cpp:76-96
        const mi_rectangle &q = s->rectangles[k];
        const uint32_t f = s->shapes[q.shape].first_face;
        o.rects.push_back(rect_record(q.to_world, q.to_object, q.shape, f));
        o.tris[f].pad = k + 1u; o.tris[f].prim = f;
    }
    for (uint32_t k = 0; k < s->sphere_count; ++k) {           // Sphere(props) + update(), sphere.cpp:96-131
        const mi_sphere &q = s->spheres[k];
        const uint32_t f = s->shapes[q.shape].first_face;
        o.rects.push_back(sphere_record(q.center, q.radius, q.flip_normals != 0, q.to_world, q.to_object, q.shape, f));
        o.tris[f].pad = (uint32_t) o.rects.size(); o.tris[f].prim = f;
    }
    for (uint32_t f = 0; f < s->face_count; ++f) {
        Tri &t = o.tris[f];
        if (t.pad) continue;                                  // a rectangle's primitive slot
        for (int k = 0; k < 3; ++k) {
            uint32_t vi = s->faces[3 * f + k];
            float *dst = k == 0 ? t.p0 : (k == 1 ? t.p1 : t.p2);
            std::memcpy(dst, s->vertex_positions + 3 * (size_t) vi, 12);
            if (any_normals && (o.shapes[t.shape].flags & 1u))
                std::memcpy(&o.tri_vn[(size_t) f * 9 + 3 * k], s->vertex_normals + 3 * (size_t) vi, 12);
        }
        t.prim = f; t.pad = 0;
    }
    o.bsdfs.resize(s->bsdf_count);
    for (uint32_t i = 0; i < s->bsdf_count; ++i) {
        int slot = 0;
        if (bsdf_record_from_abi(s->bsdfs[i], s->bitmap_count, s->bsdf_table_floats, o.bsdfs[i], &slot)) return false;
    }
    { uint32_t bad = 0; if (build_bitmap_table(s, o.bitmaps, &bad)) return false; }
    // Mesh::build_pmf (mesh.cpp:285-312) + DiscreteDistribution::update (distr_1d.h:55-87)
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
        if (sh.flags & (MI_SHAPE_RECTANGLE | MI_SHAPE_SPHERE)) {   // analytic shape: sampled by its own routines, no tables
            const uint32_t k = o.tris[sh.first_face].pad - 1u;
            r.shape = e.shape; r.tri_first = k; r.tri_count = 0; r.flags = 2u;
            r.normalization = o.rects[k].inv_area; r.sum = rcp(o.rects[k].inv_area);
            o.emitters.push_back(r);
            continue;
        }
        r.shape = e.shape; r.tri_first = (uint32_t) o.emit_pmf.size(); r.tri_count = sh.face_count;
        r.flags = (sh.flags & 1u);
        emit_normals = emit_normals || r.flags;
        double sum = 0.0; uint32_t vlo = 0xffffffffu, vhi = 0;
        for (uint32_t k = 0; k < sh.face_count; ++k) {
            const Tri &t = o.tris[sh.first_face + k];
            float area = face_area(ld3(t.p0), ld3(t.p1), ld3(t.p2));
            o.emit_pmf.push_back(area);
            sum += (double) area;
            o.emit_cdf.push_back((float) sum);
            if (area > 0.f) { if (vlo == 0xffffffffu) vlo = k; vhi = k; }
            o.emit_tri.insert(o.emit_tri.end(), t.p0, t.p0 + 3);
            o.emit_tri.insert(o.emit_tri.end(), t.p1, t.p1 + 3);
            o.emit_tri.insert(o.emit_tri.end(), t.p2, t.p2 + 3);
            for (int q = 0; q < 9; ++q)
                o.emit_vnorm.push_back(r.flags ? o.tri_vn[(size_t) (sh.first_face + k) * 9 + q] : 0.f);
        }
        if (vlo == 0xffffffffu) return false;
        r.valid_lo = vlo; r.valid_hi = vhi; r.sum = (float) sum; r.normalization = (float) (1.0 / sum);
        o.emitters.push_back(r);
    }
    if (s->envmap && s->envmap->emitter_index >= s->emitter_count) push_env();
    if (s->envmap) {
        if (MIW_SPECTRAL && !s->envmap->density) return false;     // as the product: coefficient texels need their sampling density (include/miwave.h)
        o.env = envmap_build(*s->envmap);
        if (!o.env.ok) return false;
        o.env.rec.data = o.env.data.data(); o.env.rec.levels = o.env.levels.data();
    }
    SceneView &v = o.view;
    v.env = s->envmap ? &o.env.rec : nullptr;
    v.env_top = nullptr; v.env_top_count = v.env_top_base = 0;
    v.rects = o.rects.empty() ? nullptr : o.rects.data(); v.rect_count = (uint32_t) o.rects.size();
    {   // shape.h: the bounds rule of every triangle hit; same extent as the product (mesh vertices + rectangle corners)
        std::vector<Tri> ext;
        for (const Tri &t : o.tris) if (!t.pad) ext.push_back(t);
        for (size_t k = 0; k < o.rects.size(); ++k) { Tri two[2]; analytic_bounding_tris(o.rects[k], (uint32_t) k, two); ext.push_back(two[0]); ext.push_back(two[1]); }
        v.accept_pad = scene_pad_unit(ext);
    }
    v.tri_bounds = nullptr;
    v.leaf_boxes = nullptr; v.nodes4 = nullptr; v.nodes8 = nullptr;
    v.nodes = nullptr; v.node_count = 0;
    v.tris = o.tris.data(); v.tri_count = (uint32_t) o.tris.size();
    v.tri_vn = o.tri_vn.empty() ? nullptr : o.tri_vn.data();
    v.tri_uv = o.tri_uv.empty() ? nullptr : o.tri_uv.data();
    v.bitmaps = o.bitmaps.empty() ? nullptr : o.bitmaps.data();
    v.bsdf_tables = s->bsdf_table_floats ? s->bsdf_tables : nullptr;   // the caller's array
    v.shapes = o.shapes.data(); v.shape_count = (uint32_t) o.shapes.size();
    v.bsdfs = o.bsdfs.data(); v.bsdf_count = (uint32_t) o.bsdfs.size();
    v.emitters = o.emitters.data(); v.emitter_count = (uint32_t) o.emitters.size();
    scene_view_prepare(v);
    v.emit_tri = o.emit_tri.data(); v.emit_vnorm = emit_normals ? o.emit_vnorm.data() : nullptr;
    v.emit_pmf = o.emit_pmf.data(); v.emit_cdf = o.emit_cdf.data();
    o.accel.on = false;
    if (g_accel.load()) accel_build(o);
    return true;
}

// ---- Scene::ray_intersect / ray_test, observable behaviour -----------------------------------
// scene_native.inl:23-41 -> kdtree.h:2079-2171: closest accepted triangle; the tie
// rule (smaller global primitive id) is this code base's definition.
struct OHit { bool valid; float t, u, v; uint32_t prim; };

// the indexed form of the two loops below: same leaf test, same closest / tie rule; visits only what the ray can reach
template <bool Any>
OHit accel_query(const OScene &sc, const Ray &ray) {
    OHit best{ false, std::numeric_limits<float>::infinity(), 0.f, 0.f, 0xffffffffu };
    const OAccel &A = sc.accel;
    const PrimCtx ctx = prim_ctx(sc.view);
    auto test = [&](uint32_t i) -> bool {
        float t, u, v;
        if (prim_intersect(sc.tris[i], ctx, ray.o, ray.d, ray.mint, ray.maxt, t, u, v)) {
            if (Any) { best.valid = true; return true; }
            if (t < best.t || (t == best.t && i < best.prim)) { best.valid = true; best.t = t; best.u = u; best.v = v; best.prim = i; }
        }
        return false;
    };
    for (uint32_t i : A.always) if (test(i)) return best;
    if (A.nodes.empty()) return best;
    const double o[3] = { ray.o.x, ray.o.y, ray.o.z }, d[3] = { ray.d.x, ray.d.y, ray.d.z };
    const double mint = ray.mint, maxt = ray.maxt;
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0u;
    while (sp > 0) {
        const OAccelNode &nd = A.nodes[stack[--sp]];
        double tn, tf;
        if (!accel_slab(nd, o, d, tn, tf)) continue;
        const double hi = std::min(maxt, (double) best.t);             // ties (t == best t) stay reachable: the test is inclusive
        const double slack = 1e-9;
        if (tn > tf + slack * std::fabs(tf) + 1e-300) continue;
        if (tf < mint - slack * std::fabs(mint) || tn > hi + slack * std::fabs(hi)) continue;
        if (nd.count) {
            for (uint32_t k = nd.first; k < nd.first + nd.count; ++k) if (test(A.prims[k])) return best;
            continue;
        }
        // nearer child last onto the stack (it is popped first): shrinks best.t early; the order does not change the answer
        double ln, lf, rn, rf;
        const bool hl = accel_slab(A.nodes[nd.left], o, d, ln, lf), hr = accel_slab(A.nodes[nd.right], o, d, rn, rf);
        if (sp + 2 > 128) {                                            // cannot happen (median splits: depth <= 32); a checker must not answer "miss" if it does
            fprintf(stderr, "[miw_oracle] accel_query: traversal stack overflow (index deeper than 128 entries) - aborting, no golden answer may come from this run\n");
            abort();
        }
        if (hl && hr && ln < rn) { stack[sp++] = nd.right; stack[sp++] = nd.left; }
        else { if (hl) stack[sp++] = nd.left; if (hr) stack[sp++] = nd.right; }
    }
    return best;
}

OHit ray_intersect_preliminary(const OScene &sc, const Ray &ray) {
    if (sc.accel.on) return accel_query<false>(sc, ray);
    OHit best{ false, std::numeric_limits<float>::infinity(), 0.f, 0.f, 0xffffffffu };
    for (uint32_t i = 0; i < sc.tris.size(); ++i) {
        const Tri &tr = sc.tris[i];
        float t, u, v;
        // kdtree.h:2362-2391 intersect_prim: the mesh's triangle test or the shape's own ray_intersect_preliminary
        if (prim_intersect(tr, prim_ctx(sc.view), ray.o, ray.d, ray.mint, ray.maxt, t, u, v)) {
            if (t < best.t || (t == best.t && i < best.prim)) { best.valid = true; best.t = t; best.u = u; best.v = v; best.prim = i; }
        }
    }
    return best;
}
bool ray_test(const OScene &sc, const Ray &ray) {
    if (sc.accel.on) return accel_query<true>(sc, ray).valid;
    for (const Tri &tr : sc.tris) {
        float t, u, v;
        if (prim_intersect(tr, prim_ctx(sc.view), ray.o, ray.d, ray.mint, ray.maxt, t, u, v)) return true;
    }
    return false;
}
// Scene::ray_intersect: scene_native.inl:23-41 + interaction.h:571-596
bool ray_intersect(const OScene &sc, const Ray &ray, SurfaceInteraction &si) {
    OHit h = ray_intersect_preliminary(sc, ray);
    if (!h.valid) { si.t = std::numeric_limits<float>::infinity(); si.wi = -ray.d; return false; }
    const Tri &tr = sc.tris[h.prim];
    if (tr.pad) {                                          // the analytic shape's own compute_surface_interaction
        const AnalyticRec &a = sc.rects[tr.pad - 1u];
        if (a.kind == ANALYTIC_SPHERE) compute_surface_interaction_sphere(a, h.t, ray.o, ray.d, si);   // sphere.cpp:338-402
        else compute_surface_interaction_rect(a, h.t, h.u, h.v, ray.o, ray.d, si);                     // rectangle.cpp:175-208
    } else {
        const float *vn = (sc.shapes[tr.shape].flags & 1u) ? &sc.tri_vn[(size_t) h.prim * 9] : nullptr;
        const float *tc = (sc.shapes[tr.shape].flags & SHAPE_HAS_TEXCOORDS) ? &sc.tri_uv[(size_t) h.prim * 6] : nullptr;   // mesh.cpp:492-511
        compute_surface_interaction(ld3(tr.p0), ld3(tr.p1), ld3(tr.p2), vn, tc, h.t, h.u, h.v, ray.d, si);
    }
    si.shape = tr.shape; si.prim = h.prim;
    return true;
}

// ---- sampler (src/samplers/independent.cpp, src/librender/sampler.cpp) --------------------------
struct Sampler {
    PCG32 rng; uint64_t base_seed;
    void seed(uint64_t seed_offset) { pcg32_seed(rng, base_seed + seed_offset, MIW_PCG32_DEFAULT_STREAM); }  // sampler.cpp:83-96
    float next_1d() { return pcg32_next_f32(rng); }                                                        // independent.cpp:73-76
    V2 next_2d() { float f1 = next_1d(), f2 = next_1d(); return v2(f1, f2); }                             // :78-82
};

// ---- PathIntegrator::sample, src/integrators/path.cpp:100-211 (scalar semantics) -------------------
struct PathStats { uint64_t segments = 0, shadow_rays = 0; };

void path_sample(const OScene &sc, Sampler &sampler, Ray ray, const Wavelengths &wl, int max_depth, int rr_depth,
                 Spec &result_out, bool &valid_ray_out, PathStats &stats) {
    const SceneView &view = sc.view;
    float eta = 1.f;                                     // :111
    float emission_weight = 1.f;                         // :114
    Spec throughput = spec(1.f), result = spec(0.f);     // :116
    bool active = true;                                  // MTS_MASKED_FUNCTION, fwd.h:290-294

    SurfaceInteraction si;                               // :120
    bool si_valid = ray_intersect(sc, ray, si);
    bool valid_ray = si_valid;                           // :121
    const int32_t env_id = view.env ? (int32_t) view.env->emitter_index : -1;
    // :122, scene.h:243-253: a miss returns scene->environment()
    int32_t emitter = si_valid ? sc.shapes[si.shape].emitter : env_id;
    V3 miss_d = ray.d;                                   // -si.wi of an invalid interaction (interaction.h:591)

    for (int depth = 1;; ++depth) {
        if (emitter >= 0) {                              // :126-129
            if (active) {
                Spec radiance = si_valid ? emitter_eval(sc.emitters[emitter], si.wi, wl)  // area.cpp:63-71
                                         : env_eval_spec(*view.env, miss_d, wl);             // envmap.cpp:134-146
                result = result + emission_weight * throughput * radiance;
            }
        }
        active = active && si_valid;                     // :131

        if (depth > rr_depth) {                          // :137-141
            float q = min_(hmax(throughput) * sqr(eta), .95f);
            active = (sampler.next_1d() < q) && active;
            throughput = throughput * rcp(q);
        }

        if ((uint32_t) depth >= (uint32_t) max_depth || !active)   // :147-149
            break;

        stats.segments++;
        // :154 si.bsdf(ray); a twosided plugin resolves to its front / back record here (twosided.cpp:105-124)
        const BsdfSide bsdf = bsdf_side(sc.bsdfs.data(), sc.shapes[si.shape].bsdf, si.wi);
        bool active_e = active && (bsdf.flags & BSDF_Smooth) != 0;   // :155

        if (active_e) {                                  // :157-172
            // Scene::sample_emitter_direction(si, next_2d, test_visibility = true), scene.cpp:164-214
            DirectionSample ds;
            Spec emitter_val = sample_emitter_direction(view, si.p, sampler.next_2d(), ds, wl);
            if (ds.pdf != 0.f) {                         // scene.cpp:200-207
                Ray shadow;
                shadow.o = si.p; shadow.d = ds.d;
                shadow.mint = MIW_RAY_EPSILON * (1.f + hmax(abs3(si.p)));
                shadow.maxt = ds.dist * (1.f - MIW_SHADOW_EPSILON);
                stats.shadow_rays++;
                if (ray_test(sc, shadow)) emitter_val = spec(0.f);
            }
            active_e = active_e && ds.pdf != 0.f;        // :160
            V3 wo = to_local(si.sh, ds.d);               // :163
            Spec bsdf_val = bsdf_side_eval(bsdf, si.wi, wo, TexCtx(wl, si.uv, view.bitmaps, view.bsdf_tables));   // :164
            float bpdf = bsdf_side_pdf(bsdf, si.wi, wo, TexCtx(wl, si.uv, view.bitmaps, view.bsdf_tables));      // :168
            float mis = mis_weight(ds.pdf, bpdf);        // :170 (ds.delta is false for area lights)
            if (active_e)
                result = result + mis * throughput * bsdf_val * emitter_val;   // :171
        }

        // :177-178 — argument order as Clang evaluates it: next_1d, then next_2d
        float sample1 = sampler.next_1d();
        V2 sample2 = sampler.next_2d();
        BSDFSample bs;
        Spec bsdf_val = bsdf_side_sample(bsdf, si.wi, sample1, sample2, bs, TexCtx(wl, si.uv, view.bitmaps, view.bsdf_tables));

        throughput = throughput * bsdf_val;              // :181
        active = active && !all_zero(throughput);        // :182
        if (!active) break;                              // :183-184

        eta *= bs.eta;                                   // :186

        // :189-190, interaction.h:58-61
        Ray next;
        next.o = si.p; next.d = to_world(si.sh, bs.wo);
        next.mint = (1.f + hmax(abs3(si.p))) * MIW_RAY_EPSILON;
        next.maxt = std::numeric_limits<float>::infinity();
        SurfaceInteraction si_bsdf;
        bool si_bsdf_valid = ray_intersect(sc, next, si_bsdf);

        // :194-205
        emitter = si_bsdf_valid ? sc.shapes[si_bsdf.shape].emitter : env_id;
        miss_d = next.d;
        if (emitter >= 0) {
            // DirectionSample3f ds(si_bsdf, si), records.h:167-173: d = -wi (= ray.d) for environment emitters
            V3 d = next.d; float dist = 0.f; V3 n = v3(0.f);
            if (si_bsdf_valid) {
                d = si_bsdf.p - si.p;
                dist = norm(d);
                d = d / dist;
                n = si_bsdf.sh.n;
            }
            float emitter_pdf = 0.f;
            if (!(bs.sampled_type & BSDF_Delta))
                emitter_pdf = pdf_emitter_direction(view, (uint32_t) emitter, d, dist, n, si.p);   // it = si
            emission_weight = mis_weight(bs.pdf, emitter_pdf);
        }
        si = si_bsdf; si_valid = si_bsdf_valid;          // :207
    }
    result_out = result; valid_ray_out = valid_ray;
}

// ---- DirectIntegrator::sample, src/integrators/direct.cpp:105-198 (scalar semantics) --------------------
struct DirectConfig {
    uint32_t emitter_samples, bsdf_samples; bool hide_emitters;
    float frac_bsdf, frac_lum, weight_bsdf, weight_lum;
    // direct.cpp:82-103
    void set(uint32_t ne, uint32_t nb, bool hide) {
        emitter_samples = ne; bsdf_samples = nb; hide_emitters = hide;
        uint32_t sum = ne + nb;
        weight_bsdf = 1.f / (float) nb; weight_lum = 1.f / (float) ne;
        frac_bsdf = (float) nb / (float) sum; frac_lum = (float) ne / (float) sum;
    }
};

void direct_sample(const OScene &sc, Sampler &sampler, Ray ray, const Wavelengths &wl, const DirectConfig &D,
                   Spec &result_out, bool &valid_ray_out, PathStats &stats) {
    const SceneView &view = sc.view;
    const int32_t env_id = view.env ? (int32_t) view.env->emitter_index : -1;
    SurfaceInteraction si;                               // :113
    bool valid_ray = ray_intersect(sc, ray, si);         // :114
    Spec result = spec(0.f);                             // :116

    if (!D.hide_emitters) {                              // :119-123
        int32_t emitter_vis = valid_ray ? sc.shapes[si.shape].emitter : env_id;
        if (emitter_vis >= 0)
            result = result + (valid_ray ? emitter_eval(sc.emitters[emitter_vis], si.wi, wl) : env_eval_spec(*view.env, ray.d, wl));
    }
    if (!valid_ray) { result_out = result; valid_ray_out = false; return; }   // :125-127
    stats.segments++;

    const BsdfSide bsdf = bsdf_side(sc.bsdfs.data(), sc.shapes[si.shape].bsdf, si.wi);   // :132
    const bool sample_emitter = (bsdf.flags & BSDF_Smooth) != 0;                        // :133

    if (sample_emitter) {                                // :135-161
        for (uint32_t i = 0; i < D.emitter_samples; ++i) {
            DirectionSample ds;
            Spec emitter_val = sample_emitter_direction(view, si.p, sampler.next_2d(), ds, wl);   // :141-142
            if (ds.pdf != 0.f) {                         // test_visibility = true, scene.cpp:200-207
                Ray shadow;
                shadow.o = si.p; shadow.d = ds.d;
                shadow.mint = MIW_RAY_EPSILON * (1.f + hmax(abs3(si.p)));
                shadow.maxt = ds.dist * (1.f - MIW_SHADOW_EPSILON);
                stats.shadow_rays++;
                if (ray_test(sc, shadow)) emitter_val = spec(0.f);
            }
            if (ds.pdf == 0.f) continue;                 // :143-145
            V3 wo = to_local(si.sh, ds.d);               // :148
            Spec bsdf_val = bsdf_side_eval(bsdf, si.wi, wo, TexCtx(wl, si.uv, view.bitmaps, view.bsdf_tables));   // :150
            float bsdf_pdf = bsdf_side_pdf(bsdf, si.wi, wo, TexCtx(wl, si.uv, view.bitmaps, view.bsdf_tables));       // :155
            float mis = mis_weight(ds.pdf * D.frac_lum, bsdf_pdf * D.frac_bsdf) * D.weight_lum;   // :157-158
            result = result + mis * bsdf_val * emitter_val;        // :159
        }
    }

    for (uint32_t i = 0; i < D.bsdf_samples; ++i) {      // :165-196
        float sample1 = sampler.next_1d();               // :166-167, Clang's argument order
        V2 sample2 = sampler.next_2d();
        BSDFSample bs;
        Spec bsdf_val = bsdf_side_sample(bsdf, si.wi, sample1, sample2, bs, TexCtx(wl, si.uv, view.bitmaps, view.bsdf_tables));
        if (all_zero(bsdf_val)) continue;                // :170: active_b
        Ray next;                                        // :173-174, interaction.h:58-61
        next.o = si.p; next.d = to_world(si.sh, bs.wo);
        next.mint = (1.f + hmax(abs3(si.p))) * MIW_RAY_EPSILON;
        next.maxt = std::numeric_limits<float>::infinity();
        SurfaceInteraction si_bsdf;
        bool hit = ray_intersect(sc, next, si_bsdf);
        int32_t emitter = hit ? sc.shapes[si_bsdf.shape].emitter : env_id;   // :177
        if (emitter < 0) continue;                       // :178
        Spec emitter_val = hit ? emitter_eval(sc.emitters[emitter], si_bsdf.wi, wl) : env_eval_spec(*view.env, next.d, wl);   // :181
        // DirectionSample3f ds(si_bsdf, si), records.h:167-173
        V3 d = next.d; float dist = 0.f; V3 n = v3(0.f);
        if (hit) {
            d = si_bsdf.p - si.p;
            dist = norm(d);
            d = d / dist;
            n = si_bsdf.sh.n;
        }
        float emitter_pdf = (bs.sampled_type & BSDF_Delta) ? 0.f
                          : pdf_emitter_direction(view, (uint32_t) emitter, d, dist, n, si.p);   // :189-190
        result = result + bsdf_val * emitter_val * mis_weight(bs.pdf * D.frac_bsdf, emitter_pdf * D.frac_lum) * D.weight_bsdf;   // :192-195
    }
    result_out = result; valid_ray_out = true;
}

// ---- ImageBlock (src/librender/imageblock.cpp) ---------------------------------------------------
struct ImageBlock {
    int off_x = 0, off_y = 0, w = 0, h = 0, border = 0;
    std::vector<float> data;            // (w + 2b) * (h + 2b) * 5
    std::vector<double> data64;         // same layout, exact-sum twin (parity instrument)
    void set(int ox, int oy, int w_, int h_, int b, bool with64) {
        off_x = ox; off_y = oy; w = w_; h = h_; border = b;
        data.assign((size_t) (w + 2 * b) * (h + 2 * b) * 5, 0.f);
        if (with64) data64.assign(data.size(), 0.0); else data64.clear();
    }
};

// imageblock.cpp:79-172 — restated on its own (NOT via miw/film.h) so that the
// shared splat helper is checked against it.
void block_put(ImageBlock &b, const FilmRec &f, V2 pos_, const float *value) {
    const int C = 5;
    bool is_valid = true;
    if (f.warn_negative)                                       // imageblock.cpp:88-91 (m_warn_negative)
        for (int k = 0; k < C; ++k) is_valid = is_valid && value[k] >= -1e-5f;
    for (int k = 0; k < C; ++k) is_valid = is_valid && std::isfinite(value[k]);
    if (!is_valid) return;
    float filter_radius = f.radius;
    int size_x = b.w + 2 * b.border, size_y = b.h + 2 * b.border;
    float pos_x = pos_.x - ((float) (b.off_x - b.border) + .5f),
          pos_y = pos_.y - ((float) (b.off_y - b.border) + .5f);
    if (filter_radius > 0.5f + MIW_RAY_EPSILON) {
        int lo_x = std::max((int) std::ceil(pos_x - filter_radius), 0),
            lo_y = std::max((int) std::ceil(pos_y - filter_radius), 0);
        int hi_x = std::min((int) std::floor(pos_x + filter_radius), size_x - 1),
            hi_y = std::min((int) std::floor(pos_y + filter_radius), size_y - 1);
        int n = (int) std::ceil((f.radius - 2.f * MIW_RAY_EPSILON) * 2.f);
        float base_x = (float) lo_x - pos_x, base_y = (float) lo_y - pos_y;
        float weights_x[16], weights_y[16];
        for (int i = 0; i < n; ++i) {
            float px = base_x + (float) i, py = base_y + (float) i;
            int ix = std::min((int) std::fabs(px * f.scale_factor), MIW_FILTER_RESOLUTION),
                iy = std::min((int) std::fabs(py * f.scale_factor), MIW_FILTER_RESOLUTION);
            weights_x[i] = f.lut[ix]; weights_y[i] = f.lut[iy];
        }
        for (int yr = 0; yr < n; ++yr) {
            int y = lo_y + yr;
            bool enabled = y <= hi_y;
            for (int xr = 0; xr < n; ++xr) {
                int x = lo_x + xr;
                size_t offset = (size_t) C * ((size_t) y * size_x + x);
                float weight = weights_y[yr] * weights_x[xr];
                enabled = enabled && x <= hi_x;
                if (enabled)
                    for (int k = 0; k < C; ++k) {
                        float term = value[k] * weight;
                        b.data[offset + k] += term;
                        if (!b.data64.empty()) b.data64[offset + k] += (double) term;
                    }
            }
        }
    } else {
        int lo_x = (int) std::ceil(pos_x - .5f), lo_y = (int) std::ceil(pos_y - .5f);
        if (lo_x >= 0 && lo_y >= 0 && lo_x < size_x && lo_y < size_y) {
            size_t offset = (size_t) C * ((size_t) lo_y * size_x + lo_x);
            for (int k = 0; k < C; ++k) {
                b.data[offset + k] += value[k];
                if (!b.data64.empty()) b.data64[offset + k] += (double) value[k];
            }
        }
    }
}

// ImageBlock::put(block) into the borderless film (imageblock.cpp:49-77, bitmap.h:657-712)
void film_put(const ImageBlock &b, int crop_x, int crop_y, int crop_w, int crop_h, float *film, double *film64) {
    int sw = b.w + 2 * b.border, sh = b.h + 2 * b.border;
    for (int y = 0; y < sh; ++y) {
        int fy = y + b.off_y - b.border - crop_y;
        if (fy < 0 || fy >= crop_h) continue;
        for (int x = 0; x < sw; ++x) {
            int fx = x + b.off_x - b.border - crop_x;
            if (fx < 0 || fx >= crop_w) continue;
            size_t src = ((size_t) y * sw + x) * 5, dst = ((size_t) fy * crop_w + fx) * 5;
            for (int k = 0; k < 5; ++k) {
                if (film) film[dst + k] += b.data[src + k];
                if (film64) film64[dst + k] += b.data64[src + k];
            }
        }
    }
}

// ---- Spiral (src/librender/spiral.cpp) — own restatement --------------------------------------------
struct SpiralBlock { int off_x, off_y, w, h; uint32_t id; };
std::vector<SpiralBlock> spiral_blocks(int size_x, int size_y, int off_x, int off_y, int bs) {
    int blocks_x = (int) std::ceil((float) size_x / (float) bs), blocks_y = (int) std::ceil((float) size_y / (float) bs);
    int count = blocks_x * blocks_y;
    std::vector<SpiralBlock> out;
    int px = blocks_x / 2, py = blocks_y / 2, dir = 0, steps_left = 1, steps = 1;   // reset(), :19-25
    for (int counter = 0; counter < count;) {
        SpiralBlock b;
        b.id = (uint32_t) counter;                        // :41 with one pass
        int ox = px * bs, oy = py * bs;
        b.w = std::min(bs, size_x - ox); b.h = std::min(bs, size_y - oy);
        b.off_x = ox + off_x; b.off_y = oy + off_y;
        out.push_back(b);
        ++counter;
        if (counter != count) {
            do {                                          // :51-69
                switch (dir) { case 0: ++px; break; case 1: ++py; break; case 2: --px; break; default: --py; break; }
                if (--steps_left == 0) {
                    dir = (dir + 1) % 4;
                    if (dir == 2 || dir == 0) ++steps;
                    steps_left = steps;
                }
            } while (px < 0 || py < 0 || px >= blocks_x || py >= blocks_y);
        }
    }
    return out;
}

void fill_records(const mi_render_cfg *cfg, SensorRec &sensor, FilmRec &film) {
    std::memcpy(sensor.sample_to_camera, cfg->sample_to_camera, 64);
    std::memcpy(sensor.to_world, cfg->to_world, 64);
    sensor.near_clip = cfg->near_clip; sensor.far_clip = cfg->far_clip;
    sensor.pp_offset[0] = cfg->principal_point_offset[0]; sensor.pp_offset[1] = cfg->principal_point_offset[1];
    film.crop_w = cfg->crop_w; film.crop_h = cfg->crop_h; film.crop_x = cfg->crop_x; film.crop_y = cfg->crop_y;
    film.block_size = cfg->block_size; film.border = cfg->filter_border; film.radius = cfg->filter_radius;
    film.scale_factor = (float) MIW_FILTER_RESOLUTION / cfg->filter_radius;
    std::memcpy(film.lut, cfg->filter_lut, sizeof film.lut);
    film.warn_negative = cfg->moment_pass ? 0u : 1u;          // integrator.cpp:113: ImageBlock(..., warn_negative = !has_aovs)
}

} // namespace

extern "C" {

// 0 (default): scene queries by brute force, the definition; 1: through the checker's own spatial index (OAccel above) — same
// answers (tests/test_oracle_accel.py), needed for the configured sizes of configs 3 / 4. Process-wide, read when a scene is built.
void orc_set_accel(int on) { g_accel.store(on ? 1 : 0); }
int orc_get_accel(void) { return g_accel.load(); }

struct orc_stats { uint64_t samples, segments, shadow_rays; double seconds; };

// SamplingIntegrator::render (src/librender/integrator.cpp:51-139) on `n_threads`
// std::threads pulling spiral blocks. film32: crop_w*crop_h*5 float (reference
// semantics: float32 bordered blocks merged in block-id order); film64 (may be
// NULL): the same samples summed in double — the exact-sum instrument the device
// film is compared against bit for bit. `only_blocks`/`n_only`: optional subset
// of block ids to render (bounded CPU-baseline sample); NULL = all.
int orc_render(const mi_scene_desc *scene, const mi_render_cfg *cfg, float *film32, double *film64,
               int n_threads, const uint32_t *only_blocks, uint32_t n_only, orc_stats *stats_out) {
    OScene sc;
    if (!build_scene(scene, sc)) return -1;
    SensorRec sensor; FilmRec film;
    fill_records(cfg, sensor, film);
    const int bs = cfg->block_size;
    std::vector<SpiralBlock> blocks = spiral_blocks(cfg->crop_w, cfg->crop_h, cfg->crop_x, cfg->crop_y, bs);
    // the caller's block-id table must agree with this spiral (host logic cross-check); with several passes
    // (samples_per_pass < sample_count) the ids of a pass are shifted by a multiple of the block count, spiral.cpp:41
    uint32_t id_offset = 0;
    {
        int nbx = (cfg->crop_w + bs - 1) / bs;
        if (cfg->block_ids && !blocks.empty()) {
            const SpiralBlock &b0 = blocks[0];
            id_offset = cfg->block_ids[((b0.off_y - cfg->crop_y) / bs) * nbx + (b0.off_x - cfg->crop_x) / bs] - b0.id;
            if (id_offset % (uint32_t) blocks.size() != 0) return -2;
            for (const SpiralBlock &b : blocks) {
                int bx = (b.off_x - cfg->crop_x) / bs, by = (b.off_y - cfg->crop_y) / bs;
                if (cfg->block_ids[by * nbx + bx] != b.id + id_offset) return -2;
            }
        }
    }
    std::vector<uint32_t> todo;
    if (only_blocks) todo.assign(only_blocks, only_blocks + n_only);
    else if (cfg->tile_list) {
        int nbx = (cfg->crop_w + bs - 1) / bs;
        std::map<uint32_t, uint32_t> rm_to_id;
        for (const SpiralBlock &b : blocks) rm_to_id[(uint32_t) (((b.off_y - cfg->crop_y) / bs) * nbx + (b.off_x - cfg->crop_x) / bs)] = b.id;
        for (uint32_t i = 0; i < cfg->tile_count; ++i) todo.push_back(rm_to_id[cfg->tile_list[i]]);
        std::sort(todo.begin(), todo.end());
    } else for (const SpiralBlock &b : blocks) todo.push_back(b.id);

    size_t film_n = (size_t) cfg->crop_w * cfg->crop_h * 5;
    if (!cfg->accumulate) {                                  // earlier passes (smaller block ids) stay in the film
        if (film32) std::memset(film32, 0, film_n * sizeof(float));
        if (film64) std::memset(film64, 0, film_n * sizeof(double));
    }

    auto t0 = std::chrono::steady_clock::now();
    std::atomic<size_t> next{0};
    std::mutex mutex;
    std::map<size_t, ImageBlock> ready;    // finished blocks waiting for their turn
    size_t next_merge = 0;
    std::atomic<uint64_t> total_samples{0}, total_segments{0}, total_shadow{0};
    if (n_threads < 1) n_threads = 1;

    DirectConfig direct_cfg; direct_cfg.set(cfg->emitter_samples, cfg->bsdf_samples, cfg->hide_emitters != 0);
    auto worker = [&]() {
        FtzScope ftz;                                         // :117
        Sampler sampler; sampler.base_seed = cfg->base_seed;  // sampler->clone(), :113
        PathStats st; uint64_t samples = 0;
        for (;;) {
            size_t job = next.fetch_add(1);
            if (job >= todo.size()) break;
            const SpiralBlock &blk = blocks[todo[job]];
            ImageBlock block;                                 // :114-116: bordered block
            block.set(blk.off_x, blk.off_y, blk.w, blk.h, cfg->filter_border, film64 != nullptr);
            // render_block, :181-209 (scalar branch)
            uint32_t pixel_count = (uint32_t) (bs * bs);
            for (uint32_t i = 0; i < pixel_count; ++i) {
                sampler.seed((uint64_t) (blk.id + id_offset) * pixel_count + i);   // :198
                uint32_t px, py; morton_decode2(i, px, py);                         // :200
                if ((int) px >= blk.w || (int) py >= blk.h) continue;               // :201-202
                float pos_x = (float) (px + (uint32_t) blk.off_x), pos_y = (float) (py + (uint32_t) blk.off_y);   // :204
                for (uint32_t j = 0; j < cfg->spp; ++j) {
                    // render_sample, :233-288
                    V2 jit = sampler.next_2d();
                    V2 position_sample = v2(pos_x + jit.x, pos_y + jit.y);          // :242
                    float wavelength_sample = sampler.next_1d();                    // :252
                    V2 adjusted = v2((position_sample.x - (float) cfg->crop_x) / (float) cfg->crop_w,
                                     (position_sample.y - (float) cfg->crop_y) / (float) cfg->crop_h);   // :254-256
                    Ray ray = sensor_sample_ray(sensor, adjusted);                  // :258
                    Wavelengths wl; Spec ray_weight;
#if MIW_SPECTRAL
                    sample_wavelengths(wavelength_sample, wl, ray_weight);          // perspective.cpp:191-192, spectrum.h:305-314
#else
                    (void) wavelength_sample; ray_weight = spec(1.f);
#endif
                    Spec L; bool valid;
                    if (cfg->integrator == MI_INTEGRATOR_DIRECT) direct_sample(sc, sampler, ray, wl, direct_cfg, L, valid, st);
                    else path_sample(sc, sampler, ray, wl, cfg->max_depth, cfg->rr_depth, L, valid, st);   // :264
#if MIW_SPECTRAL
                    V3 xyz = spectrum_to_xyz(ray_weight * L, wl);                   // :266-271
#else
                    V3 xyz = srgb_to_xyz(L);                                        // :272-273
#endif
                    float aovs[5] = { xyz.x, xyz.y, xyz.z, valid ? 1.f : 0.f, 1.f };   // :279-283
                    if (cfg->moment_pass) {
                        // MomentIntegrator::sample, moment.cpp:66-91 with one nested integrator: the block holds
                        // X Y Z A W | nested.X nested.Y nested.Z | m2_nested.X m2_nested.Y m2_nested.Z, all splatted by the
                        // one put() below (:285) — same weights, one validity verdict for the eleven values
                        float full[11] = { xyz.x, xyz.y, xyz.z, aovs[3], 1.f, xyz.x, xyz.y, xyz.z, 0.f, 0.f, 0.f };
                        for (int k = 0; k < 3; ++k) full[8 + k] = full[5 + k] * full[5 + k];      // :87 sqr()
                        bool all_valid = true;
                        for (int k = 0; k < 11; ++k) all_valid = all_valid && std::isfinite(full[k]);   // imageblock.cpp:93-96; warn_negative is off: the block has AOVs (integrator.cpp:113)
                        if (cfg->moment_pass == MI_MOMENT_SQUARES) for (int k = 0; k < 3; ++k) aovs[k] = full[8 + k];
                        if (!all_valid) { ++samples; continue; }                    // put() warns and drops it; :287 still advances
                    }
                    block_put(block, film, position_sample, aovs);                  // :285
                    ++samples;                                                      // sampler->advance(), :287
                }
            }
            // film->put(block), :130 — merged in block-id order so that the float32
            // film is deterministic (the reference's order is thread-timing dependent,
            // independent.cpp:36-40)
            std::lock_guard<std::mutex> lock(mutex);
            ready.emplace(job, std::move(block));
            while (!ready.empty() && ready.begin()->first == next_merge) {
                film_put(ready.begin()->second, cfg->crop_x, cfg->crop_y, cfg->crop_w, cfg->crop_h, film32, film64);
                ready.erase(ready.begin());
                ++next_merge;
            }
        }
        total_samples += samples; total_segments += st.segments; total_shadow += st.shadow_rays;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (stats_out) {
        stats_out->samples = total_samples; stats_out->segments = total_segments; stats_out->shadow_rays = total_shadow;
        stats_out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

// Scene::ray_intersect_preliminary / ray_test by brute force (the definition the BVH must match)
int orc_trace(const mi_scene_desc *scene, const mi_rays_soa *r, const mi_hits_soa *h, uint64_t n, int any_hit) {
    OScene sc;
    if (!build_scene(scene, sc)) return -1;
    FtzScope ftz;
    for (uint64_t i = 0; i < n; ++i) {
        Ray ray; ray.o = v3(r->ox[i], r->oy[i], r->oz[i]); ray.d = v3(r->dx[i], r->dy[i], r->dz[i]);
        ray.mint = r->mint[i]; ray.maxt = r->maxt[i];
        if (any_hit) {
            h->t[i] = ray_test(sc, ray) ? 0.f : std::numeric_limits<float>::infinity();
        } else {
            OHit o = ray_intersect_preliminary(sc, ray);
            h->t[i] = o.valid ? o.t : std::numeric_limits<float>::infinity();
            if (h->u) h->u[i] = o.u;
            if (h->v) h->v[i] = o.v;
            if (h->prim) h->prim[i] = o.valid ? o.prim : 0xffffffffu;
            if (h->shape) h->shape[i] = o.valid ? sc.tris[o.prim].shape : 0xffffffffu;
        }
    }
    return 0;
}

// Full SurfaceInteraction for one ray: out = t, p.xyz, n.xyz, sh_n.xyz, sh_s.xyz, sh_t.xyz, wi.xyz, uv (21)
int orc_ray_intersect_full(const mi_scene_desc *scene, const float *ray8, float *out21) {
    OScene sc;
    if (!build_scene(scene, sc)) return -1;
    FtzScope ftz;
    Ray ray; ray.o = v3(ray8[0], ray8[1], ray8[2]); ray.d = v3(ray8[3], ray8[4], ray8[5]); ray.mint = ray8[6]; ray.maxt = ray8[7];
    SurfaceInteraction si; std::memset(&si, 0, sizeof si);
    bool ok = ray_intersect(sc, ray, si);
    float *o = out21;
    o[0] = si.t; o[1] = si.p.x; o[2] = si.p.y; o[3] = si.p.z; o[4] = si.n.x; o[5] = si.n.y; o[6] = si.n.z;
    o[7] = si.sh.n.x; o[8] = si.sh.n.y; o[9] = si.sh.n.z; o[10] = si.sh.s.x; o[11] = si.sh.s.y; o[12] = si.sh.s.z;
    o[13] = si.sh.t.x; o[14] = si.sh.t.y; o[15] = si.sh.t.z; o[16] = si.wi.x; o[17] = si.wi.y; o[18] = si.wi.z;
    o[19] = si.uv.x; o[20] = si.uv.y;
    return ok ? 1 : 0;
}

// ---- the Scene query surface, scalar semantics (the checker of mi_ray_intersect / mi_sample_emitter_direction /
// mi_pdf_emitter_direction / mi_emitter_eval): Scene::ray_intersect (scene.cpp:113-121), sample_emitter_direction with
// its visibility test (:164-214), pdf_emitter_direction (:216-231), si.emitter(scene)->eval(si) (scene.h:243-253). ----
static void fill_si_record(const OScene &sc, const Ray &ray, mi_surface_interaction &r) {
    SurfaceInteraction si; std::memset(&si, 0, sizeof si);
    std::memset(&r, 0, sizeof r);
    if (ray_intersect(sc, ray, si)) {
        r.t = si.t;
        const V3 *src[6] = { &si.p, &si.n, &si.sh.s, &si.sh.t, &si.sh.n, &si.wi };
        float *dst[6] = { r.p, r.n, r.sh_s, r.sh_t, r.sh_n, r.wi };
        for (int k = 0; k < 6; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
        r.uv[0] = si.uv.x; r.uv[1] = si.uv.y;
        r.prim_index = sc.tris[si.prim].prim; r.shape_index = si.shape;
        r.emitter_index = sc.shapes[si.shape].emitter;
    } else {
        r.t = std::numeric_limits<float>::infinity();
        r.wi[0] = -ray.d.x; r.wi[1] = -ray.d.y; r.wi[2] = -ray.d.z;
        r.prim_index = r.shape_index = 0xffffffffu;
        r.emitter_index = sc.view.env ? (int32_t) sc.view.env->emitter_index : -1;
    }
}
int orc_ray_intersect(const mi_scene_desc *scene, const mi_rays_soa *r, mi_surface_interaction *out, uint64_t n) {
    OScene sc;
    if (!build_scene(scene, sc)) return -1;
    FtzScope ftz;
    for (uint64_t i = 0; i < n; ++i) {
        Ray ray; ray.o = v3(r->ox[i], r->oy[i], r->oz[i]); ray.d = v3(r->dx[i], r->dy[i], r->dz[i]); ray.mint = r->mint[i]; ray.maxt = r->maxt[i];
        fill_si_record(sc, ray, out[i]);
    }
    return 0;
}
int orc_sample_emitter_direction(const mi_scene_desc *scene, int32_t emitter, const float *ref_p, const float *sample, const float *wavelengths,
                                 int32_t test_visibility, mi_direction_sample *out, float *spec_out, uint64_t n) {
    OScene sc;
    if (!build_scene(scene, sc)) return -1;
    if (emitter >= (int32_t) sc.view.emitter_count) return -1;
    FtzScope ftz;
    for (uint64_t i = 0; i < n; ++i) {
        Wavelengths wl;
#if MIW_SPECTRAL
        for (int k = 0; k < 4; ++k) wl.l[k] = wavelengths[4 * i + k];
#else
        (void) wavelengths;
#endif
        const V3 ref = v3(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]);
        const V2 u = v2(sample[2 * i], sample[2 * i + 1]);
        DirectionSample ds;
        Spec value = emitter < 0 ? sample_emitter_direction(sc.view, ref, u, ds, wl) : emitter_sample_direction(sc.view, (uint32_t) emitter, ref, u, ds, wl);
        if (test_visibility && ds.pdf != 0.f) {                // scene.cpp:203-207
            Ray shadow; shadow.o = ref; shadow.d = ds.d; shadow.mint = spawn_mint(ref); shadow.maxt = ds.dist * (1.f - MIW_SHADOW_EPSILON);
            if (ray_test(sc, shadow)) value = spec(0.f);
        }
        mi_direction_sample &r = out[i];
        r.p[0] = ds.p.x; r.p[1] = ds.p.y; r.p[2] = ds.p.z; r.n[0] = ds.n.x; r.n[1] = ds.n.y; r.n[2] = ds.n.z;
        r.d[0] = ds.d.x; r.d[1] = ds.d.y; r.d[2] = ds.d.z; r.dist = ds.dist; r.pdf = ds.pdf; r.emitter_index = (int32_t) ds.emitter;
        const float *vf = reinterpret_cast<const float *>(&value);
        for (int k = 0; k < MIW_SPEC_N; ++k) spec_out[MIW_SPEC_N * i + k] = vf[k];
    }
    return 0;
}
int orc_pdf_emitter_direction(const mi_scene_desc *scene, int32_t emitter, const float *ref_p, const mi_direction_sample *ds, float *pdf, uint64_t n) {
    OScene sc;
    if (!build_scene(scene, sc)) return -1;
    FtzScope ftz;
    for (uint64_t i = 0; i < n; ++i) {
        const mi_direction_sample &r = ds[i];
        const V3 ref = v3(ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]);
        const uint32_t e = emitter < 0 ? (uint32_t) r.emitter_index : (uint32_t) emitter;
        float v = 0.f;
        if (e < sc.view.emitter_count)
            v = emitter < 0 ? pdf_emitter_direction(sc.view, e, ld3(r.d), r.dist, ld3(r.n), ref) : emitter_pdf_direction(sc.view, e, ld3(r.d), r.dist, ld3(r.n), ref);
        pdf[i] = v;
    }
    return 0;
}
int orc_emitter_eval(const mi_scene_desc *scene, const mi_surface_interaction *si, const float *wavelengths, float *spec_out, uint64_t n) {
    OScene sc;
    if (!build_scene(scene, sc)) return -1;
    FtzScope ftz;
    for (uint64_t i = 0; i < n; ++i) {
        Wavelengths wl;
#if MIW_SPECTRAL
        for (int k = 0; k < 4; ++k) wl.l[k] = wavelengths[4 * i + k];
#else
        (void) wavelengths;
#endif
        const mi_surface_interaction &r = si[i];
        Spec value = spec(0.f);
        if (r.emitter_index >= 0 && (uint32_t) r.emitter_index < sc.view.emitter_count) {
            const EmitterRec &e = sc.view.emitters[r.emitter_index];
            if (e.type == EMITTER_ENVMAP) { if (sc.view.env) value = env_eval_spec(*sc.view.env, -ld3(r.wi), wl); }
            else value = emitter_eval(e, ld3(r.wi), wl);
        }
        const float *vf = reinterpret_cast<const float *>(&value);
        for (int k = 0; k < MIW_SPEC_N; ++k) spec_out[MIW_SPEC_N * i + k] = vf[k];
    }
    return 0;
}

// ---- known-answer entry points -------------------------------------------------------------------
float  orc_tea_float32(uint32_t v0, uint32_t v1, int rounds) { return sample_tea_float32(v0, v1, rounds); }
double orc_tea_float64(uint32_t v0, uint32_t v1, int rounds) { return sample_tea_float64(v0, v1, rounds); }
uint32_t orc_tea_32(uint32_t v0, uint32_t v1, int rounds) { return sample_tea_32(v0, v1, rounds); }
void orc_pcg32_u32(uint64_t initstate, uint64_t initseq, uint32_t *out, int n) {
    PCG32 r; pcg32_seed(r, initstate, initseq);
    for (int i = 0; i < n; ++i) out[i] = pcg32_next_u32(r);
}
void orc_pcg32_f32(uint64_t initstate, uint64_t initseq, float *out, int n) {
    PCG32 r; pcg32_seed(r, initstate, initseq);
    for (int i = 0; i < n; ++i) out[i] = pcg32_next_f32(r);
}
void orc_morton_decode(uint32_t i, uint32_t *xy) { morton_decode2(i, xy[0], xy[1]); }
void orc_fresnel(float cos_theta_i, float eta, float *out4) { FtzScope f; fresnel(cos_theta_i, eta, out4[0], out4[1], out4[2], out4[3]); }
float orc_fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) { FtzScope f; return fresnel_conductor(cos_theta_i, eta_r, eta_i); }
// microfacet: op 0 eval(m) 1 pdf(wi,m) 2 smith_g1(v=wi,m) 3 sample(wi,u) -> m.xyz,pdf
void orc_microfacet(int op, uint32_t type, float au, float av, int sample_visible, const float *wi, const float *m_or_u, float *out) {
    FtzScope f;
    Microfacet d = microfacet_make(type, au, av, sample_visible != 0);
    V3 w = v3(wi[0], wi[1], wi[2]);
    switch (op) {
        case 0: out[0] = mf_eval(d, v3(m_or_u[0], m_or_u[1], m_or_u[2])); break;
        case 1: out[0] = mf_pdf(d, w, v3(m_or_u[0], m_or_u[1], m_or_u[2])); break;
        case 2: out[0] = mf_smith_g1(d, w, v3(m_or_u[0], m_or_u[1], m_or_u[2])); break;
        default: { V3 m; float pdf; mf_sample(d, w, v2(m_or_u[0], m_or_u[1]), m, pdf); out[0] = m.x; out[1] = m.y; out[2] = m.z; out[3] = pdf; }
    }
}
// Hierarchical2D<Float, 0> over caller data (w x h floats): op 0 = sample(xy) -> x, y, pdf ; op 1 = eval(xy) -> pdf ;
// op 200 + k = the same through hier2d_sample<Paired = false>, one level per lookup (round 5: the default reads two levels per lookup);
// op 100 + k = sample(xy) with the hierarchy's k smallest levels read through a SEPARATE copy (envmap.h: EnvTop — what the device
// kernels do with their LDS copy of those levels): must be the sample of op 0, bit for bit, for every k
int orc_hier2d(const float *data, uint32_t w, uint32_t h, int op, const float *xy, float *out3) {
    FtzScope f;
    // reuse the envmap table builder on a grey bitmap whose luminance * sin(theta) equals `data` is not possible in
    // general (sin(theta) = 0 at the poles), so the hierarchy is built here directly from `data`
    // exactly as envmap_build does for its luminance array.
    std::vector<float> rgba((size_t) w * h * 4, 1.f);
    mi_envmap e; std::memset(&e, 0, sizeof e);
    e.rgba = rgba.data(); e.width = w; e.height = h; e.scale = 1.f;
    for (int i = 0; i < 4; ++i) e.to_world[i * 5] = 1.f;
    e.bsphere_radius = 1.f;
    EnvmapTables t = envmap_build_from_density(e, data);
    if (!t.ok) return -1;
    t.rec.data = t.data.data(); t.rec.levels = t.levels.data();
    if (op == 0) { float pdf; V2 r = hier2d_sample(t.rec, v2(xy[0], xy[1]), pdf); out3[0] = r.x; out3[1] = r.y; out3[2] = pdf; }
    else if (op >= 100) {
        const bool plain = op >= 200;                               // op 200 + k: the one-level-per-fetch descent (Paired = false), the definition the paired form must equal
        const uint32_t k = (uint32_t) (op - (plain ? 200 : 100));
        if (k >= t.rec.n_levels) return -2;
        EnvTop top = env_top_none();
        std::vector<float> copy;
        if (k) {
            top.base = t.rec.level_offset[t.rec.n_levels - k]; top.count = k;
            copy.assign(t.levels.begin() + top.base, t.levels.end());
            top.p = copy.data();
            for (size_t i = top.base; i < t.levels.size(); ++i) t.levels[i] = -1e30f;     // the originals of those levels must not be read
        }
        float pdf; V2 r = plain ? hier2d_sample<false>(t.rec, v2(xy[0], xy[1]), pdf, top) : hier2d_sample<true>(t.rec, v2(xy[0], xy[1]), pdf, top);
        out3[0] = r.x; out3[1] = r.y; out3[2] = pdf;
    }
    else out3[0] = hier2d_eval(t.rec, v2(xy[0], xy[1]));
    return 0;
}
#if MIW_SPECTRAL
// spectrum.h restatements for the known-answer tests (src/librender/tests/test_spectra.py):
// op 0: cie1931_xyz(in[0]) -> xyz ; 1: d65 (scale in[0]) at in[1] ; 2: sample_wavelengths(in[0]) -> wl[4], weight[4]
// op 3: texture record {type = in[0] bits, v = in[1..4]} at wavelengths in[5..8] -> 4 values
void orc_spectral(int op, const float *in, float *out) {
    FtzScope f;
    if (op == 0) { V3 v = cie1931_xyz(in[0]); out[0] = v.x; out[1] = v.y; out[2] = v.z; }
    else if (op == 1) out[0] = d65_eval(in[0], in[1]);
    else if (op == 2) { Wavelengths wl; Spec w; sample_wavelengths(in[0], wl, w); for (int i = 0; i < 4; ++i) { out[i] = wl.l[i]; out[4 + i] = w.c[i]; } }
    else { TexRec t; t.type = f2u(in[0]); for (int i = 0; i < 4; ++i) t.v[i] = in[1 + i];
           Wavelengths wl; for (int i = 0; i < 4; ++i) wl.l[i] = in[5 + i];
           Spec r = tex_eval(t, wl); for (int i = 0; i < 4; ++i) out[i] = r.c[i]; }
}
#endif
// special.h restatements: out4 = exp, log, erf, erfinv of x
void orc_special(float x, float *out4) {
    FtzScope f;
    out4[0] = exp_(x); out4[1] = log_(x); out4[2] = erf_(x); out4[3] = erfinv_(x);
}
// Mesh::ray_intersect_triangle: tri9 = p0,p1,p2 ; ray8 ; out = hit, t, u, v
void orc_ray_triangle(const float *tri9, const float *ray8, float *out4) {
    FtzScope f;
    float t, u, v;
    bool hit = ray_intersect_triangle(ld3(tri9), ld3(tri9 + 3), ld3(tri9 + 6), v3(ray8[0], ray8[1], ray8[2]),
                                      v3(ray8[3], ray8[4], ray8[5]), ray8[6], ray8[7], t, u, v);
    out4[0] = hit ? 1.f : 0.f; out4[1] = t; out4[2] = u; out4[3] = v;
}
// ImageBlock::put for n samples into a fresh bordered block; out = (w+2b)*(h+2b)*5 floats
int orc_imageblock_put(const mi_render_cfg *cfg, int off_x, int off_y, int w, int h, int border,
                       const float *pos_xy, const float *values5, int n, float *out) {
    FtzScope f;
    SensorRec sensor; FilmRec film; fill_records(cfg, sensor, film);
    ImageBlock b; b.set(off_x, off_y, w, h, border, false);
    for (int i = 0; i < n; ++i) block_put(b, film, v2(pos_xy[2 * i], pos_xy[2 * i + 1]), values5 + 5 * i);
    std::memcpy(out, b.data.data(), b.data.size() * sizeof(float));
    return (int) b.data.size();
}
// the shared splat helper (miw/film.h) over the same samples, into a film-sized f64 buffer
void orc_film_splat_shared(const mi_render_cfg *cfg, const float *pos_xy, const float *values5, int n, double *film64) {
    FtzScope f;
    SensorRec sensor; FilmRec film; fill_records(cfg, sensor, film);
    for (int i = 0; i < n; ++i) {
        int px = (int) std::floor(pos_xy[2 * i]), py = (int) std::floor(pos_xy[2 * i + 1]);
        film_splat(film, px, py, v2(pos_xy[2 * i], pos_xy[2 * i + 1]), values5 + 5 * i,
                   [&](int texel, int k, float v) { film64[(size_t) texel * 5 + k] += (double) v; });
    }
}
int orc_spiral(int w, int h, int off_x, int off_y, int bs, int32_t *out5, int capacity) {
    std::vector<SpiralBlock> b = spiral_blocks(w, h, off_x, off_y, bs);
    for (size_t i = 0; i < b.size() && (int) i < capacity; ++i) {
        out5[i * 5] = b[i].off_x; out5[i * 5 + 1] = b[i].off_y; out5[i * 5 + 2] = b[i].w; out5[i * 5 + 3] = b[i].h; out5[i * 5 + 4] = (int32_t) b[i].id;
    }
    return (int) b.size();
}
void orc_coordinate_system(const float *n, float *st6) {
    FtzScope f; V3 s, t; coordinate_system(v3(n[0], n[1], n[2]), s, t);
    st6[0] = s.x; st6[1] = s.y; st6[2] = s.z; st6[3] = t.x; st6[4] = t.y; st6[5] = t.z;
}
void orc_warp(int op, const float *u2, float *out) {
    FtzScope f;
    V2 u = v2(u2[0], u2[1]);
    switch (op) {
        case 0: { V2 p = square_to_uniform_disk_concentric(u); out[0] = p.x; out[1] = p.y; } break;
        case 1: { V3 p = square_to_cosine_hemisphere(u); out[0] = p.x; out[1] = p.y; out[2] = p.z; out[3] = square_to_cosine_hemisphere_pdf(p); } break;
        default: { V2 p = square_to_uniform_triangle(u); out[0] = p.x; out[1] = p.y; }
    }
}

// Mirror of the device-side mi_eval: same ops, same layouts, evaluated on the CPU.
int orc_eval(int op, const mi_scene_desc *scene, const mi_render_cfg *cfg, const float *in, int is, float *out, int os, uint64_t n) {
    FtzScope ftz;
    OScene sc; bool have_scene = false;
    if (scene) { if (!build_scene(scene, sc)) return -1; have_scene = true; }
    SensorRec sensor; FilmRec film;
    if (cfg) fill_records(cfg, sensor, film);
    for (uint64_t i = 0; i < n; ++i) {
        const float *a = in + i * (uint64_t) is; float *o = out + i * (uint64_t) os;
        switch (op) {
            case MI_EVAL_PCG32: {
                PCG32 r; pcg32_seed(r, (uint64_t) f2u(a[0]) | ((uint64_t) f2u(a[1]) << 32), MIW_PCG32_DEFAULT_STREAM);
                for (int k = 0; k < 8; ++k) o[k] = pcg32_next_f32(r);
            } break;
            case MI_EVAL_SINCOS: sincos_(a[0], o[0], o[1]); break;
            case MI_EVAL_COSINE_HEMISPHERE: {
                V3 w = square_to_cosine_hemisphere(v2(a[0], a[1]));
                o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = square_to_cosine_hemisphere_pdf(w);
            } break;
            case MI_EVAL_BSDF: {
                if (!have_scene) return -1;
                const uint32_t b_index = f2u(a[0]);
                V3 wi = v3(a[1], a[2], a[3]), wo = v3(a[7], a[8], a[9]);
                Wavelengths wl;
#if MIW_SPECTRAL
                for (int k = 0; k < 4; ++k) wl.l[k] = a[10 + k];
#endif
                const BsdfSide b = bsdf_side(sc.bsdfs.data(), b_index, wi);
                const TexCtx tc(wl, v2(0.f, 0.f), nullptr, sc.view.bsdf_tables);
                BSDFSample bs; Spec w = bsdf_side_sample(b, wi, a[4], v2(a[5], a[6]), bs, tc);
                o[0] = bs.wo.x; o[1] = bs.wo.y; o[2] = bs.wo.z; o[3] = bs.pdf; o[4] = bs.eta; o[5] = u2f(bs.sampled_type);
                Spec e = bsdf_side_eval(b, wi, wo, tc);
                const float *wf = reinterpret_cast<const float *>(&w), *ef = reinterpret_cast<const float *>(&e);
                for (int k = 0; k < MIW_SPEC_N; ++k) { o[6 + k] = wf[k]; o[6 + MIW_SPEC_N + k] = ef[k]; }
                o[6 + 2 * MIW_SPEC_N] = bsdf_side_pdf(b, wi, wo, tc);
            } break;
            case MI_EVAL_FRESNEL: fresnel(a[0], a[1], o[0], o[1], o[2], o[3]); break;
            case MI_EVAL_CAMERA_RAY: {
                if (!cfg) return -1;
                V2 adj = v2((a[0] - (float) film.crop_x) / (float) film.crop_w, (a[1] - (float) film.crop_y) / (float) film.crop_h);
                Ray r = sensor_sample_ray(sensor, adj);
                o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.mint; o[7] = r.maxt;
            } break;
            case MI_EVAL_EMITTER_SAMPLE: {
                if (!have_scene) return -1;
                Wavelengths wl;
#if MIW_SPECTRAL
                for (int k = 0; k < 4; ++k) wl.l[k] = a[5 + k];
#endif
                DirectionSample ds; Spec sv = sample_emitter_direction(sc.view, v3(a[0], a[1], a[2]), v2(a[3], a[4]), ds, wl);
                o[0] = ds.d.x; o[1] = ds.d.y; o[2] = ds.d.z; o[3] = ds.dist; o[4] = ds.pdf;
                o[5] = ds.p.x; o[6] = ds.p.y; o[7] = ds.p.z; o[8] = ds.n.x; o[9] = ds.n.y; o[10] = ds.n.z;
                const float *sf = reinterpret_cast<const float *>(&sv);
                for (int k = 0; k < MIW_SPEC_N; ++k) o[11 + k] = sf[k];
            } break;
            case MI_EVAL_FP_SEMANTICS: {
                float x = a[0], y = a[1], z = a[2];
                o[0] = x + y; o[1] = x * y; o[2] = x / y; o[3] = __builtin_sqrtf(abs_(x));
                o[4] = fmadd(x, y, z); o[5] = rcp(x); o[6] = min_(x, y); o[7] = max_(x, y);
            } break;
            case MI_EVAL_SPECIAL: o[0] = exp_(a[0]); o[1] = log_(a[0]); o[2] = erf_(a[0]); o[3] = erfinv_(a[0]); break;
            case MI_EVAL_ENVMAP: {
                if (!have_scene || !sc.view.env) return -1;
                Wavelengths wl;
#if MIW_SPECTRAL
                for (int k = 0; k < 4; ++k) wl.l[k] = a[8 + k];
#endif
                V3 d = v3(a[0], a[1], a[2]);
                const Spec e = env_eval_spec(*sc.view.env, d, wl);
                V3 sd, sp, sn; float dist, pdf;
                const Spec ss = env_sample_direction_spec(*sc.view.env, v3(a[3], a[4], a[5]), v2(a[6], a[7]), sd, dist, pdf, sp, sn, wl);
                const float *ef = reinterpret_cast<const float *>(&e), *sf = reinterpret_cast<const float *>(&ss);
            for (int k = 0; k < MIW_SPEC_N; ++k) { o[k] = ef[k]; o[MIW_SPEC_N + 6 + k] = sf[k]; }
                o[MIW_SPEC_N] = env_pdf_direction(*sc.view.env, d);
                o[MIW_SPEC_N + 1] = sd.x; o[MIW_SPEC_N + 2] = sd.y; o[MIW_SPEC_N + 3] = sd.z; o[MIW_SPEC_N + 4] = dist; o[MIW_SPEC_N + 5] = pdf;
            } break;
#if MIW_SPECTRAL
            case MI_EVAL_SPECTRUM: {
                Wavelengths wl; Spec wt;
                sample_wavelengths(a[0], wl, wt);
                TexRec t; t.type = TEX_SRGB_D65; t.v[0] = a[1]; t.v[1] = a[2]; t.v[2] = a[3]; t.v[3] = a[4];
                Spec sd = tex_eval(t, wl);
                t.type = TEX_SRGB; Spec sr = tex_eval(t, wl);
                for (int k = 0; k < 4; ++k) { o[k] = wl.l[k]; o[4 + k] = wt.c[k]; o[8 + k] = sr.c[k]; o[12 + k] = sd.c[k]; }
                V3 xyz = spectrum_to_xyz(wt * sd, wl);
                o[16] = xyz.x; o[17] = xyz.y; o[18] = xyz.z;
            } break;
#endif
            case MI_EVAL_INVTRIG: o[0] = atan2_(a[0], a[1]); o[1] = acos_(a[1]); o[2] = asin_(a[1]); break;
            case MI_EVAL_TEXTURE: {
                if (!have_scene || !sc.view.bitmaps) return -1;
                Wavelengths wl;
    #if MIW_SPECTRAL
                Spec wt; sample_wavelengths(a[3], wl, wt);
    #endif
                TexRec t; t.type = TEX_BITMAP; t.v[0] = a[2]; t.v[1] = t.v[2] = t.v[3] = 0.f;
                Spec r = tex_eval(t, TexCtx(wl, v2(a[0], a[1]), sc.view.bitmaps));
                const float *rf = reinterpret_cast<const float *>(&r);
                for (int k = 0; k < MIW_SPEC_N; ++k) o[k] = rf[k];
            } break;
            default: return -1;
        }
    }
    return 0;
}

} // extern "C"
