// The direct-illumination integrator (src/integrators/direct.cpp:105-198) on the register-resident plan.
//
// DirectIntegrator::sample makes 1 + emitter_samples + bsdf_samples scene queries per camera sample, all but the
// first from the same surface point: emitter_samples shadow rays (scene.cpp:203-207), then bsdf_samples rays
// that only count when they end on an emitter. They are fed to the same paired query the path integrator
// uses (trace2: one extension ray E + one shadow ray S sharing origin and mint): shadow rays one per pass, the
// last of them together with the first BSDF-sampled ray, the remaining BSDF-sampled rays one per pass.
// Random numbers are drawn and terms are added into `result` in the reference's order
// (all emitter samples, :137-160, then all BSDF samples, :165-195); with the default shading_samples = 1 a
// camera sample costs two passes over the geometry.
#pragma once
#include "path.h"

namespace miw {

// The BSDF sample whose ray is in flight (direct.cpp:166-176)
struct DirectPending { Spec bsdf_val; float pdf; bool delta; };

// direct.cpp:113-133: the primary hit. Returns false when the camera sample is complete (a miss).
// `Mats` / `Analytic` as in path.h: MATS_ALL kernels serve scenes with texture coordinates, bitmaps or the extended plugins,
// MATS_PLAIN the rest (the lookups compiled out); Analytic = false compiles the analytic shapes out.
template <int Mats, bool Analytic, typename Cnt>
MIW_HD bool direct_primary(const RenderParams &P, const SceneView &sc, LaneRegs &L, F4 h, V3 ray_o,
                           SurfaceInteraction &si, BsdfSide &bsdf, Cnt *cnt_local) {
    const bool valid = f2u(h.w) != MIW_MISS;
    const V3 ray_d = L.ray.d;
    int32_t emitter = -1;
    uint32_t bsdf_index = 0;
    if (valid) {
        hit_surface_interaction<Analytic, Mats == MATS_ALL>(sc, f2u(h.w), h.x, h.y, h.z, [ray_o]() { return ray_o; }, ray_d, si, bsdf_index, emitter);
        L.flags |= LF_VALID_RAY;                                  // :114
    }
    else if (sc.env) emitter = (int32_t) sc.env->emitter_index;  // scene.h:248-249
    if (!P.direct.hide_emitters && emitter >= 0)                  // :119-123
        L.res = L.res + (valid ? emitter_eval(sc.emitters[emitter], si.wi, L.wl) : env_eval_spec(*sc.env, ray_d, L.wl));
    if (!valid) return false;                                     // :125-127
    if (cnt_local) cnt_local->segments++;
    bsdf = bsdf_side(sc.bsdfs, bsdf_index, si.wi);                // :132
    L.ray.o = si.p; L.ray.mint = spawn_mint(si.p);                // every further ray leaves from here
    L.ray.d = v3(0.f); L.ray.maxt = -1.f;
    return true;
}

// One emitter sample, direct.cpp:137-160. Returns true when a shadow ray is queued in `sh`.
template <int Mats, bool Analytic, typename Cnt>
MIW_HD bool direct_emitter_sample(const RenderParams &P, const SceneView &sc, LaneRegs &L, const SurfaceInteraction &si,
                                  const BsdfSide &bsdf, ShadowOut &sh, Cnt *cnt_local) {
    const DirectRec &D = P.direct;
    DirectionSample ds;
    Spec emitter_val = sample_emitter_direction<Analytic>(sc, si.p, next_2d(L.rng), ds, L.wl);   // :141-142
    if (ds.pdf == 0.f) return false;                              // :143-145
    V3 wo = to_local(si.sh, ds.d);                                // :148
    const TexCtx tc(L.wl, si.uv, Mats == MATS_ALL ? sc.bitmaps : nullptr, Mats == MATS_ALL ? sc.bsdf_tables : nullptr);
    Spec bsdf_val = bsdf_side_eval<Mats == MATS_ALL>(bsdf, si.wi, wo, tc);   // :150
    float bsdf_pdf = bsdf_side_pdf<Mats == MATS_ALL>(bsdf, si.wi, wo, tc);   // :155
    float mis = mis_weight(ds.pdf * D.frac_lum, bsdf_pdf * D.frac_bsdf) * D.weight_lum;   // :157-158 (no delta emitters)
    Spec c = mis * bsdf_val * emitter_val;                        // :159
    if (all_zero(c)) return false;
    sh.has = true; sh.d = ds.d; sh.maxt = ds.dist * (1.f - MIW_SHADOW_EPSILON); sh.c = c;   // scene.cpp:203-205
    if (cnt_local) cnt_local->shadow_rays++;
    return true;
}

// One BSDF sample, direct.cpp:166-176. Returns true when its ray is queued in L.ray.
template <int Mats>
MIW_HD bool direct_bsdf_sample(const SceneView &sc, LaneRegs &L, const SurfaceInteraction &si, const BsdfSide &bsdf, DirectPending &pend) {
    float s1 = next_1d(L.rng);                                    // :166-167 (Clang order: next_1d, then next_2d)
    V2 s2 = next_2d(L.rng);
    BSDFSample bs;
    pend.bsdf_val = bsdf_side_sample<Mats == MATS_ALL>(bsdf, si.wi, s1, s2, bs, TexCtx(L.wl, si.uv, Mats == MATS_ALL ? sc.bitmaps : nullptr,
                                                                                   Mats == MATS_ALL ? sc.bsdf_tables : nullptr));
    if (all_zero(pend.bsdf_val)) return false;                    // :170
    pend.pdf = bs.pdf; pend.delta = (bs.sampled_type & BSDF_Delta) != 0;
    L.ray.d = to_world(si.sh, bs.wo);                             // :173-174, interaction.h:58-61
    L.ray.maxt = MIW_INFINITY;
    return true;
}

// Where a BSDF-sampled ray ended, direct.cpp:177-195
template <bool Analytic>
MIW_HD void direct_bsdf_hit(const RenderParams &P, const SceneView &sc, LaneRegs &L, F4 h, V3 ref_p,
                            const DirectPending &pend) {
    const DirectRec &D = P.direct;
    const uint32_t tri_idx = f2u(h.w);
    const bool valid = tri_idx != MIW_MISS;
    const V3 ray_d = L.ray.d;
    SurfaceInteraction sb;
    int32_t emitter = -1;
    if (valid) {
        const Tri &tr = sc.tris[tri_idx];
        const ShapeRec &shape = sc.shapes[tr.shape];
        emitter = shape.emitter;
        if (emitter < 0) return;                                  // :178-179
        if (Analytic && tr.pad) {
            const AnalyticRec &a = sc.rects[tr.pad - 1u];
            if (a.kind == ANALYTIC_SPHERE) compute_surface_interaction_sphere(a, h.x, ref_p, ray_d, sb);
            else compute_surface_interaction_rect(a, h.x, h.y, h.z, ref_p, ray_d, sb);
        } else {
            const float *vn = (shape.flags & 1u) ? sc.tri_vn + 9 * (size_t) tri_idx : nullptr;
            const float *tc = (shape.flags & SHAPE_HAS_TEXCOORDS) ? sc.tri_uv + 6 * (size_t) tr.prim : nullptr;
            compute_surface_interaction(ld3(tr.p0), ld3(tr.p1), ld3(tr.p2), vn, tc, h.x, h.y, h.z, ray_d, sb);
        }
    }
    else if (sc.env) emitter = (int32_t) sc.env->emitter_index;
    if (emitter < 0) return;
    Spec emitter_val = valid ? emitter_eval(sc.emitters[emitter], sb.wi, L.wl) : env_eval_spec(*sc.env, ray_d, L.wl);   // :182
    float emitter_pdf = 0.f;                                      // :187-191
    if (!pend.delta) {
        // DirectionSample3f ds(si_bsdf, si), records.h:167-173 (d = -wi = ray.d for a miss)
        V3 d = ray_d; float dist = 0.f; V3 n = v3(0.f);
        if (valid) {
            d = sb.p - ref_p;
            dist = norm(d);
            d = d / dist;
            n = sb.sh.n;
        }
        emitter_pdf = pdf_emitter_direction<Analytic>(sc, (uint32_t) emitter, d, dist, n, ref_p);
    }
    L.res = L.res + pend.bsdf_val * emitter_val * mis_weight(pend.pdf * D.frac_bsdf, emitter_pdf * D.frac_lum) * D.weight_bsdf;   // :193-196
}

// The sample loop of one pixel stream (cf. pixel_stream_render in path.h: same `work` / `trace2` contract).
//
// Two query sites: the camera ray of every lane in one, the rays that leave the surface point in the other (an inner loop, one trip
// with the default shading_samples = 1). The lanes of a wavefront stay on the same camera sample, so the rays of the first query
// are neighbours and the second query sees only lanes that have something to ask (a lane whose camera ray left the scene sits it
// out); the surface interaction and the BSDF are built after the first query and used up before the second — what crosses the
// second query is the hit record of the camera ray (4 + 6 registers), from which hit_surface_interaction() rebuilds them, bit for
// bit, in the rare case that further samples are due from the same point. Against the earlier form (one query site shared by
// camera and secondary rays, the interaction carried across it) the path kernel of a frame measured 92.7 ms instead of 135.8 on
// the Cornell box, 207 / 235 on the material balls, 174 / 188 on the 0.9 M-triangle interior (round-3 session I). The kernels
// still spill at three wavefronts per SIMD; see MIW_DIRECT_WAVES in device/resident_kernel.h for why they stay there.
template <int Mats = MATS_ALL, bool Analytic = true, typename Work, typename Trace2, typename Cnt>
MIW_HD void pixel_stream_render_direct(const RenderParams &P, const SceneView &sc, uint32_t sample_end, Work &work,
                                       Trace2 trace2, Cnt *cnt_local) {
    const uint32_t n_emitter = P.direct.emitter_samples, n_bsdf = P.direct.bsdf_samples;
    LaneRegs L;
    L.flags = LF_DONE; L.sample_idx = 0; L.rng.state = 0; L.rng.inc = MIW_PCG32_SCALAR_INC;
    uint32_t pixel = 0;
    bool have = false;
    auto sink = [&work](uint32_t px, uint32_t sample_idx, V2 pos, const float *aovs) { work.put(px, sample_idx, pos, aovs); };
    for (;;) {
        if (L.flags & LF_DONE) {
            if (have) {
                U4 st; st.x = (uint32_t) L.rng.state; st.y = (uint32_t) (L.rng.state >> 32);
                st.z = L.sample_idx >= P.spp ? (uint32_t) LF_DONE : 0u; st.w = L.sample_idx;
                work.store(st);
            }
            U4 st;
            have = work.fetch(pixel, st);
            if (!have) { if (work.exhausted()) break; continue; }    // (chunk jobs, path.h: pixel_stream_render)
            L.rng.state = (uint64_t) st.x | ((uint64_t) st.y << 32);
            L.sample_idx = st.w; L.flags = 0;
            lane_begin_sample(P, pixel, L, sample_end);
            continue;
        }
        const V3 o0 = L.ray.o, d0 = L.ray.d;
        F4 h0; bool unused = false;
        trace2(o0, L.ray.mint, d0, L.ray.maxt, true, v3(0.f), -1.f, false, h0, unused);
        {
            SurfaceInteraction si; BsdfSide bsdf;
            bsdf.b = sc.bsdfs; bsdf.flip = bsdf.none = false; bsdf.flags = 0;
            if (direct_primary<Mats, Analytic>(P, sc, L, h0, o0, si, bsdf, cnt_local)) {
                uint32_t ie = (bsdf.flags & BSDF_Smooth) ? 0u : n_emitter, ib = 0;   // :134-136: no emitter samples (and no draws) otherwise
                for (;;) {
                    ShadowOut sh; sh.has = false; sh.d = v3(0.f); sh.maxt = -1.f; sh.c = spec(0.f);
                    DirectPending pend; pend.bsdf_val = spec(0.f); pend.pdf = 0.f; pend.delta = false;
                    bool queued = false;
                    while (ie < n_emitter && !queued) { ++ie; queued = direct_emitter_sample<Mats, Analytic>(P, sc, L, si, bsdf, sh, cnt_local); }
                    if (ie == n_emitter)                           // the last shadow ray travels with the first BSDF-sampled ray
                        while (ib < n_bsdf && !(L.ray.maxt >= 0.f)) { ++ib; if (direct_bsdf_sample<Mats>(sc, L, si, bsdf, pend)) queued = true; }
                    if (!queued) break;
                    const V3 o = L.ray.o;
                    const bool has_e = L.ray.maxt >= 0.f;
                    F4 h; bool occluded = false;
                    trace2(o, L.ray.mint, L.ray.d, L.ray.maxt, has_e, sh.d, sh.maxt, sh.has, h, occluded);
                    if (sh.has && !occluded) L.res = L.res + sh.c;     // direct.cpp:159 of the emitter sample that was in flight
                    if (has_e) {
                        direct_bsdf_hit<Analytic>(P, sc, L, h, o, pend);
                        L.ray.d = v3(0.f); L.ray.maxt = -1.f;
                    }
                    if (ie == n_emitter && ib == n_bsdf) break;
                    int32_t emitter; uint32_t bsdf_index;              // more samples from this point: the interaction again
                    hit_surface_interaction<Analytic, Mats == MATS_ALL>(sc, f2u(h0.w), h0.x, h0.y, h0.z, [o0]() { return o0; }, d0, si, bsdf_index, emitter);
                    bsdf = bsdf_side(sc.bsdfs, bsdf_index, si.wi);
                }
            }
        }
        lane_finish_sample(P, pixel, L, sink);
        if (cnt_local) cnt_local->samples++;
        L.flags = 0;
        lane_begin_sample(P, pixel, L, work.job_end(L.sample_idx, sample_end));
    }
}

} // namespace miw
