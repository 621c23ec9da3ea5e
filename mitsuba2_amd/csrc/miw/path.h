// The path integrator re-expressed as per-lane wavefront stages.
//
// One lane = one pixel. scalar_rgb seeds one PCG32 per pixel and runs that
// pixel's spp samples back to back on the same stream
// (src/librender/integrator.cpp:196-209, src/librender/sampler.cpp:30-34), so
// the sample loop of a pixel is serial; parallelism is across pixels. A lane
// that finishes sample j regenerates sample j+1 in place.
//
// PathIntegrator::sample (src/integrators/path.cpp:100-211) is cut at its two
// Scene::ray_intersect / ray_test calls into:
//   lane_generate : render_sample prologue (integrator.cpp:233-261) -> primary ray
//   [trace closest]                            -> hit queue
//   lane_shade    : one iteration of the depth loop (path.cpp:124-208) -> shadow
//                   ray + pending contribution, extension ray; on termination
//                   the render_sample epilogue (integrator.cpp:264-287): splat,
//                   advance, regenerate.
//   [trace any]   : shadow visibility (scene.cpp:203-207)
// The pending emitter-sampling contribution of iteration k is added at the top
// of lane_shade in iteration k+1, before that iteration's emission term — the
// same order of float additions into `result` as the reference.
//
// Queue layout in HBM: structure-of-16-byte-fields, one array per field group,
// indexed by lane; every access below is one aligned dwordx4 (or dwordx2) per
// lane, contiguous across a wavefront.
#pragma once
#include "base.h"
#include "rng.h"
#include "scene.h"
#include "film.h"

namespace miw {

// lane flag word (st.z)
enum : uint32_t {
    LF_DEPTH_MASK  = 0x0fffu,
    LF_VALID_RAY   = 1u << 12,   // first intersection was valid (alpha, path.cpp:121)
    LF_PREV_DELTA  = 1u << 13,   // last BSDF sample was a delta lobe
    LF_HAS_SHADOW  = 1u << 14,   // shadow ray + pending contribution queued
    LF_RAY_ACTIVE  = 1u << 15,   // extension / primary ray queued
    LF_DONE        = 1u << 16,   // all spp of this pixel finished
    LF_DEAD_PENDING= 1u << 17,   // path over, waiting for its last shadow ray
};

struct LaneQueues {
    // path state (48 B/lane + 8 B position sample + 4 B static pixel id)
    F4 *tp;        // throughput.rgb, eta
    F4 *res;       // result.rgb, prev bsdf pdf
    U4 *st;        // rng state lo, hi, flags|depth, sample index
    F2 *pos;       // position_sample of the sample in flight
    const uint32_t *pixel;  // x | y << 16 (film coordinates incl. crop offset)
    // extension / primary ray queue (32 B/lane)
    F4 *ray_o;     // o.xyz, mint
    F4 *ray_d;     // d.xyz, maxt
    // hit queue (16 B/lane)
    F4 *hit;       // t, u, v, bits(tri index in leaf order | MIW_MISS)
    // shadow queue (32 B + 4 B/lane); origin and mint are shared with ray_o
    F4 *sh_d;      // d.xyz, maxt
    F4 *sh_c;      // pending contribution rgb
    uint32_t *sh_vis; // 1 = unoccluded (written by the any-hit trace)
    // finished-sample log, [lane][sample j]: what ImageBlock::put received. Two formats, chosen per render by the host:
    //   log_rec != nullptr   16 B/sample: X, Y, Z, phase class x | class y << 8 | alpha << 16 (film.h: phase classes) —
    //                        filters the class enumeration covers (box, tent, gaussian, mitchell, catmullrom)
    //   else                 24 B/sample: position_sample (x = NaN: sample rejected by imageblock.cpp:85-109) + X, Y, Z, alpha
    F2 *log_pos;
    F4 *log_val;   // X, Y, Z, alpha   (weight channel W is the constant 1)
    U4 *log_rec;
    const float *log_thr;   // the 256 phase thresholds (global memory; kernels that stage them in LDS pass their own pointer)
    uint32_t log_rej;       // bits 0-7: class index of a rejected sample = the class count (the first all-zero row of the weight table); bits 8+: the log's layout (log_index's il: 0 = [lane][sample j])
    // placed pixel queues of small shards (device/resident_kernel.h: QueueWork): what every pixel cost (written by the measuring
    // launch, nullptr otherwise), the lanes sorted by that cost (64 consecutive entries = one piece: pixels of about equal cost
    // share a wavefront), four pieces per SIMD queue, the SIMD registry (word 0: "all dry" flag; then by hardware id)
    // piece k's j-th lane is lane_sorted[k * piece_a + j * piece_b]: (64, 1) — a piece is 64 consecutive sorted lanes, its pixels cost
    // about the same and finish together — or (1, pieces) — a piece takes every pieces-th sorted lane, one pixel of every cost
    // stratum (which cut a launch gets: mi_render, by measurement)
    uint32_t *lane_cost; const uint32_t *lane_sorted; const uint32_t *piece_list; uint32_t *simd_ids; uint32_t piece_a, piece_b;
    // film replay beside the render (round 6; full frames over the tile-interleaved log, device/resident_kernel.h: QueueWork::store): per group of
    // 64 tiles, how many pixels have run all their samples (group_done, device memory) against how many there are (group_expected); the pixel
    // that completes a group raises its flag (group_flag: host-coherent memory the replay's stream waits on). nullptr: off
    uint32_t *group_done = nullptr; const uint32_t *group_expected = nullptr; uint32_t *group_flag = nullptr; uint32_t group_shift = 0;
    // pixel jobs cut into CHUNKS of samples (round 6; device/resident_kernel.h: QueueWork::fetch_job, full frames, all samples in one launch): the queue holds
    // (chunk, pixel slot) pairs, chunk-major — job_total = slots x chunks. The chunks HALVE: a chunk ends where the number of samples still to do is a power of two
    // >= job_min (512 spp, job_min 64: 256 + 128 + 64 + 64 — half of a pixel's work in its first job, the launch's tail as long as the last), so chunk j > 0 starts
    // at sample spp - (job_pow >> (j - 1)), job_pow = the largest power of two below spp; a chunk is ready once the slot's state word stands at its first sample.
    // job_chunk = 0: a job = all the launch's samples of a pixel (every other launch); job_min = 2^31 then: no sample count reaches it.
    uint32_t job_chunk = 0, job_pow = 0, job_total = 0, job_min = 0x80000000u;
};

// Which SamplingIntegrator::sample runs per camera sample, and the direct integrator's constants (direct.cpp:78-104)
enum : uint32_t { INTEG_PATH = 0, INTEG_DIRECT = 1 };
struct DirectRec {
    uint32_t emitter_samples, bsdf_samples, hide_emitters;
    float frac_bsdf, frac_lum, weight_bsdf, weight_lum;
};

struct RenderParams {
    SensorRec sensor;
    FilmRec film;
    uint32_t spp;
    int32_t max_depth, rr_depth;
    uint32_t n_lanes;
    uint32_t integrator;             // INTEG_*
    DirectRec direct;
    uint32_t moment_pass;            // moment.cpp around the integrator: 0 off, 1 values, 2 squares (include/miwave.h)
    // derived by render_params_prepare(): what every camera sample needs of the records above, as floats — kernel arguments live in
    // scalar registers, the same values computed in the kernel (int -> float conversions, the fmas of the camera origin) are
    // loop-invariant VECTOR registers the packet kernel has none to spare for (it spilled twelve)
    float cam_o[3];                  // sensor_origin(sensor)
    float crop_f[4];                 // (float) film.crop_x, crop_y, crop_w, crop_h
};
MIW_HD void render_params_prepare(RenderParams &P) {
    const V3 o = sensor_origin(P.sensor);
    P.cam_o[0] = o.x; P.cam_o[1] = o.y; P.cam_o[2] = o.z;
    P.crop_f[0] = (float) P.film.crop_x; P.crop_f[1] = (float) P.film.crop_y; P.crop_f[2] = (float) P.film.crop_w; P.crop_f[3] = (float) P.film.crop_h;
}

// per-launch device counters (mi_get_counters)
struct Counters {
    unsigned long long segments;   // iterations of the depth loop that reached shading
    unsigned long long samples;    // finished camera samples
    unsigned long long shadow_rays;
    unsigned long long active_lanes; // lanes not DONE after the last shade pass
};
// what one lane counts during one launch (the kernels sum these per wavefront into a Counters shard at the end): 32 bits are
// plenty for one launch and half the registers — they are live through the whole render loop
struct LaneCounters { uint32_t segments, samples, shadow_rays; };

MIW_HD float next_1d(PCG32 &r) { return pcg32_next_f32(r); }
MIW_HD V2 next_2d(PCG32 &r) { float a = pcg32_next_f32(r); float b = pcg32_next_f32(r); return v2(a, b); }

// integrator.cpp:233-261 — start sample `sample_idx` of this lane's pixel, or
// retire the lane. Writes ray + fresh path state into the caller's registers.
struct LaneRegs {
    PCG32 rng;
    uint32_t flags, sample_idx;
    Spec tp, res; float eta, prev_pdf;
    V2 pos;
    Ray ray;
    Wavelengths wl; Spec ray_weight;     // sample_wavelength (spectrum.h:305-314); unused words in RGB builds
};

MIW_HD void lane_begin_sample(const RenderParams &P, uint32_t pixel, LaneRegs &L, uint32_t sample_end) {
    if (L.sample_idx >= sample_end) {                    // pixel finished (or end of this pass): retire the lane
        L.flags = LF_DONE;
        L.ray.o = L.ray.d = v3(0.f); L.ray.mint = 0.f; L.ray.maxt = -1.f;   // maxt < 0: no ray queued
        return;
    }
    float px = (float) (pixel & 0xffffu), py = (float) (pixel >> 16);
    V2 j = next_2d(L.rng);                               // :242
    L.pos = v2(px + j.x, py + j.y);
    const float wavelength_sample = next_1d(L.rng);     // :252 wavelength sample, always drawn
#if MIW_SPECTRAL
    sample_wavelengths(wavelength_sample, L.wl, L.ray_weight);   // perspective.cpp:191-192
#else
    (void) wavelength_sample; L.ray_weight = spec(1.f);
#endif
    V2 adj = v2((L.pos.x - P.crop_f[0]) / P.crop_f[2],    // :254-256
                (L.pos.y - P.crop_f[1]) / P.crop_f[3]);
    L.ray = sensor_sample_ray(P.sensor, adj, v3(P.cam_o[0], P.cam_o[1], P.cam_o[2]));   // :258
    L.tp = spec(1.f); L.res = spec(0.f); L.eta = 1.f; L.prev_pdf = 0.f;   // path.cpp:111-116
    L.flags = 1u | LF_RAY_ACTIVE;                        // depth = 1
}

// integrator.cpp:264-287 — convert, splat, advance
// `sink(pixel, sample_idx, position_sample, aovs)` stands for block->put(position_sample, aovs), :285
template <typename Sink>
MIW_HD void lane_finish_sample(const RenderParams &P, uint32_t pixel, LaneRegs &L, Sink sink) {
#if MIW_SPECTRAL
    V3 xyz = spectrum_to_xyz(L.ray_weight * L.res, L.wl);   // :266-271
#else
    V3 xyz = srgb_to_xyz(L.res);                         // :272-273 (ray_weight == 1 in RGB)
#endif
    float aovs[5] = { xyz.x, xyz.y, xyz.z, (L.flags & LF_VALID_RAY) ? 1.f : 0.f, 1.f };
    if (P.moment_pass) {                                 // moment.cpp:83-88: nested.XYZ and their squares ride along
        const float sq[3] = { sqr(xyz.x), sqr(xyz.y), sqr(xyz.z) };
        bool ok = true;                                  // ImageBlock::put tests all channels of the sample together; the block of an
        for (int k = 0; k < 3; ++k) ok = ok && isfinite_(aovs[k]) && isfinite_(sq[k]);   // integrator with AOVs has warn_negative = false (integrator.cpp:113)
        if (P.moment_pass == 2u) { aovs[0] = sq[0]; aovs[1] = sq[1]; aovs[2] = sq[2]; }
        if (!ok) aovs[0] = __builtin_nanf("");
    }
    sink(pixel, L.sample_idx, L.pos, aovs);
    L.sample_idx++;                                      // :287
}

// Sink 1: splat immediately into float64 accumulators (atomics on the device)
template <typename Add> struct SplatSink {
    const FilmRec *film; Add add;
    MIW_HD void operator()(uint32_t pixel, uint32_t, V2 pos, const float *aovs) const {
        film_splat(*film, (int) (pixel & 0xffffu), (int) (pixel >> 16), pos, aovs, add);
    }
};
// Sink 1b: splat through add_xy(fx, fy, channel, value) (workgroup-local film tile on the device)
template <typename AddXY> struct SplatXYSink {
    const FilmRec *film; AddXY add;
    MIW_HD void operator()(uint32_t pixel, uint32_t, V2 pos, const float *aovs) const {
        film_splat_xy(*film, (int) (pixel & 0xffffu), (int) (pixel >> 16), pos, aovs, add);
    }
};
// Sink 2: append to the lane's sample log; the film is assembled afterwards by the
// ordered gather (miw/film_gather.h), in the reference's float32 accumulation order.
struct LogSink {
    F2 *log_pos; F4 *log_val; uint32_t lane, spp;        // [lane][sample]: one contiguous run per pixel
    uint32_t warn_negative;                              // FilmRec::warn_negative
    MIW_HD void operator()(uint32_t, uint32_t sample_idx, V2 pos, const float *aovs) const {
        size_t i = (size_t) lane * spp + sample_idx;
        F2 p; p.x = sample_is_valid(aovs, warn_negative != 0) ? pos.x : __builtin_nanf(""); p.y = pos.y;
        F4 v; v.x = aovs[0]; v.y = aovs[1]; v.z = aovs[2]; v.w = aovs[3];
        log_pos[i] = p; log_val[i] = v;
    }
};

#ifndef MIW_LOG_NT
#define MIW_LOG_NT 1              /* the 16-byte log records are written with streaming (nontemporal) stores: 35 GB instead of 48 GB of HBM writes per C2 frame (profiles/r03), same kernel time; 0: plain stores */
#endif
// Sink 3: the 16-byte record (film.h: phase classes). `thr` = the 256 thresholds, wherever the caller keeps them (LDS on the
// device's resident kernels). A rejected sample is logged as class `rej` (= the class count: the first all-zero row of the
// weight table) with zero values, so that the replay needs no branch for it.
template <typename Thr>
struct LogSink16 {
    U4 *log_rec; Thr thr; const FilmRec *film; uint32_t lane, spp, rej, il;
    MIW_HD void operator()(uint32_t pixel, uint32_t sample_idx, V2 pos, const float *aovs) const {
        const size_t i = log_index(il, lane, spp, sample_idx);
        U4 r; r.x = r.y = r.z = 0u; r.w = film_pack_meta(rej, rej, false);
        if (sample_is_valid(aovs, film->warn_negative != 0)) {
            const uint32_t cx = film_class_of(thr, film_phase(*film, pos.x, (int) (pixel & 0xffffu), film->crop_x)),
                           cy = film_class_of(thr, film_phase(*film, pos.y, (int) (pixel >> 16), film->crop_y));
            r.x = f2u(aovs[0]); r.y = f2u(aovs[1]); r.z = f2u(aovs[2]); r.w = film_pack_meta(cx, cy, aovs[3] != 0.f);
        }
#if defined(__HIP_DEVICE_COMPILE__) && MIW_LOG_NT
        // streaming store: the record is written once and read much later by the film replay; keeping the half-written 32-byte
        // sector out of L2's write-back path makes the store cost its own sector and no more (profiles/r03: WRITE_SIZE per sample)
        typedef uint32_t u4v_ __attribute__((ext_vector_type(4)));
        u4v_ v_; v_.x = r.x; v_.y = r.y; v_.z = r.z; v_.w = r.w;
        __builtin_nontemporal_store(v_, reinterpret_cast<u4v_ *>(log_rec + i));
#else
        log_rec[i] = r;
#endif
    }
};

MIW_HD void lane_load(const LaneQueues &Q, uint32_t lane, LaneRegs &L) {
    U4 st = Q.st[lane];
    L.rng.state = (uint64_t) st.x | ((uint64_t) st.y << 32);
    L.rng.inc = MIW_PCG32_SCALAR_INC;
    L.flags = st.z; L.sample_idx = st.w;
}
#if !MIW_SPECTRAL   // the HBM-queue plan carries RGB path state (16-byte fields); spectral builds run the resident plan
MIW_HD void lane_load_path(const LaneQueues &Q, uint32_t lane, LaneRegs &L) {
    F4 a = Q.tp[lane], b = Q.res[lane];
    L.tp = v3(a.x, a.y, a.z); L.eta = a.w;
    L.res = v3(b.x, b.y, b.z); L.prev_pdf = b.w;
    L.pos = v2(0.f, 0.f);            // position_sample is fetched only when the sample is splatted
}
MIW_HD void lane_store(const LaneQueues &Q, uint32_t lane, const LaneRegs &L, bool store_pos) {
    U4 st; st.x = (uint32_t) L.rng.state; st.y = (uint32_t) (L.rng.state >> 32);
    st.z = L.flags; st.w = L.sample_idx;
    Q.st[lane] = st;
    F4 a; a.x = L.tp.x; a.y = L.tp.y; a.z = L.tp.z; a.w = L.eta; Q.tp[lane] = a;
    F4 b; b.x = L.res.x; b.y = L.res.y; b.z = L.res.z; b.w = L.prev_pdf; Q.res[lane] = b;
    if (store_pos) { F2 p; p.x = L.pos.x; p.y = L.pos.y; Q.pos[lane] = p; }
    F4 o; o.x = L.ray.o.x; o.y = L.ray.o.y; o.z = L.ray.o.z; o.w = L.ray.mint; Q.ray_o[lane] = o;
    F4 d; d.x = L.ray.d.x; d.y = L.ray.d.y; d.z = L.ray.d.z; d.w = L.ray.maxt; Q.ray_d[lane] = d;
}
MIW_HD void lane_clear_shadow(const LaneQueues &Q, uint32_t lane) {
    F4 dead; dead.x = dead.y = dead.z = 0.f; dead.w = -1.f;   // maxt < 0: no shadow ray queued
    Q.sh_d[lane] = dead;
}

// Stage 0: seed every lane (sampler.cpp:83-96 via integrator.cpp:198) and
// start its first sample. `seed` = base_seed + block_id * block_size^2 + morton_i.
MIW_HD void lane_init(const RenderParams &P, const LaneQueues &Q, uint32_t lane, uint32_t pixel, uint64_t seed) {
    LaneRegs L;
    pcg32_seed(L.rng, seed, MIW_PCG32_DEFAULT_STREAM);
    L.sample_idx = 0; L.flags = 0;
    L.tp = v3(1.f); L.res = v3(0.f); L.eta = 1.f; L.prev_pdf = 0.f; L.pos = v2(0.f, 0.f);
    L.ray.o = L.ray.d = v3(0.f); L.ray.mint = 0.f; L.ray.maxt = -1.f;
    lane_begin_sample(P, pixel, L, P.spp);
    lane_store(Q, lane, L, true);
    lane_clear_shadow(Q, lane);
    Q.sh_vis[lane] = 0;
}
// A lane of the launch grid that maps to no pixel (clipped edge block, integrator.cpp:201-202)
MIW_HD void lane_init_unused(const LaneQueues &Q, uint32_t lane) {
    LaneRegs L;
    L.rng.state = 0; L.rng.inc = MIW_PCG32_SCALAR_INC; L.sample_idx = 0; L.flags = LF_DONE;
    L.tp = v3(0.f); L.res = v3(0.f); L.eta = 0.f; L.prev_pdf = 0.f; L.pos = v2(0.f, 0.f);
    L.ray.o = L.ray.d = v3(0.f); L.ray.mint = 0.f; L.ray.maxt = -1.f;
    lane_store(Q, lane, L, true);
    lane_clear_shadow(Q, lane);
    Q.sh_vis[lane] = 0;
}
#endif   // !MIW_SPECTRAL

#ifndef MIW_SECTION
#define MIW_SECTION(i) do { } while (0)      /* section clock of debug builds (miwave.hip) */
#endif

// What one depth-loop iteration hands to the shadow stage (scene.cpp:203-207):
// origin and mint are the extension ray's (L.ray.o, L.ray.mint).
struct ShadowOut { bool has; V3 d; float maxt; Spec c; };

enum { STEP_CONTINUE = 0,        // extension ray queued in L.ray
       STEP_FINISHED = 1,        // the camera sample is complete
       STEP_DEAD_PENDING = 2 };  // complete once the shadow ray in `sh` is resolved

// One iteration of PathIntegrator::sample's depth loop (path.cpp:124-208) on register
// state. `L.ray.d` is the direction of the ray that produced hit record `h`;
// `prev_o()` returns its origin (the previous vertex; fetched lazily — only the
// emitter-hit MIS term needs it). Leaves the extension ray in L.ray (maxt < 0: none).
// Shared by every execution plan: the HBM-queue wavefront kernels (lane_shade below),
// the register-resident kernel (k_path_resident) and the CPU checker.
// `Mats` names the BSDF plugins the scene uses, so that a kernel compiled for MATS_DIFFUSE (every shape one-sided
// smooth diffuse: BASELINE config 2) carries no dispatch and none of the other plugins' code; MATS_ALL is the table;
// MATS_PLAIN is the table for scenes without texture coordinates and bitmap textures (the lookups compiled out).
// MATS_TRIO is MATS_PLAIN for scenes whose records are all diffuse / dielectric / roughconductor (BASELINE configs 3, 4).
enum { MATS_ALL = 0, MATS_DIFFUSE = 1, MATS_PLAIN = 2, MATS_TRIO = 3 };
// `Analytic` = false compiles the analytic-shape branch out (scenes the caller knows to be triangles only).
template <int Mats = MATS_ALL, bool Analytic = true, typename PrevO, typename Cnt>
MIW_HD int path_step(const RenderParams &P, const SceneView &sc, LaneRegs &L, F4 h, PrevO prev_o,
                     ShadowOut &sh, Cnt *cnt_local) {
    sh.has = false;
    const uint32_t depth = L.flags & LF_DEPTH_MASK;
    const uint32_t tri_idx = f2u(h.w);
    const bool valid = tri_idx != MIW_MISS;
    const V3 ray_d = L.ray.d;

    SurfaceInteraction si;
    uint32_t bsdf_index = 0;
    int32_t emitter = -1;                            // scene.h:243-253 (no environment emitter)
    // only MATS_ALL kernels are launched for scenes with texture coordinates (miwave.hip: diffuse_only / textured)
    if (valid) hit_surface_interaction<Analytic, Mats == MATS_ALL>(sc, tri_idx, h.x, h.y, h.z, prev_o, ray_d, si, bsdf_index, emitter);
    else if (sc.env) emitter = (int32_t) sc.env->emitter_index;   // a miss sees the environment, scene.h:248-249
    if (depth == 1 && valid) L.flags |= LF_VALID_RAY;   // path.cpp:121
    MIW_SECTION(7);                                   // (sections 6.. : the shade body of the phase machine, debug builds)

    // ---- intersection with emitters, path.cpp:126-129 ----
    if (emitter >= 0) {
        float emission_weight = 1.f;                 // :109
        if (depth > 1) {                             // :194-205, evaluated lazily
            float emitter_pdf = 0.f;
            if (!(L.flags & LF_PREV_DELTA)) {
                // DirectionSample3f ds(si_bsdf, si), records.h:167-173 (d = -wi = ray.d for a miss)
                V3 d = ray_d; float dist = 0.f; V3 n = v3(0.f), ref_p = v3(0.f);
                if (valid) {
                    ref_p = prev_o();
                    d = si.p - ref_p;
                    dist = norm(d);
                    d = d / dist;
                    n = si.sh.n;
                }
                emitter_pdf = pdf_emitter_direction<Analytic>(sc, (uint32_t) emitter, d, dist, n, ref_p);
            }
            emission_weight = mis_weight(L.prev_pdf, emitter_pdf);
        }
        Spec radiance = valid ? emitter_eval(sc.emitters[emitter], si.wi, L.wl) : env_eval_spec(*sc.env, ray_d, L.wl);
        L.res = L.res + emission_weight * L.tp * radiance;
    }

    MIW_SECTION(8);
    bool active = valid;                             // :131

    // ---- Russian roulette, :137-141 (the draw happens even for dead paths) ----
    if ((int32_t) depth > P.rr_depth) {
        float q = min_(hmax(L.tp) * sqr(L.eta), .95f);
        active = (next_1d(L.rng) < q) && active;
        L.tp = L.tp * rcp(q);
    }

    // ---- termination, :147-149 ----
    if (depth >= (uint32_t) P.max_depth || !active) return STEP_FINISHED;

    if (cnt_local) cnt_local->segments++;
    BsdfSide bsdf;                                                   // si.bsdf(ray), incl. the twosided adapter
    if (Mats == MATS_DIFFUSE) { bsdf.b = sc.bsdfs + bsdf_index; bsdf.flip = bsdf.none = false; bsdf.flags = BSDF_DiffuseReflection; }
    else bsdf = bsdf_side(sc.bsdfs, bsdf_index, si.wi);
    const uint32_t bflags = bsdf.flags;
    L.ray.o = si.p; L.ray.mint = spawn_mint(si.p);   // shared by shadow + extension ray
    L.ray.d = v3(0.f); L.ray.maxt = -1.f;

    // what the plugin's texture lookups see of `si`; only MATS_ALL kernels are launched for scenes with bitmap textures
    const TexCtx tc(L.wl, si.uv, Mats == MATS_ALL ? sc.bitmaps : nullptr, Mats == MATS_ALL ? sc.bsdf_tables : nullptr);
    constexpr bool Ext = Mats == MATS_ALL;           // plugins only "extended" scenes contain (roughplastic)
    constexpr bool Trio = Mats == MATS_TRIO;

    // ---- emitter sampling, :155-172 ----
    if (bflags & BSDF_Smooth) {
        DirectionSample ds;
        Spec emitter_val = sample_emitter_direction<Analytic>(sc, si.p, next_2d(L.rng), ds, L.wl);
        if (ds.pdf != 0.f) {
            V3 wo = to_local(si.sh, ds.d);
            Spec bsdf_val = Mats == MATS_DIFFUSE ? diffuse_eval(*bsdf.b, si.wi, wo, tc) : bsdf_side_eval<Ext, Trio>(bsdf, si.wi, wo, tc);
            float bpdf = Mats == MATS_DIFFUSE ? diffuse_pdf(si.wi, wo) : bsdf_side_pdf<Ext, Trio>(bsdf, si.wi, wo, tc);
            float mis = mis_weight(ds.pdf, bpdf);
            Spec c = mis * L.tp * bsdf_val * emitter_val;
            if (!all_zero(c)) {
                // shadow ray, scene.cpp:203-205
                sh.has = true; sh.d = ds.d; sh.maxt = ds.dist * (1.f - MIW_SHADOW_EPSILON); sh.c = c;
                if (cnt_local) cnt_local->shadow_rays++;
            }
        }
    }

    MIW_SECTION(9);
    // ---- BSDF sampling, :177-186 (Clang order: next_1d, then next_2d) ----
    float s1 = next_1d(L.rng);
    V2 s2 = next_2d(L.rng);
    BSDFSample bs;
    Spec bsdf_val = Mats == MATS_DIFFUSE ? diffuse_sample(*bsdf.b, si.wi, s2, bs, tc) : bsdf_side_sample<Ext, Trio>(bsdf, si.wi, s1, s2, bs, tc);
    L.tp = L.tp * bsdf_val;
    if (all_zero(L.tp))                              // :182-184
        return sh.has ? STEP_DEAD_PENDING : STEP_FINISHED;
    L.eta *= bs.eta;                                 // :186
    L.ray.d = to_world(si.sh, bs.wo);                // :189, interaction.h:58-61
    L.ray.maxt = MIW_INFINITY;
    L.prev_pdf = bs.pdf;
    L.flags = (L.flags & ~(LF_DEPTH_MASK | LF_PREV_DELTA)) | ((depth + 1) & LF_DEPTH_MASK)
            | ((bs.sampled_type & BSDF_Delta) ? LF_PREV_DELTA : 0u) | LF_RAY_ACTIVE;
    MIW_SECTION(10);
    return STEP_CONTINUE;
}

#if !MIW_SPECTRAL
// Stage 2 of the HBM-queue plan: one iteration of the depth loop for one lane.
// Returns the lane's new flag word: LF_DONE clear = the lane still has work; LF_RAY_ACTIVE = an
// extension / primary ray is queued; LF_HAS_SHADOW = a shadow ray is queued; LF_DEAD_PENDING = the
// sample only waits for that shadow ray (the device builds the next iteration's work lists from these).
template <typename Sink, typename Cnt>
MIW_HD uint32_t lane_shade(const RenderParams &P, const SceneView &sc, const LaneQueues &Q,
                           uint32_t lane, Cnt *cnt_local, Sink sink) {
    LaneRegs L;
    lane_load(Q, lane, L);
    if (L.flags & LF_DONE) return LF_DONE;
    lane_load_path(Q, lane, L);
    const bool had_shadow = (L.flags & LF_HAS_SHADOW) != 0;

    // (k-1)'s emitter-sampling contribution, now that its shadow ray is resolved
    // (path.cpp:171 with spec[ray_test] = 0 from scene.cpp:206)
    if (had_shadow) {
        if (Q.sh_vis[lane]) { F4 c = Q.sh_c[lane]; L.res = L.res + v3(c.x, c.y, c.z); }
        L.flags &= ~LF_HAS_SHADOW;
    }

    bool finished = false;        // the camera sample in flight is complete

    if (L.flags & LF_DEAD_PENDING) {
        finished = true;
    } else {
        F4 rd = Q.ray_d[lane];
        L.ray.d = v3(rd.x, rd.y, rd.z);
        ShadowOut sh;
        const int r = path_step(P, sc, L, Q.hit[lane],
                                [&]() { F4 ro = Q.ray_o[lane]; return v3(ro.x, ro.y, ro.z); }, sh, cnt_local);
        if (sh.has) {
            F4 sd; sd.x = sh.d.x; sd.y = sh.d.y; sd.z = sh.d.z; sd.w = sh.maxt;
            Q.sh_d[lane] = sd;
            F4 sc4; sc4.x = sh.c.x; sc4.y = sh.c.y; sc4.z = sh.c.z; sc4.w = 0.f;
            Q.sh_c[lane] = sc4;
            L.flags |= LF_HAS_SHADOW;
        }
        if (r == STEP_FINISHED) finished = true;
        else if (r == STEP_DEAD_PENDING)             // the sample ends once its last shadow ray is resolved
            L.flags = (L.flags & ~LF_RAY_ACTIVE) | LF_DEAD_PENDING;
    }

    bool store_pos = false;
    if (finished) {
        // ---- splat, advance, regenerate (integrator.cpp:264-287) ----
        const uint32_t pixel = Q.pixel[lane];
        F2 ps = Q.pos[lane]; L.pos = v2(ps.x, ps.y);
        lane_finish_sample(P, pixel, L, sink);
        if (cnt_local) cnt_local->samples++;
        L.flags = 0;
        lane_begin_sample(P, pixel, L, P.spp);
        store_pos = true;
    }
    lane_store(Q, lane, L, store_pos);
    if (had_shadow && !(L.flags & LF_HAS_SHADOW)) lane_clear_shadow(Q, lane);
    return L.flags;
}
#endif   // !MIW_SPECTRAL

// The register-resident plan: a run of one pixel's sample loop (render_block's inner
// loops, integrator.cpp:196-209) with every Scene::ray_intersect / ray_test call made
// in place. The two scene queries of one depth-loop iteration are issued TOGETHER, one
// iteration late for the shadow ray: the shadow ray of vertex k (scene.cpp:203-207) and
// the extension ray leaving vertex k (path.cpp:189) share their origin and mint, so
//   trace2(o, mint, dE, maxtE, hasE, dS, maxtS, hasS) -> (F4 hit record of E, bool S occluded)
// resolves both in one pass over the geometry. The float additions into `result` keep
// the reference's order: emitter-sampling term of vertex k (path.cpp:171), then the
// emission term of vertex k+1 (:126-129) — exactly what the HBM-queue plan does.
// A pixel's only state between two camera samples is its PCG32 state and its sample
// counter (st: state lo, hi, flags, index), so a pixel can be advanced in passes:
// this call runs samples [st.w, sample_end) and returns the updated st word.
// `work` feeds the loop: fetch(pixel, st) hands out the next pixel of this lane (false: none left),
// store(st) publishes a pixel's state when its run is over, put(...) is block->put(). A lane that
// finishes a pixel fetches the next one INSIDE the iteration loop, so the other lanes of its wavefront
// never wait for it (the device feeds lanes from one shared queue; the CPU checker hands out one pixel).
template <int Mats = MATS_ALL, bool Analytic = true, typename Work, typename Trace2, typename Cnt>
MIW_HD void pixel_stream_render(const RenderParams &P, const SceneView &sc, uint32_t sample_end, Work &work,
                                Trace2 trace2, Cnt *cnt_local) {
    LaneRegs L;
    L.flags = LF_DONE; L.sample_idx = 0; L.rng.state = 0; L.rng.inc = MIW_PCG32_SCALAR_INC;
    uint32_t pixel = 0;
    bool have = false, dead_pending = false;
    ShadowOut sh; sh.has = false; sh.d = v3(0.f); sh.maxt = -1.f; sh.c = spec(0.f);
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
            if (!have) { if (work.exhausted()) break; continue; }    // (not exhausted: the job this lane drew waits for the chunk before it — ask again next trip)
            L.rng.state = (uint64_t) st.x | ((uint64_t) st.y << 32);
            L.sample_idx = st.w; L.flags = 0;
            lane_begin_sample(P, pixel, L, sample_end);
            MIW_SECTION(0);
            continue;
        }
        work.tick(L.sample_idx, true);                               // scheduling hint of the device's queue (no-op elsewhere)
        MIW_SECTION(0);
        const V3 o = L.ray.o;
        F4 h; bool occluded = false;
        trace2(o, L.ray.mint, L.ray.d, L.ray.maxt, !dead_pending, sh.d, sh.maxt, sh.has, h, occluded);
        if (sh.has && !occluded) L.res = L.res + sh.c;              // path.cpp:171 of the previous vertex
        sh.has = false;
        int r = STEP_FINISHED;
        if (!dead_pending) {
            r = path_step<Mats, Analytic>(P, sc, L, h, [o]() { return o; }, sh, cnt_local);
            MIW_SECTION(4);
            if (r == STEP_DEAD_PENDING) { dead_pending = true; continue; }   // one more pass for its shadow ray
            if (r == STEP_CONTINUE) continue;
        }
        dead_pending = false;
        lane_finish_sample(P, pixel, L, sink);
        if (cnt_local) cnt_local->samples++;
        L.flags = 0;
        lane_begin_sample(P, pixel, L, work.job_end(L.sample_idx, sample_end));   // (the end of this lane's JOB: the launch's last sample, or the chunk's)
        MIW_SECTION(5);
    }
}

template <int Mats, bool Analytic, typename Work, typename Trace2, typename Cnt>
MIW_HD void pixel_stream_render_direct(const RenderParams &P, const SceneView &sc, uint32_t sample_end, Work &work,
                                       Trace2 trace2, Cnt *cnt_local);     // direct.h

// One pixel, samples [st.w, sample_end): returns the updated st word.
template <uint32_t Integ = INTEG_PATH, typename Trace2, typename Sink, typename Cnt>
MIW_HD U4 pixel_render(const RenderParams &P, const SceneView &sc, uint32_t pixel, U4 st, uint32_t sample_end,
                       Trace2 trace2, Sink sink, Cnt *cnt_local) {
    struct OnePixel {
        uint32_t pixel; U4 st; bool taken; Sink sink;
        MIW_HD bool fetch(uint32_t &px, U4 &s) { if (taken) return false; taken = true; px = pixel; s = st; return true; }
        MIW_HD void store(U4 s) { st = s; }
        MIW_HD bool exhausted() const { return true; }
        MIW_HD uint32_t job_end(uint32_t, uint32_t sample_end) const { return sample_end; }
        MIW_HD void tick(uint32_t, bool) { }
        MIW_HD void put(uint32_t px, uint32_t sample_idx, V2 pos, const float *aovs) { sink(px, sample_idx, pos, aovs); }
    } work{ pixel, st, false, sink };
    if constexpr (Integ == INTEG_DIRECT) pixel_stream_render_direct<MATS_ALL, true>(P, sc, sample_end, work, trace2, cnt_local);
    else pixel_stream_render(P, sc, sample_end, work, trace2, cnt_local);
    return work.st;
}

// Seed word of a fresh lane (sampler.cpp:83-96 via integrator.cpp:198)
MIW_HD U4 lane_seed_state(uint64_t seed) {
    PCG32 r; pcg32_seed(r, seed, MIW_PCG32_DEFAULT_STREAM);
    U4 st; st.x = (uint32_t) r.state; st.y = (uint32_t) (r.state >> 32); st.z = 0; st.w = 0;
    return st;
}

} // namespace miw
