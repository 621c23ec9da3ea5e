// Plan 2 for scenes whose queries walk a tree: k_path_phased — the resident pixel loop of resident_kernel.h run as a
// WAVE-LEVEL PHASE MACHINE. Part of the single translation unit csrc/miwave.hip (not a stand-alone header).
//
// Why: in k_path_resident a wavefront executes  trace(E) ; trace(S) ; path_step  once per depth-loop iteration, and
// each of the three lasts as long as its slowest lane: max_lanes(E) + max_lanes(S) + shade, iteration after iteration
// (walk statistics of the material-ball scene: node loop 40 - 45 % SIMT efficiency). Nothing ties the lanes of a
// wavefront together — one lane = one pixel = one PCG32 stream — so here every lane carries its own little state
//   SHADE -> TRAV_E -> [TRAV_S] -> SHADE ...        (a traversal being a run of node steps and triangle tests)
// and the WAVE repeatedly votes (three ballots) which body to run next for the lanes that are ready for it:
//   node step      one BVH2 node: two slab tests, push / pop on the per-lane LDS stack
//   triangle test  one Moeller-Trumbore test (+ the accept rule) of the leaf range a lane holds
//   shade          everything between two scene queries: add the resolved emitter-sampling term, path_step,
//                  sample finish (log write), next camera ray, next pixel from the shared queue
// A lane that finishes its E walk starts its S walk at once, a lane that finishes both waits for the next shade
// phase while the others keep walking, and after a shade phase the lanes re-enter the walk together. The wave's time is
// then ~ the SUM of its lanes' work divided by the (much higher) lane count per body, instead of a sum of maxima.
// Per-lane arithmetic is untouched — the same path_step / prim_intersect calls in the same per-lane order — so the
// film is the same bit for bit (the sample log is indexed by lane and sample, not by time).
//
// MIW_PHASE_SPEC (template Spec): a lane holding an untested leaf range may keep descending (it is eligible for node
// steps and triangle tests); it stalls only when it reaches a second leaf.

enum : uint32_t { PH_SHADE = 0, PH_TRAV_E = 1, PH_TRAV_S = 2, PH_OUT = 3 };

#ifndef MIW_PHASE_SPEC
#define MIW_PHASE_SPEC 0
#endif
#ifndef MIW_PHASE_NODE_BURST
#define MIW_PHASE_NODE_BURST 4      /* node steps per vote while the node lanes stay a majority */
#endif

template <int Mats, bool Analytic, bool Spec>
__global__ __launch_bounds__(MIW_BLOCK, MIW_TREE_WAVES) void k_path_phased(RenderParams P, SceneView sc, LaneQueues Q, Counters *cnt,
                                                                             TraceLds cfg, uint32_t sample_end, uint32_t *next_pixel) {
    extern __shared__ uint4 smem[];
    stage_to_lds(sc, cfg, smem);
    int32_t *stack = reinterpret_cast<int32_t *>(smem + cfg.stack16) + threadIdx.x;
    const BvhNode *gnodes = sc.nodes;
    const Tri *gtris = sc.tris;
#if MIW_LDS_TOP
    const BvhNode *lnodes = reinterpret_cast<const BvhNode *>(smem);
    const uint32_t ns = cfg.nodes_staged;
#endif
    const PrimCtx ctx = prim_ctx(sc);
    Counters local; local.segments = local.samples = local.shadow_rays = local.active_lanes = 0;

    QueueWork work; work.Q = &Q; work.next_pixel = next_pixel; work.n_lanes = P.n_lanes; work.spp = P.spp; work.lane = 0;
    LaneRegs L;
    L.flags = LF_DONE; L.sample_idx = 0; L.rng.state = 0; L.rng.inc = MIW_PCG32_SCALAR_INC;
    uint32_t pixel = 0;
    bool have = false, dead_pending = false, occluded = false;
    ShadowOut sh; sh.has = false; sh.d = v3(0.f); sh.maxt = -1.f; sh.c = spec(0.f);
    F4 hitE; hitE.x = MIW_INFINITY; hitE.y = hitE.z = 0.f; hitE.w = u2f(MIW_MISS);
    uint32_t mode = PH_SHADE;

    // the walk a lane is in: current node (or pending leaf code, or DONE), stack depth, untested leaf range, best hit
    int32_t cur = MIW_WALK_DONE, sp = 0;
    uint32_t tri_i = 0, tri_end = 0;
    float tmax = 0.f, maxt_cur = 0.f;
    V3 d_cur = v3(0.f);
    FastRay r; r.inv_d = r.neg_o_inv_d = v3(0.f); r.mint = 0.f;
    Hit best; best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;

    auto begin_walk = [&](V3 d, float maxt) {
        r = fast_ray(L.ray.o, d, L.ray.mint);
        d_cur = d; maxt_cur = maxt; tmax = maxt;
        cur = 0; sp = 0; tri_i = tri_end = 0;
        best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    };
    auto pop = [&]() -> int32_t {
        if (sp == 0) return MIW_WALK_DONE;
        --sp; return stack[sp * MIW_BLOCK];
    };
    // a walk is over: E hands its hit record to the shade phase and starts S if a shadow ray is queued
    auto end_walk = [&](bool found_any) {
        if (mode == PH_TRAV_E) {
            hitE.x = best.t; hitE.y = best.u; hitE.z = best.v; hitE.w = u2f(best.tri);
            if (sh.has) { mode = PH_TRAV_S; begin_walk(sh.d, sh.maxt); }
            else mode = PH_SHADE;
        } else {
            occluded = found_any;
            mode = PH_SHADE;
        }
    };
    // after a step: an empty leaf range is refilled from a pending leaf code, or the walk ends
    auto settle = [&]() {
        if (tri_i < tri_end) return;
        if (cur >= 0) return;
        if (cur != MIW_WALK_DONE) {
            const uint32_t code = (uint32_t) ~cur;
            tri_i = code >> 4; tri_end = tri_i + (code & 15u) + 1u;
            cur = pop();
        } else end_walk(false);
    };

    for (;;) {
        const bool trav = (mode - 1u) < 2u;
        const bool e_leaf = trav && tri_i < tri_end;
        bool e_node = trav && cur >= 0 && (Spec || !e_leaf);
        const bool e_shade = mode == PH_SHADE;
        const int n_node = __popcll(__ballot(e_node)), n_leaf = __popcll(__ballot(e_leaf)), n_shade = __popcll(__ballot(e_shade));
        if ((n_node | n_leaf | n_shade) == 0) break;

        if (n_node >= n_leaf && n_node >= n_shade) {
            // ---------------- node steps ----------------
            const int floor_ = n_leaf > n_shade ? n_leaf : n_shade;
            for (int burst = 0; burst < MIW_PHASE_NODE_BURST; ++burst) {
                if (e_node) {
#if MIW_LDS_TOP
                    const BvhNode &n = (uint32_t) cur < ns ? lnodes[cur] : gnodes[cur];
#else
                    const BvhNode &n = gnodes[cur];
#endif
                    float tn0, tn1;
                    const float wide = widen(tmax);
                    const bool h0 = box_test_fast(n.lo0, n.hi0, r, wide, tn0), h1 = box_test_fast(n.lo1, n.hi1, r, wide, tn1);
                    const int32_t c0 = n.child0, c1 = n.child1;
                    const bool second_first = tn1 < tn0;
                    int32_t next = h0 ? c0 : c1;
                    if (h0 && h1) {
                        stack[sp * MIW_BLOCK] = second_first ? c0 : c1; ++sp;
                        next = second_first ? c1 : c0;
                    } else if (!(h0 || h1)) next = pop();
                    cur = next;
                    settle();
                    e_node = ((mode - 1u) < 2u) && cur >= 0 && (Spec || tri_i >= tri_end);
                }
                if (burst + 1 < MIW_PHASE_NODE_BURST && __popcll(__ballot(e_node)) * 2 < n_node + floor_) break;
            }
        } else if (n_leaf >= n_shade) {
            // ---------------- one triangle test per lane ----------------
            if (e_leaf) {
                const Tri &tr = gtris[tri_i];
                float t, u, v;
                bool found = false;
                if (prim_intersect<Analytic>(tr, ctx, L.ray.o, d_cur, L.ray.mint, maxt_cur, t, u, v)) {
                    if (mode == PH_TRAV_S) found = true;
                    else if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                        best.t = t; best.u = u; best.v = v; best.tri = tri_i; best.prim = tr.prim;
                        tmax = t;
                    }
                }
                ++tri_i;
                if (found) { tri_i = tri_end = 0; cur = MIW_WALK_DONE; end_walk(true); }
                else settle();
            }
        } else {
            // ---------------- shade: everything between two scene queries (pixel_stream_render's loop body) ----------------
            if (e_shade) {
                if (!(L.flags & LF_DONE)) {
                    const V3 o = L.ray.o;
                    if (sh.has && !occluded) L.res = L.res + sh.c;          // path.cpp:171 of the previous vertex
                    sh.has = false; occluded = false;
                    int rstep = STEP_FINISHED;
                    if (!dead_pending) rstep = path_step<Mats, Analytic>(P, sc, L, hitE, [o]() { return o; }, sh, &local);
                    if (!dead_pending && rstep == STEP_DEAD_PENDING) dead_pending = true;   // one more pass for its shadow ray
                    else if (dead_pending || rstep == STEP_FINISHED) {
                        dead_pending = false;
                        auto sink = [&work](uint32_t px, uint32_t sample_idx, V2 pos, const float *aovs) { work.put(px, sample_idx, pos, aovs); };
                        lane_finish_sample(P, pixel, L, sink);
                        local.samples++;
                        L.flags = 0;
                        lane_begin_sample(P, pixel, L, sample_end);
                    }
                }
                while (L.flags & LF_DONE) {                                 // pixel finished (or no pixel yet): take the next one
                    if (have) {
                        U4 st; st.x = (uint32_t) L.rng.state; st.y = (uint32_t) (L.rng.state >> 32);
                        st.z = L.sample_idx >= P.spp ? (uint32_t) LF_DONE : 0u; st.w = L.sample_idx;
                        work.store(st);
                    }
                    U4 st;
                    have = work.fetch(pixel, st);
                    if (!have) break;
                    L.rng.state = (uint64_t) st.x | ((uint64_t) st.y << 32);
                    L.sample_idx = st.w; L.flags = 0;
                    lane_begin_sample(P, pixel, L, sample_end);
                }
                if (L.flags & LF_DONE) mode = PH_OUT;
                else if (!dead_pending) { mode = PH_TRAV_E; begin_walk(L.ray.d, L.ray.maxt); }
                else { mode = PH_TRAV_S; begin_walk(sh.d, sh.maxt); }       // dead_pending implies a queued shadow ray
            }
        }
    }

    unsigned long long a = wave_sum(local.segments), b = wave_sum(local.samples), c = wave_sum(local.shadow_rays);
    if ((threadIdx.x & 63) == 0) {
        Counters *shard = cnt + ((blockIdx.x * (MIW_BLOCK / 64) + (threadIdx.x >> 6)) & (MIW_CNT_SHARDS - 1));
        if (a) atomicAdd(&shard->segments, a);
        if (b) atomicAdd(&shard->samples, b);
        if (c) atomicAdd(&shard->shadow_rays, c);
    }
}
