// Plan 2 for scenes whose queries walk a tree: k_path_phased — the resident pixel loop of resident_kernel.h run as a
// WAVE-LEVEL PHASE MACHINE. Part of the single translation unit csrc/miwave.hip (not a stand-alone header).
//
// Why: in k_path_resident a wavefront executes  trace(E) ; trace(S) ; path_step  once per depth-loop iteration, and
// each of the three lasts as long as its slowest lane: max_lanes(E) + max_lanes(S) + shade, iteration after iteration.
// Walk lengths are heavy-tailed (a ray that meets a tessellated ball takes ten times the node steps of one that meets
// a wall), so the lanes of a wave mostly wait: measured SIMT efficiency of the node loop 12 % (material balls) and
// 16 % (0.9 M-triangle interior), of the triangle loop 19 % / 23 % (-DMIW_WALK_STATS=1 builds, profiles/).
// Nothing ties the lanes of a wavefront together — one lane = one pixel = one PCG32 stream — so here every lane
// carries its own little state
//   SHADE -> TRAV_E -> [TRAV_S] -> SHADE ...        (a traversal being a run of node steps and triangle tests)
// and the WAVE repeatedly votes (ballots + s_bcnt1) which body to run next for the lanes that are ready for it:
//   node step      one 64-byte BVH4 node: four quantised child boxes (miw/bvh4.h), slab tests, sort, push / pop on the per-lane
//                  LDS stack (template Wide = false: one BVH2 node, two boxes)
//   triangle test  one Moeller-Trumbore test (+ the accept rule) of the leaf range a lane holds
//   walk end       hand the hit record over / start the shadow walk (a handful of moves)
//   shade          everything between two scene queries: add the resolved emitter-sampling term, path_step,
//                  sample finish (log write), next camera ray, next pixel from the shared queue
// A lane that finishes its E walk starts its S walk at once, a lane that finishes both waits for the next shade
// phase while the others keep walking, and after a shade phase the lanes re-enter the walk together. The wave's time is
// then ~ the SUM of its lanes' work divided by the (much higher) lane count per body, instead of a sum of maxima.
// Per-lane arithmetic is untouched — the same path_step / prim_intersect calls in the same per-lane order — so the
// film is the same bit for bit (the sample log is indexed by lane and sample, not by time).
//
// Code shape matters as much as the schedule: every body is one exec-masked region that writes the few state
// registers it owns (node step: cur, sp, leaf range; triangle test: best hit, tmax, leaf range, cur, sp), and the
// rare transitions (walk end, shade) live in their own bodies, so the hot bodies carry no phi copies of the ray or
// path state. MIW_PHASE_SPEC (template Spec): a lane holding an untested leaf range may keep descending; it stalls
// only when it reaches a second leaf.

// PH_PARK (chunk jobs, resident_kernel.h): the lane drew a chunk whose predecessor is still running; it asks again in the next shade run but does NOT vote for one —
// the lane it waits for may be walking in this very wavefront, and parked lanes that outvote the walkers would starve it. A wavefront with nothing but parked lanes runs shade.
enum : uint32_t { PH_SHADE = 0, PH_TRAV_E = 1, PH_TRAV_S = 2, PH_OUT = 3, PH_PARK = 4 };

// a lane's stack: its column of the workgroup's entry-major LDS array (slot i of lane l at i * MIW_BLOCK + l: conflict-free)
// The triangle records in global memory as the walk bodies read them: 48 bytes = three 16-byte loads through an explicitly global
// pointer. tri_fetch2 (miw/bvh4.h: the two records of a triangle step) issues all six loads before it waits — inline assembly, as
// the compiler orders two plain reads "first record, wait, second record" to save registers (one more round trip per triangle
// step). vmcnt counts in issue order, so loads the compiler has in flight around these only make the wait longer, never shorter.
typedef uint32_t miw_u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) miw_u4 *GlobalU4;
struct GlobalTris {
    GlobalU4 p;
    __device__ __forceinline__ Tri operator()(uint32_t i) const {
        GlobalU4 a = p + 3 * (size_t) i;
        miw_u4 q[3] = { a[0], a[1], a[2] };
        Tri t; __builtin_memcpy(&t, q, sizeof t); return t;
    }
};
#ifndef MIW_TRI_FETCH2_ASM
#define MIW_TRI_FETCH2_ASM 1
#endif
__device__ __forceinline__ void tri_fetch2(const GlobalTris &g, uint32_t a, uint32_t b, Tri &ta, Tri &tb) {
#if MIW_TRI_FETCH2_ASM && defined(__HIP_DEVICE_COMPILE__)
    GlobalU4 pa = g.p + 3 * (size_t) a, pb = g.p + 3 * (size_t) b;
    miw_u4 q[6];
    // ONE asm statement, early-clobber outputs: the six destinations are not defined (for the compiler: may not be copied, spilled
    // or shared with the address registers) before the wait at its end has passed — ADVICE r05
    asm volatile("global_load_dwordx4 %0, %6, off\n\t"
                 "global_load_dwordx4 %1, %6, off offset:16\n\t"
                 "global_load_dwordx4 %2, %6, off offset:32\n\t"
                 "global_load_dwordx4 %3, %7, off\n\t"
                 "global_load_dwordx4 %4, %7, off offset:16\n\t"
                 "global_load_dwordx4 %5, %7, off offset:32\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]) : "v"(pa), "v"(pb) : "memory");
    __builtin_memcpy(&ta, &q[0], sizeof ta); __builtin_memcpy(&tb, &q[3], sizeof tb);
#else
    ta = g(a); tb = g(b);
#endif
}

struct LdsColumn { int32_t *p; __device__ __forceinline__ int32_t &operator[](int32_t i) const { return p[i * MIW_BLOCK]; } };
// the 8-wide walk's column: MIW_BVH8_STACK node groups of 8 bytes (ds_write_b64 / ds_read_b64; the same 128 bytes per lane)
struct LdsColumn8 { U2 *p; __device__ __forceinline__ U2 &operator[](int32_t i) const { return p[i * MIW_BLOCK]; } };
// (a launch that does not size the column by the 8-wide tree's depth reuses the 4-wide walk's: it must hold MIW_BVH8_STACK groups)
static_assert(MIW_STACK_ENTRIES * sizeof(int32_t) >= MIW_BVH8_STACK * sizeof(U2), "the 4-wide walk's LDS column is too small for the 8-wide walk's groups");

#ifndef MIW_PIN_TREE_PTRS
#define MIW_PIN_TREE_PTRS 1           /* 1: node / triangle table pointers of the walk bodies kept in registers (below); C3 +1.5 %, gpurun r4e */
#endif
#ifndef MIW_PHASE_SPEC
#define MIW_PHASE_SPEC 1
#endif
#ifndef MIW_W8_SPEC
#define MIW_W8_SPEC 1               /* 1 (default): the 8-wide walk speculates as well — a second pending triangle group, miw/bvh8.h; C4 +1.7 %, C3 +0.4 % over the 4-wide walk, gpurun r5d; 0: triangles before nodes (-2.7 % / -1.6 % behind it) */
#endif
#ifndef MIW_TRI_PAIR
#define MIW_TRI_PAIR 1              /* 1: the triangle body tests two triangles of a leaf range per iteration (both fetched up front: +2 - 4 %); 0: one */
#endif
#ifndef MIW_PHASE_END_WEIGHT
#define MIW_PHASE_END_WEIGHT 4      /* the walk-end body is cheap: it runs once a quarter as many lanes wait for it as for the leading body */
#endif
// Waves = waves per SIMD the kernel is compiled for: 3 (168 VGPRs) or 4 (128 VGPRs; the default for every tree since the kernel
// spills 20 registers at 128 instead of 121: DESIGN.md section 4, "the register diet").
// Wide = which tree the node body steps through: 2 = the 8-wide quantised tree of miw/bvh8.h (80-byte nodes, node / triangle GROUPS
// as walk state, triangles in the tree's own order: the default since round 5), 1 = the 4-wide quantised tree of miw/bvh4.h (trees the
// 8-wide collapse refuses, the radix tree, MIW_BVH8=0: A/B runs), 0 = the BVH2 (MIW_BVH4=0: instantiated for the MATS_TRIO kernels only).
// Placed: the pixel queue is QueueWork<true> (resident_kernel.h) — shards of about one pixel per resident lane: a measuring launch,
// then every wavefront takes pixels of about equal cost from the queue of its SIMD. The full-frame kernel keeps Placed = false
// and the four registers the queue choice costs.
#ifndef MIW_PHASED_JOBS
#define MIW_PHASED_JOBS 1      /* the full-frame instantiations draw chunk jobs (resident_kernel.h: QueueWork::fetch_job); 0 compiles them out (A/B builds) */
#endif
template <int Mats, bool Analytic, bool Spec, int Waves = MIW_TREE_WAVES, int Wide = 1, bool Placed = false>
__global__ __launch_bounds__(MIW_BLOCK, Waves) void k_path_phased(RenderParams P, SceneView sc, LaneQueues Q, Counters *cnt,
                                                                             TraceLds cfg, uint32_t sample_end, uint32_t *next_pixel) {
    extern __shared__ uint4 smem[];
    stage_to_lds(sc, cfg, smem);
    const float *thr = stage_thresholds(smem, cfg, Q.log_rec ? Q.log_thr : nullptr);
#if MIW_LDS_TABLES
    stage_tables<false>(sc, cfg, smem);
#endif
    int32_t *stack = reinterpret_cast<int32_t *>(smem + cfg.stack16) + threadIdx.x;
    U2 *stack8 = reinterpret_cast<U2 *>(smem + cfg.stack16) + threadIdx.x;
    (void) stack8;
    const BvhNode *gnodes = sc.nodes;
    const Bvh4Node *nodes4 = sc.nodes4;
    const Tri *gtris = sc.tris;
    (void) gnodes; (void) nodes4;
#if MIW_PIN_TREE_PTRS && defined(__HIP_DEVICE_COMPILE__)
    // The walk bodies read these two pointers on every trip, in front of the load every trip waits for. Left to itself the
    // compiler (short of SGPRs) re-reads them from the kernel arguments inside the loops — an s_load + s_waitcnt on the critical
    // path of every node step. Made opaque here they are values it has to KEEP: in SGPRs or, spilled, in a VGPR lane (v_readlane:
    // a couple of cycles instead of a scalar-cache round trip). An opaque pointer is a generic one, so the records are read
    // through explicitly global pointers to 16-byte vectors (global_load_dwordx4, as before) and handed on by value.
    GlobalU4 nodes4_g = (GlobalU4) reinterpret_cast<uintptr_t>(sc.nodes4), tris_g = (GlobalU4) reinterpret_cast<uintptr_t>(sc.tris);
    GlobalU4 nodes8_g = (GlobalU4) reinterpret_cast<uintptr_t>(sc.nodes8);
    if (Wide == 2) asm volatile("" : "+s"(nodes8_g)); else asm volatile("" : "+s"(nodes4_g));
    asm volatile("" : "+s"(tris_g));
    auto node4_at = [nodes4_g](int32_t i) -> Bvh4Node {
        GlobalU4 p = nodes4_g + 4 * (size_t) (uint32_t) i;
        miw_u4 q[4] = { p[0], p[1], p[2], p[3] };
        Bvh4Node n; __builtin_memcpy(&n, q, sizeof n); return n;
    };
    auto node8_at = [nodes8_g](uint32_t i) -> Bvh8Node {           // five 16-byte requests
        GlobalU4 p = nodes8_g + 5 * (size_t) i;
        miw_u4 q[5] = { p[0], p[1], p[2], p[3], p[4] };
        Bvh8Node n; __builtin_memcpy(&n, q, sizeof n); return n;
    };
    const GlobalTris tri_at_g{ tris_g };
#else
    auto node4_at = [nodes4](int32_t i) -> const Bvh4Node & { return nodes4[i]; };
    const Bvh8Node *nodes8 = sc.nodes8;
    auto node8_at = [nodes8](uint32_t i) -> const Bvh8Node & { return nodes8[i]; };
    auto tri_at_g = [gtris](uint32_t i) -> const Tri & { return gtris[i]; };
#endif
#if MIW_LDS_TOP
    const BvhNode *lnodes = reinterpret_cast<const BvhNode *>(smem);
    const uint32_t ns = cfg.nodes_staged;
#endif
    const PrimCtx ctx = prim_ctx(sc);
    LaneCounters local; local.segments = local.samples = local.shadow_rays = 0;
#if defined(MIW_SECTION_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if ((threadIdx.x & 63u) == 0) { unsigned long long *b_ = miw_sec_buf(); for (int i = 0; i < 15; ++i) b_[i] = 0; b_[15] = __builtin_amdgcn_s_memtime(); }
#endif

    QueueWork<Placed, true, false, MIW_PHASED_JOBS != 0 && !Placed> work; work.Q = &Q; work.next_pixel = next_pixel; work.n_lanes = P.n_lanes; work.spp = P.spp; work.lane = 0; work.warn_negative = P.film.warn_negative;
    work.film = &P.film; work.thr = thr; work.init_queues(cfg.queues ? cfg.queues : 1u);
    __shared__ uint32_t s_prog[MIW_BLOCK / 64];
    if (cfg.tail_prio) work.enable_tail_prio(sample_end, &s_prog[threadIdx.x >> 6]);
    LaneRegs L;
    L.flags = LF_DONE; L.sample_idx = 0; L.rng.state = 0; L.rng.inc = MIW_PCG32_SCALAR_INC;
    uint32_t pixel = 0;
    bool have = false, dead_pending = false, occluded = false;
    ShadowOut sh; sh.has = false; sh.d = v3(0.f); sh.maxt = -1.f; sh.c = spec(0.f);
    uint32_t mode = PH_SHADE;

    // the walk a lane is in: current node (>= 0), pending leaf code (< 0) or DONE; stack depth; untested leaf range; best hit
    int32_t cur = MIW_WALK_DONE, sp = 0;
    uint32_t tri_i = 0, tri_end = 0;
    // Wide == 2: the walk is two groups (miw/bvh8.h): node group (gb, gm: pending slots | imask | octant | stack depth), triangle group (tb, tm)
    Walk8 w8; w8.gb = 0u; w8.gm = 0u; w8.tb = 0u; w8.tm = 0u; w8.tb2 = 0u; w8.tm2 = 0u;
    constexpr bool Spec8 = Spec && (MIW_W8_SPEC != 0);
    // (direction, maxt and mint of the walk in progress are the path state's own — L.ray for an E walk, sh.d / sh.maxt for an S
    // walk, selected by `mode` where the triangle body needs them — not copies that would be live through the shade body)
    float tmax = 0.f;
    FastRay r; r.inv_d = r.neg_o_inv_d = v3(0.f); r.mint = 0.f;
    Hit best; best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;

    auto begin_walk = [&](V3 d, float maxt) {
        r = fast_ray(L.ray.o, d, L.ray.mint);
        tmax = maxt;
        if (Wide == 2) walk8_begin(w8, r); else { cur = 0; sp = 0; tri_i = tri_end = 0; }
        best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    };

    // -DMIW_PHASE_STATS=1 (debug builds): per body, how often a wave ran it, with how many lanes, and the wall cycles it took
#if defined(MIW_PHASE_STATS)
    // (round 6: every stamp stands at the END of what it charges — a node / triangle trip, the E -> S turn, a shade run, the vote — so that no
    // body is charged the last trip of the loop before it or the vote; buckets: 0 node trip, 1 triangle trip, 2 walk end (turn), 3 shade, 4 vote)
    unsigned long long ps_runs[5] = { 0, 0, 0, 0, 0 }, ps_lanes[5] = { 0, 0, 0, 0, 0 }, ps_cycles[5] = { 0, 0, 0, 0, 0 }, ps_t0 = __builtin_amdgcn_s_memtime();
#define MIW_PS(k, lanes_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ps_runs[k]++; ps_lanes[k] += (unsigned) (lanes_); ps_cycles[k] += now_ - ps_t0; ps_t0 = now_; } while (0)
#else
#define MIW_PS(k, lanes_) do { } while (0)
#endif
    // lane predicates -> lane counts: ballot + s_bcnt1 (the builtin keeps the predicate in an SGPR pair)
    auto count = [](bool p) -> int { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); };
    // shade once the shade-ready lanes outnumber the busier walk body num : den (host: 3 : 2, 2 : 1 with an environment map —
    // a shade run costs ~2 200 instruction slots whatever its lane count, a walk step ~100 - 200, so shade runs are worth filling)
    const int shade_num = (int) cfg.shade_num, shade_den = (int) cfg.shade_den;
    const int nx_c = (int) cfg.node_exit, tx_c = (int) cfg.tri_exit;
    for (;;) {
        // ---- the vote: which lanes are ready for which body ----
        const bool trav = (mode - 1u) < 2u;
        bool has_range = Wide == 2 ? walk8_tri_ready(w8) : tri_i < tri_end;
        bool e_leaf = trav && has_range;
        bool e_node = trav && (Wide == 2 ? walk8_node_ready<Spec8>(w8) : cur >= 0 && (Spec || !has_range));
        // a walk that is over: an E walk with a shadow ray queued turns into the S walk at the next entry to the node body
        // (`e_turn`; the hit record stays in `best`, which an S walk never writes), every other one is ready to shade
        const bool walk_over = trav && (Wide == 2 ? walk8_over(w8) : !has_range && cur == MIW_WALK_DONE);
        const bool e_turn = walk_over && mode == PH_TRAV_E && sh.has;
        const bool e_shade = mode == PH_SHADE || (walk_over && !e_turn);
        const int n_turn = count(e_turn);
        int n_node = count(e_node) + n_turn, n_leaf = count(e_leaf);
        const int n_shade = count(e_shade);
        const int n_end = 0;
        bool parked_only = false;
        if ((n_node | n_leaf | n_shade) == 0) {
            if (!(MIW_PHASED_JOBS != 0 && !Placed) || __ballot(mode == PH_PARK) == 0ull) break;
            parked_only = true;
        }
        const int lead = n_node > n_leaf ? n_node : n_leaf;              // the busier walk body
        MIW_PS(4, 0);                                                    // the vote (+ the loop exit tests of the body before it)

        if ((n_shade * shade_num >= lead * shade_den && n_shade > 0) || parked_only) {
            // ---------------- shade: everything between two scene queries (pixel_stream_render's loop body) ----------------
            MIW_SECTION(6);                                              // everything since the last shade body: walks + votes
            work.tick(L.sample_idx, mode != PH_OUT && !(L.flags & LF_DONE));
            if (e_shade || (MIW_PHASED_JOBS != 0 && !Placed && mode == PH_PARK)) {
                if (!(L.flags & LF_DONE)) {
                    const V3 o = L.ray.o;
                    if (sh.has && !occluded) L.res = L.res + sh.c;          // path.cpp:171 of the previous vertex
                    sh.has = false; occluded = false;
                    int rstep = STEP_FINISHED;
                    F4 hitE; hitE.x = best.t; hitE.y = best.u; hitE.z = best.v; hitE.w = u2f(best.tri);   // of the E walk (unused when dead_pending)
                    if (!dead_pending) rstep = path_step<Mats, Analytic>(P, sc, L, hitE, [o]() { return o; }, sh, &local);
                    if (!dead_pending && rstep == STEP_DEAD_PENDING) dead_pending = true;   // one more pass for its shadow ray
                    else if (dead_pending || rstep == STEP_FINISHED) {
                        dead_pending = false;
                        auto sink = [&work](uint32_t px, uint32_t sample_idx, V2 pos, const float *aovs) { work.put(px, sample_idx, pos, aovs); };
                        lane_finish_sample(P, pixel, L, sink);
                        local.samples++;
                        L.flags = 0;
                        lane_begin_sample(P, pixel, L, work.job_end(L.sample_idx, sample_end));   // (the end of the lane's JOB: chunk jobs, resident_kernel.h)
                    }
                }
                MIW_SECTION(11);
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
                if (L.flags & LF_DONE) mode = work.exhausted() ? PH_OUT : PH_PARK;    // (not exhausted: a chunk job waiting for the chunk before it — it asks again in the next shade run)
                else if (!dead_pending) { mode = PH_TRAV_E; begin_walk(L.ray.d, L.ray.maxt); }
                else { mode = PH_TRAV_S; begin_walk(sh.d, sh.maxt); }       // dead_pending implies a queued shadow ray
            }
            MIW_SECTION(12);
            MIW_PS(3, n_shade);
        } else if (n_node >= n_leaf) {                           // (a bias either way — a x node lanes >= b x leaf lanes, a : b from 1 : 2 to 3 : 1 — measured flat or worse: gpurun r5n / r5o)
            // ---------------- node steps: an inner loop that owns cur, sp and the leaf range only; it runs while the node
            // lanes remain the largest group (lanes that reach a leaf or the end of their walk drop out of it) ----------------
            const int others = n_shade > n_end * MIW_PHASE_END_WEIGHT ? n_shade : n_end * MIW_PHASE_END_WEIGHT;
            if (n_turn > 0) {                                            // E walks that ended with a shadow ray queued: start the S walk
                if (e_turn) {
                    mode = PH_TRAV_S;
                    r = fast_ray(L.ray.o, sh.d, L.ray.mint);
                    tmax = sh.maxt;
                    if (Wide == 2) walk8_begin(w8, r); else { cur = 0; sp = 0; tri_i = tri_end = 0; }
                    e_node = true;
                }
                MIW_PS(2, n_turn);
            }
            do {
                const int ps_now_ = count(e_node); (void) ps_now_;
                if (e_node) {
                    if (Wide == 2) {
                        // one 80-byte node = eight quantised child boxes: the node step of miw/bvh8.h (the CPU checker runs the same statements)
                        FastRay rn; rn.inv_d = r.inv_d; rn.neg_o_inv_d = r.neg_o_inv_d; rn.mint = L.ray.mint;
                        const auto &nd = node8_at(walk8_next_node(w8));
                        walk8_node_step<Spec8>(nd, rn, widen(tmax), w8, LdsColumn8{ stack8 });
                    } else if (Wide) {
                        // one 64-byte node = four quantised child boxes: the node step of miw/bvh4.h (the CPU checker runs the same statements)
                        const LdsColumn column{ stack };
                        FastRay rn; rn.inv_d = r.inv_d; rn.neg_o_inv_d = r.neg_o_inv_d; rn.mint = L.ray.mint;   // (r.mint would be one more register carried through the shade body)
                        const auto &nd = node4_at(cur);
                        MIW_WALK4_NODE_STEP(Spec, nd, rn, widen(tmax), cur, sp, tri_i, tri_end, column);
                    } else {
                        int32_t next = MIW_WALK_DONE;
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
                        next = h0 ? c0 : c1;
                        if (h0 && h1) {
                            stack[sp * MIW_BLOCK] = second_first ? c0 : c1; ++sp;
                            next = second_first ? c1 : c0;
                        } else if (!(h0 || h1)) {
                            next = MIW_WALK_DONE;
                            if (sp != 0) { --sp; next = stack[sp * MIW_BLOCK]; }
                        }
                        // a leaf: it becomes the lane's triangle range (if it holds none), and the next stack entry its current node
                        if (next < 0 && next != MIW_WALK_DONE && (!Spec || tri_i >= tri_end)) {
                            const uint32_t code = (uint32_t) ~next;
                            tri_i = code >> 4; tri_end = tri_i + (code & 15u) + 1u;
                            next = MIW_WALK_DONE;
                            if (sp != 0) { --sp; next = stack[sp * MIW_BLOCK]; }
                        }
                        cur = next;
                    }
                }
                MIW_PS(0, ps_now_);
                has_range = Wide == 2 ? walk8_tri_ready(w8) : tri_i < tri_end;
                e_node = trav && (Wide == 2 ? walk8_node_ready<Spec8>(w8) : cur >= 0 && (Spec || !has_range));
                const int now = count(e_node);
                n_leaf = count(trav && has_range);
                // hand over once the node lanes are clearly outnumbered: by the leaf lanes 2 : 1, or by the waiting lanes nx_c : 1 (host: 1, with an
                // environment map 2). (The loops of rounds 3 - 4 left at now < leaf lanes and bounced between the bodies: C3 +4 %, C4 +3 %, gpurun r5f - r5h.)
                if (2 * now < n_leaf || nx_c * now < others || now == 0) break;
            } while (true);
        } else {
            // ---------------- triangle tests: same shape; owns the best hit, tmax, the leaf range, cur and sp ----------------
            const int others = n_shade > n_end * MIW_PHASE_END_WEIGHT ? n_shade : n_end * MIW_PHASE_END_WEIGHT;
            do {
                const int ps_now_ = count(e_leaf); (void) ps_now_;
                if (e_leaf) {
                    const bool s_walk = mode == PH_TRAV_S;
                    const V3 d_cur = s_walk ? sh.d : L.ray.d;
                    const float maxt_cur = s_walk ? sh.maxt : L.ray.maxt;
#if MIW_TRI_PAIR
                    if (Wide == 2) {
                        // the two lowest pending triangles of the lane's triangle group (miw/bvh8.h); a drained group hands over to the next node group
                        // (the node step leaves a non-empty node group behind whenever the stack holds one: nothing to pop here)
                        walk8_tri_step<Analytic, Spec8>(tri_at_g, ctx, L.ray.o, d_cur, L.ray.mint, maxt_cur, mode == PH_TRAV_S, best, tmax, occluded, w8);
                    } else
                    // two triangles of the lane's range per trip: walk4_tri_step (miw/bvh4.h — shared with the CPU checker)
                    walk4_tri_step<Analytic>(tri_at_g, ctx, L.ray.o, d_cur, L.ray.mint, maxt_cur,
                                             mode == PH_TRAV_S, best, tmax, occluded, cur, sp, tri_i, tri_end, LdsColumn{ stack });
#else
                    static_assert(Wide != 2, "the 8-wide walk has the pair step only");
                    const Tri &tr = gtris[tri_i];
                    float t, u, v;
                    if (prim_intersect<Analytic>(tr, ctx, L.ray.o, d_cur, L.ray.mint, maxt_cur, t, u, v)) {
                        if (mode == PH_TRAV_S) {                         // any hit ends the shadow walk
                            occluded = true; tri_end = 0; cur = MIW_WALK_DONE; sp = 0;
                        } else if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                            best.t = t; best.u = u; best.v = v; best.tri = tri_i; best.prim = tr.prim;
                            tmax = t;
                        }
                    }
                    ++tri_i;
                    if (tri_i >= tri_end && cur < 0 && cur != MIW_WALK_DONE) {   // range drained and the stack handed over another leaf
                        const uint32_t code = (uint32_t) ~cur;
                        tri_i = code >> 4; tri_end = tri_i + (code & 15u) + 1u;
                        cur = MIW_WALK_DONE;
                        if (sp != 0) { --sp; cur = stack[sp * MIW_BLOCK]; }
                    }
#endif
                }
                MIW_PS(1, ps_now_);
                has_range = Wide == 2 ? walk8_tri_ready(w8) : tri_i < tri_end;
                e_leaf = trav && has_range;
                const int now = count(e_leaf);
                n_node = count(trav && (Wide == 2 ? walk8_node_ready<Spec8>(w8) : cur >= 0 && (Spec || !has_range)));
                if (2 * now <= n_node || tx_c * now < others || now == 0) break;
            } while (true);
        }
    }

#if defined(MIW_PHASE_STATS)
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 5; ++k) { atomicAdd(&g_phase_stats[k], ps_runs[k]); atomicAdd(&g_phase_stats[5 + k], ps_lanes[k]); atomicAdd(&g_phase_stats[10 + k], ps_cycles[k]); }
#endif
    unsigned long long a = wave_sum(local.segments), b = wave_sum(local.samples), c = wave_sum(local.shadow_rays);
#if defined(MIW_SECTION_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if ((threadIdx.x & 63) == 0) { unsigned long long *b_ = miw_sec_buf(); for (int i = 0; i < 15; ++i) if (b_[i]) atomicAdd(&g_sections[i], b_[i]); }
#endif
    if ((threadIdx.x & 63) == 0) {
        Counters *shard = cnt + ((blockIdx.x * (MIW_BLOCK / 64) + (threadIdx.x >> 6)) & (MIW_CNT_SHARDS - 1));
        if (a) atomicAdd(&shard->segments, a);
        if (b) atomicAdd(&shard->samples, b);
        if (c) atomicAdd(&shard->shadow_rays, c);
    }
}
