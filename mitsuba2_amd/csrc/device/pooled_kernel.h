// Plan 2 for tree scenes, round 6: k_path_pooled — the wave-level phase machine of phased_kernel.h with the WALKS POOLED ACROSS THE
// WORKGROUP. Part of the single translation unit csrc/miwave.hip (not a stand-alone header).
//
// Why. In k_path_phased a lane owns one pixel AND that pixel's walk, so the 64 lanes of a wavefront are always split three ways
// (node step / triangle test / shade) and every body runs at ~33 of 64 lanes (profiles/r05_experiments.txt r5b); raising the shade
// vote only makes the finished lanes idle longer. Here the two are separated:
//   * the PATH state (rng, throughput, radiance, pixel, sample counter — what shade needs) stays in the registers of its home lane;
//   * the WALK of that path — ray, 8-wide walk groups, best hit: 80 bytes — is a JOB RECORD in LDS, its stack a column of the
//     workgroup's LDS stack array, and ANY lane of the same lane index in ANY wavefront of the workgroup may advance it.
// Lane i of every wavefront shares COLUMN i: the jobs of the NW lanes i (one per wavefront). A wavefront that votes "node steps"
// lets each of its lanes claim a node-ready job of its column (its own first), advance it while it stays node-ready, hand it back
// and take the next one; likewise for triangle tests. So the walk bodies run on (nearly) full wavefronts whatever the mix of
// states inside one wavefront's own pixels is, and — because a lane whose own walk is over now works on somebody else's instead
// of waiting — a wavefront can afford to shade only when most of its own walks are over: the shade body fills up too.
// The column rule makes the exchange cheap: job j = w * 64 + i lives at LDS word offset ... + j, so lane i touches bank group i
// whatever w it picks (conflict-free like the per-lane stack columns), no cross-lane compaction, no prefix sums; the status of a
// column's jobs is NW bytes = NW / 4 words, read with one or two LDS loads, claimed with ONE ds_or_rtn (the CLAIMED bit).
//
// Per-job arithmetic is untouched: the same walk8_node_step / walk8_tri_step calls on the same record in the same per-job order
// (miw/bvh8.h; the order BETWEEN node steps and triangle tests of a speculating walk is free), the same path_step
// on the home lane. The film is the same bit for bit; the sample log is indexed by lane and sample, not by time.
//
// Ownership protocol (one status byte per job; bit 7 = CLAIMED):
//   a byte WITHOUT the claimed bit is N (node-ready) / T (triangle-ready) / N|T / D (done: waits for its home lane's shade), plus
//   the flags S (the walk in progress is the shadow walk) and O (the shadow ray was occluded);
//   whoever wants the job — another lane for a walk body, the home lane for shade — sets the claimed bit with an atomic OR that
//   returns the old byte: old without the bit = the job is mine (if it is not in the state I wanted, I clear the bit again);
//   the owner hands the job back with a plain byte store of the new state (state words first, s_waitcnt, then the byte);
//   a job that is with its home lane (shading, no pixel) keeps the claimed bit, so nobody else can ever own it.
// Every job is always (a) with its home lane, which shades it as soon as it votes shade, (b) ready and unclaimed — the home
// wavefront itself can take it —, or (c) claimed by a lane inside a walk loop, which hands it back within that loop: no deadlock.
// A shadow walk runs BEFORE its extension walk (both are known when the job is posted; the lane that finishes the first turns the
// record into the second), so the E -> S hand-over body of the old kernel does not exist.

enum : uint32_t { PJ_N = 1u, PJ_T = 2u, PJ_D = 4u, PJ_S = 0x10u, PJ_O = 0x20u, PJ_C = 0x80u, PJ_STATE = PJ_C | PJ_N | PJ_T | PJ_D };
enum : uint32_t { PM_SHADE = 0u, PM_WALK = 1u, PM_OUT = 2u };
#define MIW_POOL_NOJOB 0xffffffffu

template <int NJ_> struct PoolColumn8 { U2 *p; __device__ __forceinline__ U2 &operator[](int32_t i) const { return p[i * NJ_]; } };

#if defined(MIW_PHASE_STATS)
__device__ unsigned long long g_pool_stats[32];    // per bucket (vote, node, triangle, shade, idle): runs, lanes, wall cycles; + claims tried / won
#endif

// NW = wavefronts per workgroup (one workgroup per CU); PP = pixels (paths) per lane: 1, or 2 — with as many jobs as lanes every
// job would have to be in service all the time for the lanes to be full (r6a / r6b: they are not: a third of the jobs wait for
// their shade), so a lane may be the home of TWO pixels: twice the jobs for the same lanes. A column then holds PP x NW jobs:
// job j = row * 64 + i, row = p * NW + w.
template <int Mats, bool Analytic, int NW, int PP>
__global__ __launch_bounds__(NW * 64, 1) void k_path_pooled(RenderParams P, SceneView sc, LaneQueues Q, Counters *cnt,
                                                                   TraceLds cfg, uint32_t sample_end, uint32_t *next_pixel) {
    constexpr uint32_t ROWS = (uint32_t) (NW * PP), NJ = ROWS * 64u, NWQ = ROWS / 4u;
    static_assert(ROWS % 4 == 0 && ROWS >= 4 && ROWS <= 16 && (PP == 1 || PP == 2), "a column's status bytes are one to four whole words");
    constexpr bool Spec8 = true;
    extern __shared__ uint4 smem[];
    const float *thr = stage_thresholds(smem, cfg, Q.log_rec ? Q.log_thr : nullptr);
    stage_tables<false>(sc, cfg, smem);
    U2 *const stacks = reinterpret_cast<U2 *>(smem + cfg.stack16);                 // [entry][job]
    uint4 *const pool = smem + cfg.pool16;                                         // [slot 0..4][job]
    uint32_t *const statw = reinterpret_cast<uint32_t *>(smem + cfg.stat16);       // [NWQ][64]: byte b of word k of column i = job (4 k + b) * 64 + i
    const uint32_t li = threadIdx.x & 63u, wv = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // (wv: wave-uniform, in an SGPR)
    uint32_t *const colw = statw + li;                                             // this lane's column: colw[k * 64]
    for (uint32_t p = 0; p < (uint32_t) PP; ++p) {
        const uint32_t row = p * NW + wv;
        reinterpret_cast<volatile uint8_t *>(statw)[((row >> 2) * 64u + li) * 4u + (row & 3u)] = (uint8_t) PJ_C;   // with its home lane
    }
    __syncthreads();

    GlobalU4 tris_g = (GlobalU4) reinterpret_cast<uintptr_t>(sc.tris), nodes8_g = (GlobalU4) reinterpret_cast<uintptr_t>(sc.nodes8);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(nodes8_g)); asm volatile("" : "+s"(tris_g));            // (phased_kernel.h: MIW_PIN_TREE_PTRS)
#endif
    auto node8_at = [nodes8_g](uint32_t i) -> Bvh8Node {                           // five 16-byte requests
        GlobalU4 p = nodes8_g + 5 * (size_t) i;
        miw_u4 q[5] = { p[0], p[1], p[2], p[3], p[4] };
        Bvh8Node n; __builtin_memcpy(&n, q, sizeof n); return n;
    };
    const GlobalTris tri_at_g{ tris_g };
    const PrimCtx ctx = prim_ctx(sc);
    LaneCounters local; local.segments = local.samples = local.shadow_rays = 0;

    QueueWork<false, true, false, false> work; work.Q = &Q; work.next_pixel = next_pixel; work.n_lanes = P.n_lanes; work.spp = P.spp; work.lane = 0; work.warn_negative = P.film.warn_negative;
    work.film = &P.film; work.thr = thr; work.init_queues(1u);
    __shared__ uint32_t s_prog[NW];
    if (cfg.tail_prio) work.enable_tail_prio(sample_end, &s_prog[wv]);
    // what a pixel's home lane keeps in registers: the path state between two scene queries
    struct Home { LaneRegs L; ShadowOut sh; uint32_t pixel, qlane, mode; bool have, dead_pending; };
    auto home_init = [](Home &H) {
        H.L.flags = LF_DONE; H.L.sample_idx = 0; H.L.rng.state = 0; H.L.rng.inc = MIW_PCG32_SCALAR_INC;
        H.pixel = 0; H.qlane = 0; H.mode = PM_SHADE; H.have = false; H.dead_pending = false;
        H.sh.has = false; H.sh.d = v3(0.f); H.sh.maxt = -1.f; H.sh.c = spec(0.f);
    };
    Home H0, H1; home_init(H0); home_init(H1);
    if (PP == 1) H1.mode = PM_OUT;

#if defined(MIW_PHASE_STATS)
    unsigned long long ps_runs[5] = { 0, 0, 0, 0, 0 }, ps_lanes[5] = { 0, 0, 0, 0, 0 }, ps_cycles[5] = { 0, 0, 0, 0, 0 }, ps_claims[2] = { 0, 0 }, ps_t0 = __builtin_amdgcn_s_memtime();
    // every interval between two stamps is charged to the bucket named at its END: 0 vote, 1 node trip, 2 triangle trip, 3 shade, 4 idle
#define MIW_PP(k, lanes_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ps_runs[k]++; ps_lanes[k] += (unsigned) (lanes_); ps_cycles[k] += now_ - ps_t0; ps_t0 = now_; } while (0)
#define MIW_PP_CLAIM(tried_, won_) do { ps_claims[0] += (unsigned) (tried_); ps_claims[1] += (unsigned) (won_); } while (0)
#else
#define MIW_PP(k, lanes_) do { } while (0)
#define MIW_PP_CLAIM(tried_, won_) do { } while (0)
#endif
    auto count = [](bool p) -> int { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); };
    // (the column snapshot is a register VECTOR indexed by compile-time constants: an array indexed inside these lambdas ends up in scratch)
    auto col_read = [colw](miw_u4 &sw) {
#pragma unroll
        for (uint32_t k = 0; k < NWQ; ++k) sw[k] = __hip_atomic_load(colw + k * 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto status_of = [](const miw_u4 &a, uint32_t row) -> uint32_t {     // the status byte of row `row` (wave-uniform) in a snapshot
        uint32_t r = a[0];
#pragma unroll
        for (uint32_t q = 1; q < NWQ; ++q) r = (row >> 2) == q ? a[q] : r;
        return (r >> (8u * (row & 3u))) & 0xffu;
    };
    auto lds_fence = []() {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    };
    // hand a job back: `st` = its new status byte (no claimed bit); the record words were stored before this call
    auto release = [statw, li, &lds_fence](uint32_t j, uint32_t st) {
        lds_fence();
        const uint32_t row = j >> 6;
        reinterpret_cast<volatile uint8_t *>(statw)[((row >> 2) * 64u + li) * 4u + (row & 3u)] = (uint8_t) st;
    };
    // the claimed bit of row `row` of this lane's column, atomically: -> the status byte the job had (with the bit: somebody else holds it)
    auto take = [colw](uint32_t row) -> uint32_t {
        const uint32_t sh_ = 8u * (row & 3u);
        const uint32_t old = __hip_atomic_fetch_or(colw + (row >> 2) * 64u, PJ_C << sh_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::: "memory");
#endif
        return (old >> sh_) & 0xffu;
    };
    // take a job of this lane's column that is ready for the body `want` (PJ_N / PJ_T) and unclaimed: one of its own pixels' first, else the
    // first / last ready one (`from_top` alternates between lanes and trips, so that no wavefront's jobs are systematically served last).
    // -> job index or MIW_POOL_NOJOB; `st` = the status byte it had.
    auto claim = [&](miw_u4 &sw, uint32_t want, bool from_top, uint32_t &st) -> uint32_t {
        const uint32_t shift = want == PJ_N ? 0u : 1u;
        uint32_t row = MIW_POOL_NOJOB;
        if ((status_of(sw, wv) & (PJ_C | want)) == want) row = wv;
        else if (PP == 2 && (status_of(sw, NW + wv) & (PJ_C | want)) == want) row = NW + wv;
        else {
            uint32_t lo = MIW_POOL_NOJOB, hi = MIW_POOL_NOJOB;
#pragma unroll
            for (uint32_t q = 0; q < NWQ; ++q) {
                const uint32_t m = (sw[q] >> shift) & ~(sw[q] >> 7) & 0x01010101u;
                if (m != 0u) {
                    if (lo == MIW_POOL_NOJOB) lo = 4u * q + ((uint32_t) __builtin_ctz(m) >> 3);
                    hi = 4u * q + ((31u - (uint32_t) __builtin_clz(m)) >> 3);
                }
            }
            row = from_top ? hi : lo;
        }
        if (row == MIW_POOL_NOJOB) return MIW_POOL_NOJOB;
        const uint32_t ob = take(row);
#pragma unroll
        for (uint32_t q = 0; q < NWQ; ++q) sw[q] |= (row >> 2) == q ? PJ_C << (8u * (row & 3u)) : 0u;   // (the caller's snapshot: this lane's next claim looks elsewhere)
        if (ob & PJ_C) return MIW_POOL_NOJOB;                      // somebody else was faster
        if (!(ob & want)) {                                       // mine, but no longer in the state this body serves: give it back as it is
            __hip_atomic_fetch_and(colw + (row >> 2) * 64u, ~(PJ_C << (8u * (row & 3u))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return MIW_POOL_NOJOB;
        }
        st = ob;
        return row * 64u + li;
    };
    struct SignRay { V3 inv_d; };                                  // (walk8_begin reads the octant = the signs of 1 / d = the signs of d: fast_ray keeps them)
    // A walk of job j is over (`w` empty). If it was the shadow walk and an extension ray waits in slot 4, the record becomes the
    // extension walk and the job stays node-ready (-> true, status bits in sb); else the job is done (-> false).
    auto turn_or_finish = [pool](uint32_t j, uint32_t &sb, Walk8 &w, float &tmax, V3 &d_out) -> bool {
        if (!(sb & PJ_S)) return false;
        sb &= ~PJ_S;
        const uint4 s4 = pool[4u * NJ + j];
        const float maxtE = u2f(s4.w);
        if (!(maxtE >= 0.f)) return false;                         // a dead path's last shadow ray: no extension ray
        const V3 d = v3(u2f(s4.x), u2f(s4.y), u2f(s4.z));
        pool[1u * NJ + j] = s4;                                    // the ray of the walk in progress: (dE, maxtE)
        SignRay sg{ d }; walk8_begin(w, sg);
        tmax = maxtE; d_out = d;
        uint2 zw; zw.x = f2u(maxtE); zw.y = MIW_MISS;
        reinterpret_cast<uint2 *>(pool + 3u * NJ + j)[1] = zw;     // slot 3 .zw: tmax, best.tri
        uint4 b; b.x = f2u(MIW_INFINITY); b.y = 0u; b.z = 0u; b.w = 0u;
        pool[4u * NJ + j] = b;                                     // best.t, u, v
        return true;
    };
    // is the pixel H (status row `row`) ready for its shade in the snapshot sw?
    auto shade_ready = [&status_of](const Home &H, const miw_u4 &sw, uint32_t row) -> bool {
        return H.mode == PM_SHADE || (H.mode == PM_WALK && (status_of(sw, row) & PJ_STATE) == PJ_D);
    };
    // ---------------- shade: everything between two scene queries, for the pixel H of this lane (its job: row `row` of the column) ----------------
    auto shade_path = [&](Home &H, uint32_t row) {
        const uint32_t me = row * 64u + li;
        LaneRegs &L = H.L; ShadowOut &sh = H.sh;
        bool occl = false;
        F4 hitE; hitE.x = MIW_INFINITY; hitE.y = hitE.z = 0.f; hitE.w = u2f(MIW_MISS);
        V3 o = v3(0.f);
        if (H.mode == PM_WALK) {                               // take the finished job back
            const uint32_t ob = take(row);
            if ((ob & PJ_STATE) != PJ_D) return;               // (a lane that looked at it a moment ago holds it for an instant; next vote)
            const uint4 s0 = pool[0u * NJ + me], s1 = pool[1u * NJ + me], s3 = pool[3u * NJ + me], s4 = pool[4u * NJ + me];
            o = v3(u2f(s0.x), u2f(s0.y), u2f(s0.z));
            L.ray.d = v3(u2f(s1.x), u2f(s1.y), u2f(s1.z));
            hitE.x = u2f(s3.z); hitE.y = u2f(s4.y); hitE.z = u2f(s4.z); hitE.w = u2f(s3.w);   // best.t = tmax whenever best.tri is a hit
            occl = (ob & PJ_O) != 0u;
        }
        work.lane = H.qlane;
        if (!(L.flags & LF_DONE)) {
            if (sh.has && !occl) L.res = L.res + sh.c;          // path.cpp:171 of the previous vertex
            sh.has = false;
            int rstep = STEP_FINISHED;
            if (!H.dead_pending) rstep = path_step<Mats, Analytic>(P, sc, L, hitE, [o]() { return o; }, sh, &local);
            if (!H.dead_pending && rstep == STEP_DEAD_PENDING) H.dead_pending = true;   // one more job for its shadow ray
            else if (H.dead_pending || rstep == STEP_FINISHED) {
                H.dead_pending = false;
                auto sink = [&work](uint32_t px, uint32_t sample_idx, V2 pos, const float *aovs) { work.put(px, sample_idx, pos, aovs); };
                lane_finish_sample(P, H.pixel, L, sink);
                local.samples++;
                L.flags = 0;
                lane_begin_sample(P, H.pixel, L, sample_end);
            }
        }
        while (L.flags & LF_DONE) {                                 // pixel finished (or no pixel yet): take the next one
            if (H.have) {
                U4 st; st.x = (uint32_t) L.rng.state; st.y = (uint32_t) (L.rng.state >> 32);
                st.z = L.sample_idx >= P.spp ? (uint32_t) LF_DONE : 0u; st.w = L.sample_idx;
                work.store(st);
            }
            U4 st;
            H.have = work.fetch(H.pixel, st);
            if (!H.have) break;
            L.rng.state = (uint64_t) st.x | ((uint64_t) st.y << 32);
            L.sample_idx = st.w; L.flags = 0;
            lane_begin_sample(P, H.pixel, L, sample_end);
        }
        H.qlane = work.lane;
        if (L.flags & LF_DONE) { H.mode = PM_OUT; return; }         // (its byte keeps the claimed bit: with its home lane for good)
        // post the job: the shadow walk first when a shadow ray is queued, then the extension walk (dead_pending: the shadow walk only)
        H.mode = PM_WALK;
        const bool hasS = sh.has;
        const V3 d0 = hasS ? sh.d : L.ray.d;
        const float maxt0 = hasS ? sh.maxt : L.ray.maxt;
        Walk8 w; SignRay sg{ d0 }; walk8_begin(w, sg);
        uint4 s;
        s.x = f2u(L.ray.o.x); s.y = f2u(L.ray.o.y); s.z = f2u(L.ray.o.z); s.w = f2u(L.ray.mint); pool[0u * NJ + me] = s;
        s.x = f2u(d0.x); s.y = f2u(d0.y); s.z = f2u(d0.z); s.w = f2u(maxt0); pool[1u * NJ + me] = s;
        s.x = w.gb; s.y = w.gm; s.z = w.tb; s.w = w.tm; pool[2u * NJ + me] = s;
        s.x = w.tb2; s.y = w.tm2; s.z = f2u(maxt0); s.w = MIW_MISS; pool[3u * NJ + me] = s;
        if (hasS) { s.x = f2u(L.ray.d.x); s.y = f2u(L.ray.d.y); s.z = f2u(L.ray.d.z); s.w = f2u(H.dead_pending ? -1.f : L.ray.maxt); }
        else { s.x = f2u(MIW_INFINITY); s.y = 0u; s.z = 0u; s.w = 0u; }
        pool[4u * NJ + me] = s;
        release(me, PJ_N | (hasS ? PJ_S : 0u));
    };

    // the vote's constants (host: mi_render; MIW_POOL_VOTE overrides): a wavefront shades the pixels p of its lanes once `shade_min` of them wait
    // for it (three quarters of those that still have a pixel, when fewer do), or when fewer than `walk_min` of its lanes find walk work; a walk loop
    // hands over once fewer than node_min (of 128: two jobs per lane) / tri_min (of 64) of its slots hold a job, and looks for new jobs only when
    // at least claim_min of a slot's lanes are empty (the claim code runs for the whole wavefront whoever needs it)
    const int shade_min = (int) cfg.shade_num, walk_min = (int) cfg.shade_den;
    const int node_min = (int) cfg.node_exit, tri_min = (int) cfg.tri_exit, claim_min = (int) cfg.pool_claim_min;
    uint32_t trip = 0;
    for (;;) {
        // ---- the vote ----
        miw_u4 sw = { 0u, 0u, 0u, 0u };
        col_read(sw);
        const bool e_shade0 = shade_ready(H0, sw, wv), e_shade1 = PP == 2 && shade_ready(H1, sw, NW + wv);
        uint32_t any_n = 0u, any_t = 0u;
#pragma unroll
        for (uint32_t k = 0; k < NWQ; ++k) { const uint32_t free_ = ~(sw[k] >> 7) & 0x01010101u; any_n |= sw[k] & free_; any_t |= (sw[k] >> 1) & free_; }
        const int n_node = count(any_n != 0u), n_leaf = count(any_t != 0u), n_shade0 = count(e_shade0), n_shade1 = PP == 2 ? count(e_shade1) : 0;
        const int n_out0 = count(H0.mode == PM_OUT), n_out1 = PP == 2 ? count(H1.mode == PM_OUT) : 64;
        if (n_out0 == 64 && n_out1 == 64) break;
        const int lead = n_node > n_leaf ? n_node : n_leaf;
        auto thr_of = [shade_min](int n_out) -> int { const int q = (3 * (64 - n_out) + 3) / 4; return shade_min < q ? shade_min : q; };
        const int thr0 = thr_of(n_out0), thr1 = thr_of(n_out1);
        const bool starved = lead < walk_min;
        const bool go0 = n_shade0 > 0 && (n_shade0 >= thr0 || starved), go1 = n_shade1 > 0 && (n_shade1 >= thr1 || starved);
        MIW_PP(0, 0);

        if (go0 || go1) {
            work.tick(H0.L.sample_idx, H0.mode != PM_OUT && !(H0.L.flags & LF_DONE));
            if (go0) { if (e_shade0) shade_path(H0, wv); MIW_PP(3, n_shade0); }
            if (PP == 2 && go1) { if (e_shade1) shade_path(H1, NW + wv); MIW_PP(3, n_shade1); }
        } else if (n_node >= n_leaf && n_node > 0) {
            // ---------------- node steps on the column's node-ready jobs: every lane works on up to TWO of them, so that two node fetches
            // (2 x five 16-byte requests) are in flight per lane while it computes — a job record in LDS is what lets one lane hold two
            // walks (k_path_phased: one walk per lane, tied to its pixel) ----------------
            struct NodeSlot { uint32_t j, sb; Walk8 w; FastRay r; float tmax; Bvh8Node nd; };
            NodeSlot A, B;
            A.j = B.j = MIW_POOL_NOJOB; A.sb = B.sb = 0u; A.tmax = B.tmax = 0.f;
            A.w.gb = A.w.gm = A.w.tb = A.w.tm = A.w.tb2 = A.w.tm2 = 0u; B.w = A.w;
            A.r.inv_d = A.r.neg_o_inv_d = v3(0.f); A.r.mint = 0.f; B.r = A.r;
            auto store_walk = [&](const NodeSlot &t) {
                uint4 q; q.x = t.w.gb; q.y = t.w.gm; q.z = t.w.tb; q.w = t.w.tm; pool[2u * NJ + t.j] = q;
                uint2 h; h.x = t.w.tb2; h.y = t.w.tm2; reinterpret_cast<uint2 *>(pool + 3u * NJ + t.j)[0] = h;
            };
            // an empty slot takes a node-ready job of the column (when enough lanes of the slot are empty to pay for the attempt);
            // a slot that holds one issues the fetch of its next node
            auto refill = [&](NodeSlot &t, bool from_top, bool may_claim) {
                if (may_claim && count(t.j == MIW_POOL_NOJOB) >= claim_min) {
                    if (t.j == MIW_POOL_NOJOB) {
                        uint32_t st = 0u;
                        const uint32_t got = claim(sw, PJ_N, from_top, st);
                        MIW_PP_CLAIM(1, got != MIW_POOL_NOJOB);
                        if (got != MIW_POOL_NOJOB) {
                            t.j = got; t.sb = st & (PJ_S | PJ_O);
                            const uint4 s0 = pool[0u * NJ + got], s1 = pool[1u * NJ + got], s2 = pool[2u * NJ + got], s3 = pool[3u * NJ + got];
                            t.r = fast_ray(v3(u2f(s0.x), u2f(s0.y), u2f(s0.z)), v3(u2f(s1.x), u2f(s1.y), u2f(s1.z)), u2f(s0.w));
                            t.w.gb = s2.x; t.w.gm = s2.y; t.w.tb = s2.z; t.w.tm = s2.w; t.w.tb2 = s3.x; t.w.tm2 = s3.y; t.tmax = u2f(s3.z);
                        }
                    }
                }
                if (t.j != MIW_POOL_NOJOB) t.nd = node8_at(walk8_next_node(t.w));
            };
            // the slot's node has arrived: one node step; a job that is no longer node-ready goes back to the pool (or turns into its extension walk)
            auto advance = [&](NodeSlot &t) {
                if (t.j == MIW_POOL_NOJOB) return;
                walk8_node_step<Spec8>(t.nd, t.r, widen(t.tmax), t.w, PoolColumn8<(int) NJ>{ stacks + t.j });
                if (walk8_node_ready<Spec8>(t.w)) return;
                if (walk8_over(t.w)) {
                    V3 dE;
                    if (turn_or_finish(t.j, t.sb, t.w, t.tmax, dE)) { const uint4 s0 = pool[0u * NJ + t.j]; t.r = fast_ray(v3(u2f(s0.x), u2f(s0.y), u2f(s0.z)), dE, u2f(s0.w)); return; }
                    store_walk(t); release(t.j, PJ_D | t.sb);
                } else { store_walk(t); release(t.j, (walk8_tri_ready(t.w) ? PJ_T : 0u) | t.sb); }   // (not node-ready and not over: it holds triangles)
                t.j = MIW_POOL_NOJOB;
            };
            refill(A, (li & 1u) != 0u, true); refill(B, (li & 1u) == 0u, true);
            for (;;) {
                ++trip;
                const int now = count(A.j != MIW_POOL_NOJOB) + count(B.j != MIW_POOL_NOJOB);
                if (now == 0) break;
                advance(A);
                col_read(sw);
                // the home lanes whose walks are over want their shade: leave once enough of them wait
                const int ns0 = count(shade_ready(H0, sw, wv)), ns1 = PP == 2 ? count(shade_ready(H1, sw, NW + wv)) : 0;
                const bool leave = now < node_min || ns0 >= thr0 || (PP == 2 && ns1 >= thr1);   // (every claimed job got its step: a vote that finds few takers still makes progress)
                refill(A, ((li ^ trip) & 1u) != 0u, !leave);
                advance(B);
                refill(B, ((li ^ trip) & 1u) == 0u, !leave);
                MIW_PP(1, now);
                if (leave) break;
            }
            if (A.j != MIW_POOL_NOJOB) { store_walk(A); release(A.j, PJ_N | (walk8_tri_ready(A.w) ? PJ_T : 0u) | A.sb); }
            if (B.j != MIW_POOL_NOJOB) { store_walk(B); release(B.j, PJ_N | (walk8_tri_ready(B.w) ? PJ_T : 0u) | B.sb); }
        } else if (n_leaf > 0) {
            // ---------------- triangle tests on the column's triangle-ready jobs ----------------
            uint32_t j = MIW_POOL_NOJOB, sb = 0u;
            Walk8 w; w.gb = w.gm = w.tb = w.tm = w.tb2 = w.tm2 = 0u;
            V3 o = v3(0.f), d = v3(0.f); float mint = 0.f, maxt = 0.f, tmax = 0.f;
            Hit best; best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
            auto store_walk = [&]() {
                uint4 s; s.x = w.gb; s.y = w.gm; s.z = w.tb; s.w = w.tm; pool[2u * NJ + j] = s;
                s.x = w.tb2; s.y = w.tm2; s.z = f2u(tmax); s.w = best.tri; pool[3u * NJ + j] = s;
                if (!(sb & PJ_S)) { s.x = f2u(best.t); s.y = f2u(best.u); s.z = f2u(best.v); s.w = 0u; pool[4u * NJ + j] = s; }   // (a shadow walk's slot 4 holds the extension ray)
            };
            for (;;) {
                ++trip;
                if (count(j == MIW_POOL_NOJOB) >= claim_min || trip == 0u) {
                    if (j == MIW_POOL_NOJOB) {
                        uint32_t st = 0u;
                        const uint32_t got = claim(sw, PJ_T, ((li ^ trip) & 1u) != 0u, st);
                        MIW_PP_CLAIM(1, got != MIW_POOL_NOJOB);
                        if (got != MIW_POOL_NOJOB) {
                            j = got; sb = st & (PJ_S | PJ_O);
                            const uint4 s0 = pool[0u * NJ + j], s1 = pool[1u * NJ + j], s2 = pool[2u * NJ + j], s3 = pool[3u * NJ + j];
                            o = v3(u2f(s0.x), u2f(s0.y), u2f(s0.z)); mint = u2f(s0.w); d = v3(u2f(s1.x), u2f(s1.y), u2f(s1.z)); maxt = u2f(s1.w);
                            w.gb = s2.x; w.gm = s2.y; w.tb = s2.z; w.tm = s2.w; w.tb2 = s3.x; w.tm2 = s3.y; tmax = u2f(s3.z);
                            best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = s3.w;
                            if (!(sb & PJ_S)) { const uint4 s4 = pool[4u * NJ + j]; best.t = u2f(s4.x); best.u = u2f(s4.y); best.v = u2f(s4.z); }
                        }
                    }
                }
                const int now = count(j != MIW_POOL_NOJOB);
                if (now == 0) break;
                if (j != MIW_POOL_NOJOB) {
                    bool occluded = false;
                    walk8_tri_step<Analytic, Spec8>(tri_at_g, ctx, o, d, mint, maxt, (sb & PJ_S) != 0u, best, tmax, occluded, w);
                    if (occluded) sb |= PJ_O;
                    if (!walk8_tri_ready(w)) {
                        if (walk8_over(w)) {
                            store_walk();                              // (before the turn re-initialises tmax / best.tri in the record)
                            V3 dE; Walk8 w2 = w; float tmax2 = tmax;
                            const bool turned = turn_or_finish(j, sb, w2, tmax2, dE);
                            if (turned) { uint4 s; s.x = w2.gb; s.y = w2.gm; s.z = w2.tb; s.w = w2.tm; pool[2u * NJ + j] = s;
                                          uint2 t; t.x = w2.tb2; t.y = w2.tm2; reinterpret_cast<uint2 *>(pool + 3u * NJ + j)[0] = t; }
                            release(j, (turned ? PJ_N : PJ_D) | sb);
                        } else { store_walk(); release(j, PJ_N | sb); }   // (no triangles left and not over: node groups are pending)
                        j = MIW_POOL_NOJOB;
                    }
                }
                MIW_PP(2, now);
                if (now < tri_min) break;
                col_read(sw);
                const int ns0 = count(shade_ready(H0, sw, wv)), ns1 = PP == 2 ? count(shade_ready(H1, sw, NW + wv)) : 0;
                if (ns0 >= thr0 || (PP == 2 && ns1 >= thr1)) break;
            }
            if (j != MIW_POOL_NOJOB) { store_walk(); release(j, PJ_T | (walk8_node_ready<Spec8>(w) ? PJ_N : 0u) | sb); }
        } else {
            // every job of this wavefront's lanes is in another wavefront's hands right now
            __builtin_amdgcn_s_sleep(8);
            MIW_PP(4, 0);
        }
    }

#if defined(MIW_PHASE_STATS)
    if (li == 0) {
        for (int k = 0; k < 5; ++k) { atomicAdd(&g_pool_stats[k], ps_runs[k]); atomicAdd(&g_pool_stats[5 + k], ps_lanes[k]); atomicAdd(&g_pool_stats[10 + k], ps_cycles[k]); }
    }
    { const unsigned long long a_ = wave_sum(ps_claims[0]), b_ = wave_sum(ps_claims[1]); if (li == 0) { atomicAdd(&g_pool_stats[15], a_); atomicAdd(&g_pool_stats[16], b_); } }
#endif
    unsigned long long a = wave_sum(local.segments), b = wave_sum(local.samples), c = wave_sum(local.shadow_rays);
    if (li == 0) {
        Counters *shard = cnt + ((blockIdx.x * NW + wv) & (MIW_CNT_SHARDS - 1));
        if (a) atomicAdd(&shard->segments, a);
        if (b) atomicAdd(&shard->samples, b);
        if (c) atomicAdd(&shard->shadow_rays, c);
    }
}
#undef MIW_PP
#undef MIW_PP_CLAIM
