// Plan 2 (resident): k_init_pixels, the shared pixel queue and k_path_resident (path / direct integrators).
// Part of the single translation unit csrc/miwave.hip (included there, in this order; not a stand-alone header).
// ---- the resident plan -----------------------------------------------------------------
// k_init_pixels: pixel <-> lane map and PCG32 seeding only (a pixel carries nothing else
// between two camera samples).
__global__ __launch_bounds__(MIW_BLOCK) void k_init_pixels(RenderParams P, U4 *st_out, uint32_t *pixel_out, InitArgs A) {
    uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= P.n_lanes) return;
    uint32_t tile = lane >> A.bs2_log2, i = lane & ((1u << A.bs2_log2) - 1u);
    uint32_t b = A.tile_list ? A.tile_list[tile] : tile;
    uint32_t bx = b % A.blocks_x, by = b / A.blocks_x;
    uint32_t x, y;
    morton_decode2(i, x, y);                                   // integrator.cpp:200
    int32_t bw = P.film.crop_w - (int32_t) (bx * A.bs), bh = P.film.crop_h - (int32_t) (by * A.bs);
    if (bw > (int32_t) A.bs) bw = (int32_t) A.bs;
    if (bh > (int32_t) A.bs) bh = (int32_t) A.bs;
    if ((int32_t) x >= bw || (int32_t) y >= bh) {                // :201-202 — pixel outside the block
        U4 st; st.x = st.y = 0; st.z = LF_DONE; st.w = 0;
        st_out[lane] = st; pixel_out[lane] = 0;
        return;
    }
    uint32_t px = (uint32_t) P.film.crop_x + bx * A.bs + x, py = (uint32_t) P.film.crop_y + by * A.bs + y;
    pixel_out[lane] = px | (py << 16);
    st_out[lane] = lane_seed_state(A.base_seed + (uint64_t) A.block_ids[b] * (uint64_t) (A.bs * A.bs) + i);   // :198
}

// k_path_resident: one thread = one pixel, advanced from its current sample to `sample_end`.
// Path state, ray, hit record and the pending emitter contribution never leave registers;
// the geometry is swept / walked in LDS (trace_one); HBM sees 20 B of pixel state per launch
// and the 24 B/sample log (or the film atomics).
struct TileArgs {               // film_mode 2 in the resident plan; side == 0: splat straight into `accum`
    const uint32_t *tile_list; uint32_t blocks_x, bs, bs2_log2;
    uint32_t side;              // 16 + 2 * margin, margin = max(filter border, floor(radius + .5))
    uint32_t geom16;            // uint4 slots of dynamic LDS in front of the tile
};

// Log mode feeds the lanes from ONE shared pixel queue (`next_pixel`): a lane that finishes its pixel takes
// the next unclaimed one inside the iteration loop (wavefront-aggregated: ballot + one atomic per wave per
// refill), so no lane waits for the slowest pixel of its wavefront and the launch drains evenly. The grid is
// sized to the machine; workgroups that start late find the queue empty and retire.
#ifndef MIW_PLACE_PIECES
#define MIW_PLACE_PIECES 4
#endif
// Placed = false (the full-frame instantiations of the tree kernels) compiles the queue choice, the dry-queue scan and the cost
// clock out — four registers less carried through every body of the phase machine; shards of about one pixel per resident lane
// get the Placed = true instantiation of their kernel. Clock: a pixel's cost is the time it held its lane (s_memtime, the phase
// machine: its lanes do not iterate together) instead of the wavefront's iterations while it did (the lock-step packet kernel).
// Groups: store() counts finished pixels per group of 64 tiles (the film replay beside the render, miwave.hip: overlap_prepare) — compiled into ONE more
// instantiation of the plain-diffuse packet kernel, launched when that replay is asked for (the only path kernel a replay wavefront fits beside; the default
// kernels stay as they were: the statements cost the packet kernel ~1 ms of its 247 even when switched off, the phase machine 16 more spilled registers).
// Jobs: chunk jobs (below) compiled in — the packet kernels always, the phase machine's full-frame instantiations by MIW_PHASED_JOBS (phased_kernel.h).
template <bool Placed = true, bool Clock = false, bool Groups = false, bool Jobs = true>
struct QueueWork {
    const LaneQueues *Q; uint32_t *next_pixel; uint32_t n_lanes, spp, lane, warn_negative;
    const FilmRec *film; const float *thr;      // 16-byte records (Q->log_rec): the film geometry and the phase thresholds in LDS
    // One queue over all lanes — or, for shards of at most one pixel per resident lane, PLACED queues (nq = one per SIMD of the
    // device): such a launch is one pixel deep, the priorities below make the four wavefronts of a SIMD finish together, and what
    // is left is the imbalance BETWEEN SIMDs (the sum of four random 64-pixel pieces: +-7 %, its maximum over 1024 SIMDs +24 %).
    // So a first short launch measures what every PIXEL costs (Q->lane_cost: over the first eighth of the samples), the device
    // sorts the lanes by it (Q->lane_sorted) and cuts the sorted list into pieces of 64 — consecutive entries (the pixels of a
    // wavefront cost about the same and finish together) or every pieces-th entry (every wavefront gets its share of the dear
    // pixels; LaneQueues::piece_a / piece_b, chosen by mi_render) — round 3 dealt the pieces of the image as they lay — the host
    // deals the pieces to the SIMDs longest-first (Q->piece_list: four pieces per
    // queue, equal sums) and the second launch lets a wavefront take its pixels from the queue of the SIMD it runs on: HW_ID / XCC_ID name the
    // SIMD (Q->simd_ids: marked present by the measuring launch, numbered by the host). A lane whose queue has run dry moves
    // on to the next one (per lane: `q`, `dry`), so no piece can be left behind; the first lane that has found every queue
    // empty raises a flag (simd_ids[0]) that spares the others the scan.
    uint32_t nq, per, q, dry, t_fetch;
    __device__ __forceinline__ void init_queues(uint32_t queues) {
        nq = Placed ? queues : 1u; per = nq > 1u ? MIW_PLACE_PIECES * 64u : n_lanes; q = 0u; dry = 0; t_fetch = 0;
        ticks = 0; quarter = 0; tail_prio = 0; sample_end_ = spp;
        if (Jobs) job_pend() = 0u;                              // (each lane its own word: no barrier)
        if (Placed && Q->simd_ids) {
            // which SIMD this wavefront runs on. The measuring launch only marks the SIMD as present (the host numbers the present
            // ones 0 .. n - 1 afterwards); the placed launch looks its queue up.
            const uint32_t hw = (uint32_t) __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)),          // HW_REG_HW_ID: simd [5:4] cu [11:8] sh [12] se [15:13]
                           xcc = (uint32_t) __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));          // HW_REG_XCC_ID [3:0]
            const uint32_t key = ((xcc & 15u) << 10) | (((hw >> 13) & 7u) << 7) | (((hw >> 12) & 1u) << 6) | (((hw >> 8) & 15u) << 2) | ((hw >> 4) & 3u);
            if (nq > 1u) { const uint32_t id = Q->simd_ids[1u + key]; q = id < nq ? id : key % nq; }
            else if ((threadIdx.x & 63u) == 0u) Q->simd_ids[1u + key] = 1u;
        }
    }
    // Least-progress-first among the wavefronts of a SIMD, for shards with about one pixel per resident lane (tail_prio set
    // by the host): a pixel's samples are one serial PCG32 stream, so such a launch is one pixel deep and lasts as long as its
    // most expensive pixels, which share their SIMD's issue slots evenly with wavefronts that will be done long before them.
    // s_setprio makes the SIMD's arbiter prefer the wavefront that is furthest behind (priority 3 in its first quarter of
    // samples ... 0 in its last): the wavefronts of a SIMD then finish together, at (their total work) / (the SIMD's
    // throughput), instead of the expensive one running on alone at the end. A stalled high-priority wave still yields its
    // slots, so throughput is what it was. Evaluated every 16th iteration (one wave-wide minimum of the lanes' sample counters).
    uint32_t ticks, quarter, tail_prio, sample_end_; uint32_t *prog;      // prog: this wavefront's word in LDS (the wave-wide minimum)
    __device__ __forceinline__ void tick(uint32_t sample_idx, bool has_pixel) {
        if (!tail_prio) { ++ticks; return; }
        // (called from divergent code: the lanes that just fetched a pixel are not here — so no cross-lane shuffles; the
        // counter of the first active lane decides, the minimum goes through one LDS atomic per lane)
        if ((uint32_t) __builtin_amdgcn_readfirstlane((int) ++ticks) & 15u) return;
        if (has_pixel) atomicMin(prog, sample_idx);
        const uint32_t v = *prog;
        *prog = 0xffffffffu;
        // quarter of its samples the slowest lane is in: floor(4 v / sample_end) by comparisons (a division by a kernel argument would
        // keep its reciprocal in a VGPR for the whole launch)
        const uint32_t v4 = 4u * v, se = sample_end_;
        uint32_t qtr = v == 0xffffffffu ? 3u : (v4 >= se ? 1u : 0u) + (v4 >= 2u * se ? 1u : 0u) + (v4 >= 3u * se ? 1u : 0u);
        qtr = (uint32_t) __builtin_amdgcn_readfirstlane((int) qtr);
        if (qtr == quarter) return;
        quarter = qtr;
        if (qtr == 0u) __builtin_amdgcn_s_setprio(3); else if (qtr == 1u) __builtin_amdgcn_s_setprio(2);
        else if (qtr == 2u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    __device__ __forceinline__ void enable_tail_prio(uint32_t sample_end, uint32_t *wave_word) {
        tail_prio = 1u; sample_end_ = sample_end; quarter = 0; prog = wave_word; *prog = 0xffffffffu; __builtin_amdgcn_s_setprio(3);
    }
    // ---- chunk jobs (round 6; LaneQueues::job_chunk != 0: full frames, one queue, all samples in one launch) ----
    // Why: a pixel's samples are one serial job (one PCG32 stream per pixel, integrator.cpp:196-209) of ~35 ms at C2, the lanes are not in step, so when the
    // queue runs dry every lane is somewhere inside such a job: the last ~35 ms of the launch run at falling lane occupancy. Measured (gpurun r6o): the path
    // kernel's time is a straight line in the pixel count, 102.4 ns per pixel + 26.5 ms that do not depend on it — 11 % of the headline kernel. A pixel's state is
    // 16 bytes (Q->st), so a pixel can change lanes between two samples: the queue holds (chunk, slot) pairs, chunk-major, every lane draws from it until it is
    // empty, and the tail is as long as the LAST chunk (the chunks halve: LaneQueues::job_min). Chunk j of a slot is ready when the slot's sample counter stands at its first sample — its predecessor
    // is n_lanes jobs back in the queue, complete long ago in a frame of several pixels per lane. If it is not, the lane must not spin here (the lane running
    // the chunk before may be a neighbour in this very wavefront): it parks the job in its word of LDS (id + 1; 0: none; ~0: the queue is empty), reports "nothing
    // now" and asks again on its next trip. The state words cross CUs and XCDs inside one launch: sc1 accesses (past the XCD's L2), the RNG words
    // written first, the counter words after they have landed; the reader takes all four in ONE 16-byte load (never the new counter beside the old RNG words).
    // Cost: one atomic on one address per job — chunks of 8 / 16 / 32 samples ran the frame in 702 / 496 / 305 ms (gpurun r6p; 64 and 128: 226.8 / 226.5,
    // 256: 230.6, jobs of whole pixels: 239.3) — so mi_render cuts chunks of at least 64 samples.
    typedef uint32_t U4v __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ static uint32_t &job_pend() { __shared__ uint32_t s_job_pend[MIW_BLOCK]; return s_job_pend[threadIdx.x & (MIW_BLOCK - 1)]; }
    __device__ __forceinline__ bool exhausted() const { return !Jobs || !Q->job_chunk || job_pend() == 0xffffffffu; }   // after fetch() said false: for good, or "ask again"
    __device__ __forceinline__ uint32_t job_end(uint32_t sample_idx, uint32_t sample_end) const {   // called after a sample has finished: sample_idx >= 1
        const uint32_t r = spp - sample_idx;                                                        // samples still to do: a chunk ends where that is a power of two >= job_min
        return Jobs && (r & (r - 1u)) == 0u && r >= Q->job_min ? sample_idx : sample_end;         // (job_min = 2^31 without chunks; r = 0 at the pixel's end: sample_end decides)
    }
    __device__ __forceinline__ bool fetch_job(uint32_t &pixel, U4 &st) {
        for (;;) {
            const uint32_t parked = job_pend();
            if (parked == 0xffffffffu) return false;
            uint32_t id = parked - 1u;
            const bool fresh = parked == 0u;
            const unsigned long long b = __ballot(fresh);
            if (fresh) {
                const uint32_t me = threadIdx.x & 63u, leader = (uint32_t) __ffsll((long long) b) - 1u;
                uint32_t base = 0;
                if (me == leader) base = atomicAdd(next_pixel, (uint32_t) __popcll(b));
                base = (uint32_t) __shfl((int) base, (int) leader, 64);
                id = base + __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0u));
                if (id >= Q->job_total) { job_pend() = 0xffffffffu; return false; }
            }
            const uint32_t j = id / n_lanes, slot = id - j * n_lanes;
            const uint32_t px = Q->pixel[slot];
            U4v w;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(Q->st + slot) : "memory");
#endif
            if (w.z & LF_DONE) { job_pend() = 0u; continue; }                     // pixel outside its clipped block, or already complete
            if (w.w != (j ? spp - (Q->job_pow >> (j - 1u)) : 0u)) { job_pend() = id + 1u; return false; }   // the chunk before is still running: ask again
            job_pend() = 0u;
            lane = slot; pixel = px;
            st.x = w.x; st.y = w.y; st.z = w.z; st.w = w.w;
            return true;
        }
    }
    __device__ __forceinline__ bool fetch(uint32_t &pixel, U4 &st) {
        if (Jobs && Q->job_chunk) return fetch_job(pixel, st);       // (wave-uniform)
        for (;;) {
            if (!Placed) {                                      // one queue: a ballot, one atomic, the lanes' ranks
                const unsigned long long b = __ballot(1);
                const uint32_t me = threadIdx.x & 63u, leader = (uint32_t) __ffsll((long long) b) - 1u;
                uint32_t base = 0;
                if (me == leader) base = atomicAdd(next_pixel, (uint32_t) __popcll(b));
                base = (uint32_t) __shfl((int) base, (int) leader, 64);
                const uint32_t l = base + __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0u));
                if (l >= n_lanes) return false;
                lane = l;
                st = Q->st[lane];
                if (st.z & LF_DONE) continue;                   // pixel outside its clipped block, or already complete
                pixel = Q->pixel[lane];
                return true;
            }
            if (dry >= nq) {                                    // every queue is empty: say so to everybody (placed queues: a scan of all
                if (nq > 1u) Q->simd_ids[0] = 1u;               // of them costs a thousand round trips — once per launch is enough)
                return false;
            }
            if (nq > 1u && __hip_atomic_load(Q->simd_ids, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { dry = nq; return false; }
            // the lanes asking now are served queue by queue: this trip, those that ask the leader's queue — one atomic for all of them
            const unsigned long long all = __ballot(1);
            const uint32_t me = threadIdx.x & 63u, leader = (uint32_t) __ffsll((long long) all) - 1u;
            const uint32_t q_lead = (uint32_t) __shfl((int) q, (int) leader, 64);
            const bool mine = q == q_lead;
            const unsigned long long b = __ballot(mine);
            if (!mine) continue;
            uint32_t base = 0;
            if (me == leader) base = atomicAdd(next_pixel + q, (uint32_t) __popcll(b));
            base = (uint32_t) __shfl((int) base, (int) leader, 64);
            // rank among the asking lanes: v_mbcnt counts the ballot's bits below this lane (no 64-bit lane mask kept in registers)
            const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0u));
            uint32_t l = idx;
            if (nq > 1u) {                                      // placed queues: queue q = up to MIW_PLACE_PIECES pieces of 64 sorted lanes
                const uint32_t piece = idx < per ? Q->piece_list[q * MIW_PLACE_PIECES + (idx >> 6)] : 0xffffffffu;
                if (piece == 0xffffffffu) { q = q + 1u == nq ? 0u : q + 1u; ++dry; continue; }   // this queue is empty: on to the next
                l = Q->lane_sorted[piece * Q->piece_a + (idx & 63u) * Q->piece_b];
                if (l >= n_lanes) continue;                     // a slot past the end of the partial last piece: ask the same queue again
            } else if (l >= n_lanes) { ++dry; continue; }
            lane = l;
            st = Q->st[lane];
            if (st.z & LF_DONE) continue;                       // pixel outside its clipped block, or already complete
            pixel = Q->pixel[lane];
            if (Placed) t_fetch = cost_clock();
            return true;
        }
    }
    __device__ __forceinline__ uint32_t cost_clock() const {
        return Clock ? (uint32_t) (__builtin_amdgcn_s_memtime() >> 8) : ticks;      // units of 256 shader cycles, or wavefront iterations
    }
    __device__ __forceinline__ void store(U4 st) {
        if (Jobs && Q->job_chunk) {                             // chunk jobs: another lane (another XCD) takes the pixel on inside this launch
            unsigned long long *words = reinterpret_cast<unsigned long long *>(Q->st + lane);
            __hip_atomic_store(words, (unsigned long long) st.x | ((unsigned long long) st.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the RNG words have landed before the counter that publishes them leaves
            __hip_atomic_store(words + 1, (unsigned long long) st.z | ((unsigned long long) st.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        Q->st[lane] = st;
        if (Placed && Q->lane_cost) Q->lane_cost[lane] = cost_clock() - t_fetch + 1u;   // what this pixel cost in this (the measuring) launch
        // film replay beside the render: this pixel's log is complete — count it in its group of 64 tiles; the pixel that completes the group
        // makes everything written before the counts visible and raises the group's flag (mi_render: the replay's stream waits on it)
        if (Groups && Q->group_done && (st.z & LF_DONE)) {        // (nullptr in every launch but the single full-frame one)
            __threadfence();
            const uint32_t g = lane >> Q->group_shift;
            if (atomicAdd(Q->group_done + g, 1u) + 1u == Q->group_expected[g]) {
                __threadfence_system();
                __hip_atomic_store(Q->group_flag + g, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __device__ __forceinline__ void put(uint32_t pixel, uint32_t sample_idx, V2 pos, const float *aovs) {
        if (Q->log_rec) {                                       // wave-uniform: one format per render
            LogSink16<const float *> sink{ Q->log_rec, thr, film, lane, spp, Q->log_rej & 255u, Q->log_rej >> 8 };   // (log_rej: the class count | log_il << 8 — one uniform, the phase machine has no scalar register to spare)
            sink(pixel, sample_idx, pos, aovs);
        } else {
            LogSink sink; sink.log_pos = Q->log_pos; sink.log_val = Q->log_val; sink.lane = lane; sink.spp = spp; sink.warn_negative = warn_negative;
            sink(0u, sample_idx, pos, aovs);
        }
    }
};

#ifndef MIW_DIRECT_WAVES
/* Waves per SIMD the direct-integrator kernels are compiled for. At 2 (256 VGPRs) none of them spills, at 3 (168) they keep
   16 - 108 registers in scratch (60 / 216 in the MATS_ALL ones when this was measured) — and are faster all the same (path kernel of one frame,
   round-3 session I: Cornell box 92.7 ms against 115.2, material balls 207 / 204, 0.9 M-triangle interior 174 / 227): the
   spills sit in the prologue and around the BSDF code, the third wave hides the query's latency. */
#define MIW_DIRECT_WAVES 3
#endif
// Analytic: the scene holds analytic shapes (rectangles); packet scenes (Tiny) never do.
// Integ: which SamplingIntegrator::sample the pixel loop runs (path.h / direct.h).
// Wavefronts per SIMD the packet kernels are compiled for = workgroups per CU mi_render launches of them. Round 6 (gpurun r6n): the plain-diffuse kernel — the
// headline's — at FIVE (96 registers; 12 values spilled, stored once in front of the pixel loop and reloaded at a dozen places of its body, none inside the box or
// candidate loops): 248.1 -> 238.3 ms at C2, the film bit-identical — the kernel's throughput follows its resident wavefronts (32 workgroups short of the device cost
// it 2.4 %, r6l), and the spills that made a fifth wavefront 21 - 30 % SLOWER in round 3 went away with the register diets of rounds 3 - 5. Six: 81 spilled, no.
// (Five is for the kernel with 32-bit candidate masks, Tiny == 2: scenes of <= 32 triangles; its 64-bit twin, Tiny == 1, would spill 45 at 96 registers
// and stays at four, unmeasured at five.)
// The other scalar_rgb packet kernels (BSDF dispatch, textures) fit 128 registers without scratch: four instead of three. The spectral ones keep 93 values in scratch at 128
// registers — and are 10.5 % faster there all the same (C5: path kernel 575.4 -> 519.2 ms, 1 790 -> 1 978 Msamples/s, gpurun r6u): four as well.
#ifndef MIW_PACKET_WAVES_ALL
#define MIW_PACKET_WAVES_ALL 4
#endif
#ifndef MIW_PACKET_WAVES
#define MIW_PACKET_WAVES 5
#endif
// Waves != 0 names the wavefronts per SIMD outright: <..., 4> of the plain-diffuse kernel is the 128-register form (no scratch) that a SHARD of at most four
// wavefronts' worth of pixels per SIMD gets — a rank's eighth of a 1080p frame is 4 050 wavefronts for 1 024 SIMDs: a fifth wave slot would stay empty and the
// 96-register squeeze costs each wavefront 3.4 % (r6n: 256.4 vs 248.0 ms with four workgroups per CU).
template <bool UseLog, int Tiny, int Mats = MATS_ALL, bool Analytic = (Tiny == 0), uint32_t Integ = INTEG_PATH, bool Groups = false, int Waves = 0>
__global__ __launch_bounds__(MIW_BLOCK, Waves ? Waves : Integ == INTEG_DIRECT ? MIW_DIRECT_WAVES : Tiny ? ((Mats == MATS_DIFFUSE && !MIW_SPECTRAL) ? (Tiny == 2 ? MIW_PACKET_WAVES : 4) : MIW_PACKET_WAVES_ALL) : MIW_TREE_WAVES) void k_path_resident(RenderParams P, SceneView sc, LaneQueues Q, double *accum, Counters *cnt,
                                                               TraceLds cfg, uint32_t sample_end, TileArgs T, uint32_t *next_pixel) {
    extern __shared__ uint4 smem[];
    stage_to_lds(sc, cfg, smem);
    const float *thr = stage_thresholds(smem, cfg, UseLog && Q.log_rec ? Q.log_thr : nullptr);
#if MIW_LDS_TABLES
    if constexpr (UseLog && Tiny != 0) stage_tables<true>(sc, cfg, smem);
#endif
#if defined(MIW_SECTION_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if ((threadIdx.x & 63u) == 0) { unsigned long long *b_ = miw_sec_buf(); for (int i = 0; i < 15; ++i) b_[i] = 0; b_[15] = __builtin_amdgcn_s_memtime(); }
#endif
    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    // workgroup film tile
    double *tile = reinterpret_cast<double *>(smem + T.geom16);
    int tile_x0 = 0, tile_y0 = 0;
    if (!UseLog && T.side) {
        const uint32_t lane0 = blockIdx.x * blockDim.x, t = lane0 >> T.bs2_log2, q0 = lane0 & ((1u << T.bs2_log2) - 1u);
        const uint32_t b = T.tile_list ? T.tile_list[t] : t;
        uint32_t qx, qy;
        morton_decode2(q0, qx, qy);
        const int margin = (int) (T.side - 16u) / 2;
        tile_x0 = (int) ((b % T.blocks_x) * T.bs + qx) - margin;
        tile_y0 = (int) ((b / T.blocks_x) * T.bs + qy) - margin;
        for (uint32_t i = threadIdx.x; i < T.side * T.side * MIW_FILM_CHANNELS; i += blockDim.x) tile[i] = 0.0;
        __syncthreads();
    }
    LaneCounters local; local.segments = local.samples = local.shadow_rays = 0;
    auto tr2 = [&](V3 o, float mint, V3 dE, float maxtE, bool hasE, V3 dS, float maxtS, bool hasS, F4 &hE, bool &occS) {
        trace2<Tiny, Analytic>(sc, cfg, smem, o, mint, dE, maxtE, hasE, dS, maxtS, hasS, hE, occS);
    };
    if (UseLog) {
        QueueWork<Tiny != 0, false, Groups> work; work.Q = &Q; work.next_pixel = next_pixel; work.n_lanes = P.n_lanes; work.spp = P.spp; work.lane = 0; work.warn_negative = P.film.warn_negative;
        work.film = &P.film; work.thr = thr; work.init_queues(cfg.queues ? cfg.queues : 1u);
        __shared__ uint32_t s_prog[MIW_BLOCK / 64];
        if (cfg.tail_prio) work.enable_tail_prio(sample_end, &s_prog[threadIdx.x >> 6]);
        if constexpr (Integ == INTEG_DIRECT) pixel_stream_render_direct<Mats, Analytic>(P, sc, sample_end, work, tr2, &local);
        else pixel_stream_render<Mats, Analytic>(P, sc, sample_end, work, tr2, &local);
    } else if (lane < P.n_lanes) {
        U4 st = Q.st[lane];
        if (!(st.z & LF_DONE)) {
            const uint32_t pixel = Q.pixel[lane];
            if (T.side) {
                TileAdd add; add.tile = tile; add.x0 = tile_x0; add.y0 = tile_y0; add.side = (int) T.side;
                SplatXYSink<TileAdd> sink; sink.film = &P.film; sink.add = add;
                st = pixel_render<Integ>(P, sc, pixel, st, sample_end, tr2, sink, &local);
            } else {
                FilmAdd add; add.accum = accum;
                SplatSink<FilmAdd> sink; sink.film = &P.film; sink.add = add;
                st = pixel_render<Integ>(P, sc, pixel, st, sample_end, tr2, sink, &local);
            }
            Q.st[lane] = st;
        }
    }
    if (!UseLog && T.side) {                                     // flush the tile: one f64 atomic per touched slot
        __syncthreads();
        const int n = (int) (T.side * T.side);
        for (int i = (int) threadIdx.x; i < n * MIW_FILM_CHANNELS; i += (int) blockDim.x) {
            const double v = tile[i];
            if (v == 0.0) continue;
            const int texel = i / MIW_FILM_CHANNELS, k = i - texel * MIW_FILM_CHANNELS;
            const int fx = tile_x0 + texel % (int) T.side, fy = tile_y0 + texel / (int) T.side;
            unsafeAtomicAdd(accum + ((size_t) fy * P.film.crop_w + fx) * MIW_FILM_CHANNELS + k, v);
        }
    }
    unsigned long long a = wave_sum(local.segments), b = wave_sum(local.samples), c = wave_sum(local.shadow_rays);
    if ((threadIdx.x & 63) == 0) {
        Counters *shard = cnt + ((blockIdx.x * (MIW_BLOCK / 64) + (threadIdx.x >> 6)) & (MIW_CNT_SHARDS - 1));
        if (a) atomicAdd(&shard->segments, a);
        if (b) atomicAdd(&shard->samples, b);
        if (c) atomicAdd(&shard->shadow_rays, c);
#if defined(MIW_SECTION_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        { unsigned long long *b_ = miw_sec_buf(); for (int i = 0; i < 15; ++i) if (b_[i]) atomicAdd(&g_sections[i], b_[i]); }
#endif
    }
}
