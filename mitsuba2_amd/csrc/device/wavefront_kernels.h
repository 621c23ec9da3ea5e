// Plan 1 (wavefront): work lists with wave-aggregated appends, k_init_lanes, k_trace<closest|any>, k_shade; the film
// adders both plans share.
// Part of the single translation unit csrc/miwave.hip (included there, in this order; not a stand-alone header).
// ---------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------

// ---- wavefront plan: stream compaction and material sorting -----------------------------
// The stage kernels do not sweep all lanes: every workgroup consumes a dense list of the lane ids
// (of its own 256-lane slice) that need the stage, and appends the lanes that need the next stage
// to that stage's list. Appending is a wavefront-level ballot + popcount prefix sum with one LDS
// atomic per wave per list — no global atomics (same-address device atomics serialise at ~12 ns),
// no memsets: a workgroup owns segment [g*256, g*256+256) of every list and publishes its counts when
// it finishes. Lists are double-buffered by iteration parity. k_trace<closest> files every traced
// lane under the BSDF type of the surface it hit, so k_shade walks the lists type by type: a
// wavefront shades one material (at most three wavefronts per workgroup straddle a boundary), idle
// wavefronts retire at once. Lane state stays lane-indexed (SoA of 16-byte fields), so list order
// never changes a result.
enum { WL_E = 0, WL_S = 1, WL_SHADE0 = 2, WL_KEYS = 4, WL_LISTS = 6 };   // shade keys: bsdf type 0..2, 3 = no surface
struct WorkLists {
    uint32_t *list[WL_LISTS];     // n_lanes entries each, segmented per workgroup
    uint32_t *count;              // [workgroup][WL_LISTS]
};

__device__ __forceinline__ void wave_append(bool pred, uint32_t *segment, uint32_t *lds_counter, uint32_t value) {
    const unsigned long long b = __ballot(pred);
    if (b == 0ull) return;                                     // wave-uniform
    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t) __ffsll((long long) b) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(lds_counter, (uint32_t) __popcll(b));
    base = (uint32_t) __shfl((int) base, (int) leader, 64);
    if (pred) segment[base + (uint32_t) __popcll(b & ((1ull << lane) - 1ull))] = value;
}

struct InitArgs {
    const uint32_t *block_ids;    // per block (row-major grid)
    const uint32_t *tile_list;    // or nullptr
    uint32_t blocks_x, blocks_y;
    uint32_t bs, bs2_log2;
    uint64_t base_seed;
};

#if !MIW_SPECTRAL   // HBM-queue plan: RGB builds only (path.h)
// -> true when the lane starts with a camera ray queued
__device__ __forceinline__ bool init_one_lane(const RenderParams &P, const LaneQueues &Q, uint32_t *pixel_out, const InitArgs &A, uint32_t lane) {
    uint32_t tile = lane >> A.bs2_log2, i = lane & ((1u << A.bs2_log2) - 1u);
    uint32_t b = A.tile_list ? A.tile_list[tile] : tile;
    uint32_t bx = b % A.blocks_x, by = b / A.blocks_x;
    uint32_t x, y;
    morton_decode2(i, x, y);                                   // integrator.cpp:200
    int32_t bw = P.film.crop_w - (int32_t) (bx * A.bs), bh = P.film.crop_h - (int32_t) (by * A.bs);
    if (bw > (int32_t) A.bs) bw = (int32_t) A.bs;
    if (bh > (int32_t) A.bs) bh = (int32_t) A.bs;
    if ((int32_t) x >= bw || (int32_t) y >= bh) {                // :201-202 — pixel outside the block
        pixel_out[lane] = 0;
        lane_init_unused(Q, lane);
        return false;
    }
    uint32_t px = (uint32_t) P.film.crop_x + bx * A.bs + x, py = (uint32_t) P.film.crop_y + by * A.bs + y;
    uint32_t pixel = px | (py << 16);
    pixel_out[lane] = pixel;
    uint64_t seed = A.base_seed + (uint64_t) A.block_ids[b] * (uint64_t) (A.bs * A.bs) + i;   // :198
    lane_init(P, Q, lane, pixel, seed);
    return P.spp > 0;
}

__global__ __launch_bounds__(MIW_BLOCK) void k_init_lanes(RenderParams P, LaneQueues Q, uint32_t *pixel_out, InitArgs A, WorkLists W) {
    __shared__ uint32_t s_cnt[WL_LISTS];
    if (threadIdx.x < WL_LISTS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    bool has_ray = false;
    if (lane < P.n_lanes) has_ray = init_one_lane(P, Q, pixel_out, A, lane);
    wave_append(has_ray, W.list[WL_E] + blockIdx.x * MIW_BLOCK, &s_cnt[WL_E], lane);
    __syncthreads();
    if (threadIdx.x < WL_LISTS) W.count[blockIdx.x * WL_LISTS + threadIdx.x] = s_cnt[threadIdx.x];
}
#endif

template <bool AnyHit>
__global__ __launch_bounds__(MIW_BLOCK) void k_trace(SceneView sc, LaneQueues Q, TraceLds cfg, WorkLists io) {
    extern __shared__ uint4 smem[];
    __shared__ uint32_t s_cnt[WL_KEYS];
    const uint32_t seg = blockIdx.x * MIW_BLOCK, *cnt = io.count + blockIdx.x * WL_LISTS;
    const uint32_t n = cnt[AnyHit ? WL_S : WL_E];
    if (AnyHit && n == 0) return;                              // nothing to test in this slice (uniform)
    // the "no surface" list already holds the lanes k_shade parked there (samples waiting for a shadow ray)
    if (!AnyHit && threadIdx.x < WL_KEYS) s_cnt[threadIdx.x] = threadIdx.x == WL_KEYS - 1 ? cnt[WL_SHADE0 + WL_KEYS - 1] : 0u;
    stage_to_lds(sc, cfg, smem);                               // ends with __syncthreads()
    const bool mine = threadIdx.x < n;
    uint32_t lane = 0, key = WL_KEYS - 1;
    if (mine) {
        lane = io.list[AnyHit ? WL_S : WL_E][seg + threadIdx.x];
        F4 d = AnyHit ? Q.sh_d[lane] : Q.ray_d[lane];
        F4 o = Q.ray_o[lane];
        Hit h;
        bool hit = trace_one<AnyHit>(sc, cfg, smem, v3(o.x, o.y, o.z), v3(d.x, d.y, d.z), o.w, d.w, h);
        if (AnyHit) {
            Q.sh_vis[lane] = hit ? 0u : 1u;
        } else {
            F4 r; r.x = h.t; r.y = h.u; r.z = h.v; r.w = u2f(h.tri);
            Q.hit[lane] = r;
            if (hit) { key = sc.bsdfs[sc.shapes[sc.tris[h.tri].shape].bsdf].type; if (key > 2u) key = 2u; }   // material sort key (conductor / plastic share the rough conductor's list)
        }
    }
    if (!AnyHit) {
#pragma unroll
        for (uint32_t k = 0; k < WL_KEYS; ++k)
            wave_append(mine && key == k, io.list[WL_SHADE0 + k] + seg, &s_cnt[k], lane);
        __syncthreads();
        if (threadIdx.x < WL_KEYS) io.count[blockIdx.x * WL_LISTS + WL_SHADE0 + threadIdx.x] = s_cnt[threadIdx.x];
    }
}

struct FilmAdd {
    double *accum;
    __device__ __forceinline__ void operator()(int texel, int k, float v) const {
        unsafeAtomicAdd(accum + (size_t) texel * MIW_FILM_CHANNELS + k, (double) v);
    }
};

// Workgroup-local film tile (resident plan, film_mode 2): the 256 lanes of a workgroup are one
// Morton-contiguous 16 x 16 pixel quad of a spiral block, so everything they splat lands in the
// (16 + 2*border)^2 texels around it. The tile is accumulated in LDS with float64 adds (ds_add_f64:
// the sum is order-free to float32 precision) and flushed to the film accumulators once per launch.
struct TileAdd {
    double *tile; int x0, y0, side;     // tile origin in crop-relative film coordinates
    __device__ __forceinline__ void operator()(int fx, int fy, int k, float v) const {
        const int tx = fx - x0, ty = fy - y0;
        if ((unsigned) tx < (unsigned) side && (unsigned) ty < (unsigned) side)
            unsafeAtomicAdd(tile + (ty * side + tx) * MIW_FILM_CHANNELS + k, (double) v);
    }
};

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

#if !MIW_SPECTRAL
template <bool UseLog>
__global__ __launch_bounds__(MIW_BLOCK) void k_shade(RenderParams P, SceneView sc, LaneQueues Q, double *accum, Counters *cnt,
                                                       uint32_t count_active, WorkLists in, WorkLists out) {
    __shared__ uint32_t s_cnt[WL_LISTS];
    if (threadIdx.x < WL_LISTS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t seg = blockIdx.x * MIW_BLOCK, *cnt_in = in.count + blockIdx.x * WL_LISTS;
    LaneCounters local; local.segments = local.samples = local.shadow_rays = 0;
    uint32_t active_lane = 0;
    // this workgroup's four material lists, back to back
    uint32_t lane = 0; bool mine = false;
    {
        uint32_t base = 0;
#pragma unroll
        for (uint32_t k = 0; k < WL_KEYS; ++k) {
            const uint32_t n = cnt_in[WL_SHADE0 + k];
            if (!mine && threadIdx.x - base < n) { lane = in.list[WL_SHADE0 + k][seg + threadIdx.x - base]; mine = true; }
            base += n;
        }
    }
    uint32_t flags = LF_DONE;
    if (mine) {
        if (UseLog && Q.log_rec) {                              // 16-byte records; the thresholds through L1 (plan 1 is not the fast path)
            LogSink16<const float *> sink{ Q.log_rec, Q.log_thr, &P.film, lane, P.spp, Q.log_rej & 255u, Q.log_rej >> 8 };
            flags = lane_shade(P, sc, Q, lane, &local, sink);
        } else if (UseLog) {
            LogSink sink; sink.log_pos = Q.log_pos; sink.log_val = Q.log_val; sink.lane = lane; sink.spp = P.spp; sink.warn_negative = P.film.warn_negative;
            flags = lane_shade(P, sc, Q, lane, &local, sink);
        } else {
            FilmAdd add; add.accum = accum;
            SplatSink<FilmAdd> sink; sink.film = &P.film; sink.add = add;
            flags = lane_shade(P, sc, Q, lane, &local, sink);
        }
        active_lane = (!(flags & LF_DONE) && count_active) ? 1u : 0u;
    }
    // next iteration's work: rays to trace, shadow rays to test, samples that only wait for a shadow ray
    const bool alive = mine && !(flags & LF_DONE);
    wave_append(alive && (flags & LF_RAY_ACTIVE), out.list[WL_E] + seg, &s_cnt[WL_E], lane);
    wave_append(alive && (flags & LF_HAS_SHADOW), out.list[WL_S] + seg, &s_cnt[WL_S], lane);
    wave_append(alive && (flags & LF_DEAD_PENDING), out.list[WL_SHADE0 + WL_KEYS - 1] + seg, &s_cnt[WL_SHADE0 + WL_KEYS - 1], lane);
    __syncthreads();
    if (threadIdx.x < WL_LISTS) out.count[blockIdx.x * WL_LISTS + threadIdx.x] = s_cnt[threadIdx.x];
    // Statistics: wave-level reduce, then one atomic per wave into one of
    // MIW_CNT_SHARDS counter records (same-address device atomics serialise at
    // ~12 ns each — 131k waves on one word would cost more than the shading).
    unsigned long long a = wave_sum(local.segments), b = wave_sum(local.samples),
                       c = wave_sum(local.shadow_rays), d = wave_sum(active_lane);
    if ((threadIdx.x & 63) == 0) {
        Counters *shard = cnt + ((blockIdx.x * (MIW_BLOCK / 64) + (threadIdx.x >> 6)) & (MIW_CNT_SHARDS - 1));
        if (a) atomicAdd(&shard->segments, a);
        if (b) atomicAdd(&shard->samples, b);
        if (c) atomicAdd(&shard->shadow_rays, c);
        if (d) atomicAdd(&shard->active_lanes, d);
    }
}
#endif   // !MIW_SPECTRAL
