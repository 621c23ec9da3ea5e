// Plan 1 (wavefront) over a tree: k_trace_stream — a PERSISTENT scene-query kernel with dynamic ray fetch — and
// k_sort_hits. Part of the single translation unit csrc/miwave.hip (not a stand-alone header).
//
// k_trace<closest|any> (wavefront_kernels.h) gives every workgroup its own 256-entry slice of the ray lists and walks
// 64 rays per wavefront in lock step: the wave lasts as long as its longest walk, and walk lengths are heavy-tailed
// (measured SIMT efficiency of the node loop 12 - 16 %, profiles/). Here the rays are a STREAM: a wavefront claims
// a list segment (one atomic per 256 rays), its lanes take rays from it, and a lane whose walk ends writes its result
// and takes the next ray while its neighbours keep walking — the classic persistent while-while traversal with
// dynamic fetch (Aila & Laine 2009), with the vote between node steps and triangle tests of phased_kernel.h.
// The kernel carries nothing but walk state (~60 VGPRs: 5 waves per SIMD — the per-lane LDS stack is what
// bounds it), so the L2 latency of the node fetches hides behind the other waves, and E (closest hit) and S (any
// hit) rays of one path iteration are served by ONE launch.
// Results are per ray and order-free (closest t, ties to the smaller primitive id; any-hit flag), so the film does not
// depend on which lane walked which ray.
#ifndef MIW_STREAM_WAVES
#define MIW_STREAM_WAVES 5          /* waves per SIMD: 5 workgroups x 32 KiB of stack = the CU's 160 KiB of LDS */
#endif
#ifndef MIW_STREAM_BATCH
#define MIW_STREAM_BATCH 16         /* lanes that must have finished / be idle before the wave stops stepping to retire / refill them */
#endif

// Wide = 0: the BVH2 walk of rounds 2 - 5 (one node = two boxes, one triangle per test). Wide = 2 (round 6): the 8-wide quantised tree and the
// two walk bodies of the resident plan's phase machine (miw/bvh8.h: walk8_node_step / walk8_tri_step — node groups, triangle pairs, the speculating
// walk), the stack one 8-byte group per level in the same 128-byte LDS column; the hit record then names triangles in the 8-wide tree's order, so
// the host hands k_sort_hits and k_shade the same view (tris / tri_vn of that order).
template <int Wide>
__global__ __launch_bounds__(MIW_BLOCK, Wide == 2 ? 4 : MIW_STREAM_WAVES) void k_trace_stream(SceneView sc, LaneQueues Q, TraceLds cfg, WorkLists io,
                                                                                uint32_t n_segments, uint32_t *next_segment) {
    extern __shared__ uint4 smem[];
    if constexpr (Wide == 2) {
        U2 *stack8 = reinterpret_cast<U2 *>(smem + cfg.stack16) + threadIdx.x;
        GlobalU4 tris_g = (GlobalU4) reinterpret_cast<uintptr_t>(sc.tris), nodes8_g = (GlobalU4) reinterpret_cast<uintptr_t>(sc.nodes8);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(nodes8_g)); asm volatile("" : "+s"(tris_g));        // (phased_kernel.h: MIW_PIN_TREE_PTRS)
#endif
        auto node8_at = [nodes8_g](uint32_t i) -> Bvh8Node {
            GlobalU4 p = nodes8_g + 5 * (size_t) i;
            miw_u4 q[5] = { p[0], p[1], p[2], p[3], p[4] };
            Bvh8Node n; __builtin_memcpy(&n, q, sizeof n); return n;
        };
        const GlobalTris tri_at_g{ tris_g };
        const PrimCtx ctx = prim_ctx(sc);
        const uint32_t me = threadIdx.x & 63u;
        auto count = [](bool p) -> int { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); };
        uint32_t seg = 0, n_e = 0, n_tot = 0, next = 0;
        bool more = true;
        bool has_ray = false, any_hit = false, found = false;
        uint32_t lane = 0;
        V3 o = v3(0.f), d = v3(0.f);
        float mint = 0.f, maxt = 0.f, tmax = 0.f;
        FastRay r; r.inv_d = r.neg_o_inv_d = v3(0.f); r.mint = 0.f;
        Walk8 w8; w8.gb = w8.gm = w8.tb = w8.tm = w8.tb2 = w8.tm2 = 0u;
        Hit best; best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
        for (;;) {
            // ---- retire finished walks ----
            const bool e_end = has_ray && walk8_over(w8);
            int n_end = count(e_end);
            const int n_live = count(has_ray) - n_end;
            if (n_end > 0 && (n_end >= MIW_STREAM_BATCH || n_live == 0 || !more)) {
                if (e_end) {
                    if (any_hit) Q.sh_vis[lane] = found ? 0u : 1u;
                    else { F4 h; h.x = best.t; h.y = best.u; h.z = best.v; h.w = u2f(best.tri); Q.hit[lane] = h; }
                    has_ray = false;
                }
                n_end = 0;
            }
            // ---- refill idle lanes from the stream ----
            int n_idle = count(!has_ray);
            if (more && (n_idle >= MIW_STREAM_BATCH || n_idle + n_end == 64)) {
                while (n_idle > 0) {
                    if (next == n_tot) {
                        uint32_t sg = 0;
                        if (me == 0) sg = atomicAdd(next_segment, 1u);
                        sg = (uint32_t) __builtin_amdgcn_readfirstlane((int) sg);
                        if (sg >= n_segments) { more = false; break; }
                        seg = sg;
                        n_e = io.count[sg * WL_LISTS + WL_E];
                        n_tot = n_e + io.count[sg * WL_LISTS + WL_S];
                        next = 0;
                        continue;
                    }
                    const unsigned long long idle = __builtin_amdgcn_ballot_w64(!has_ray);
                    const uint32_t rank = (uint32_t) __builtin_popcountll(idle & ((1ull << me) - 1ull));
                    const uint32_t avail = n_tot - next, take = avail < (uint32_t) n_idle ? avail : (uint32_t) n_idle;
                    if (!has_ray && rank < take) {
                        const uint32_t e = next + rank;
                        any_hit = e >= n_e;
                        lane = any_hit ? io.list[WL_S][seg * MIW_BLOCK + (e - n_e)] : io.list[WL_E][seg * MIW_BLOCK + e];
                        const F4 ro = Q.ray_o[lane], rd = any_hit ? Q.sh_d[lane] : Q.ray_d[lane];
                        o = v3(ro.x, ro.y, ro.z); mint = ro.w; d = v3(rd.x, rd.y, rd.z); maxt = rd.w;
                        r = fast_ray(o, d, mint); tmax = maxt;
                        walk8_begin(w8, r); found = false;
                        best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
                        has_ray = true;
                    }
                    next += take; n_idle -= (int) take;
                }
            }
            if (count(has_ray) == 0) break;
            // ---- walk: the body with more ready lanes; a loop hands over once it is outnumbered 2 : 1 or enough walks have ended to retire ----
            bool e_node = has_ray && walk8_node_ready<true>(w8), e_leaf = has_ray && walk8_tri_ready(w8);
            int n_node = count(e_node), n_leaf = count(e_leaf);
            if (n_node >= n_leaf && n_node > 0) {
                do {
                    if (e_node) {
                        const auto &nd = node8_at(walk8_next_node(w8));
                        walk8_node_step<true>(nd, r, widen(tmax), w8, LdsColumn8{ stack8 });
                    }
                    e_node = has_ray && walk8_node_ready<true>(w8);
                    const int now = count(e_node);
                    n_leaf = count(has_ray && walk8_tri_ready(w8));
                    if (now == 0 || 2 * now < n_leaf || count(has_ray && walk8_over(w8)) >= MIW_STREAM_BATCH) break;
                } while (true);
            } else if (n_leaf > 0) {
                do {
                    if (e_leaf) walk8_tri_step<true, true>(tri_at_g, ctx, o, d, mint, maxt, any_hit, best, tmax, found, w8);
                    e_leaf = has_ray && walk8_tri_ready(w8);
                    const int now = count(e_leaf);
                    n_node = count(has_ray && walk8_node_ready<true>(w8));
                    if (now == 0 || 2 * now <= n_node || count(has_ray && walk8_over(w8)) >= MIW_STREAM_BATCH) break;
                } while (true);
            }
        }
        return;
    }
    int32_t *stack = reinterpret_cast<int32_t *>(smem + cfg.stack16) + threadIdx.x;
    const BvhNode *gnodes = sc.nodes;
    const Tri *gtris = sc.tris;
    const PrimCtx ctx = prim_ctx(sc);
    const uint32_t me = threadIdx.x & 63u;
    auto count = [](bool p) -> int { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); };

    // the wave's current list segment (wave-uniform): entries [next, n_e) are E rays, [n_e, n_tot) S rays
    uint32_t seg = 0, n_e = 0, n_tot = 0, next = 0;
    bool more = true;                                          // segments left to claim

    // the ray a lane walks
    bool has_ray = false, any_hit = false, found = false;
    uint32_t lane = 0;
    V3 o = v3(0.f), d = v3(0.f);
    float mint = 0.f, maxt = 0.f, tmax = 0.f;
    FastRay r; r.inv_d = r.neg_o_inv_d = v3(0.f); r.mint = 0.f;
    int32_t cur = MIW_WALK_DONE, sp = 0;
    uint32_t tri_i = 0, tri_end = 0;
    Hit best; best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;

    for (;;) {
        // ---- retire finished walks ----
        bool e_end = has_ray && tri_i >= tri_end && cur == MIW_WALK_DONE;
        int n_end = count(e_end);
        const int n_live = count(has_ray) - n_end;
        if (n_end > 0 && (n_end >= MIW_STREAM_BATCH || n_live == 0 || !more)) {
            if (e_end) {
                if (any_hit) Q.sh_vis[lane] = found ? 0u : 1u;
                else { F4 h; h.x = best.t; h.y = best.u; h.z = best.v; h.w = u2f(best.tri); Q.hit[lane] = h; }
                has_ray = false;
            }
            n_end = 0;
        }
        // ---- refill idle lanes from the stream ----
        int n_idle = count(!has_ray);
        if (more && (n_idle >= MIW_STREAM_BATCH || n_idle + n_end == 64)) {
            while (n_idle > 0) {
                if (next == n_tot) {                               // claim the next segment (one atomic per wave)
                    uint32_t s = 0;
                    if (me == 0) s = atomicAdd(next_segment, 1u);
                    s = (uint32_t) __builtin_amdgcn_readfirstlane((int) s);
                    if (s >= n_segments) { more = false; break; }
                    seg = s;
                    n_e = io.count[s * WL_LISTS + WL_E];
                    n_tot = n_e + io.count[s * WL_LISTS + WL_S];
                    next = 0;
                    continue;
                }
                const unsigned long long idle = __builtin_amdgcn_ballot_w64(!has_ray);
                const uint32_t rank = (uint32_t) __builtin_popcountll(idle & ((1ull << me) - 1ull));
                const uint32_t avail = n_tot - next, take = avail < (uint32_t) n_idle ? avail : (uint32_t) n_idle;
                if (!has_ray && rank < take) {
                    const uint32_t e = next + rank;
                    any_hit = e >= n_e;
                    lane = any_hit ? io.list[WL_S][seg * MIW_BLOCK + (e - n_e)] : io.list[WL_E][seg * MIW_BLOCK + e];
                    const F4 ro = Q.ray_o[lane], rd = any_hit ? Q.sh_d[lane] : Q.ray_d[lane];
                    o = v3(ro.x, ro.y, ro.z); mint = ro.w; d = v3(rd.x, rd.y, rd.z); maxt = rd.w;
                    r = fast_ray(o, d, mint); tmax = maxt;
                    cur = 0; sp = 0; tri_i = tri_end = 0; found = false;
                    best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
                    has_ray = true;
                }
                next += take; n_idle -= (int) take;
            }
        }
        if (count(has_ray) == 0) break;                        // nothing left to claim, every walk retired

        // ---- walk: node steps while the node lanes are the larger group, else triangle tests ----
        bool has_range = tri_i < tri_end;
        bool e_node = has_ray && cur >= 0 && !has_range, e_leaf = has_ray && has_range;
        int n_node = count(e_node), n_leaf = count(e_leaf);
        if (n_node >= n_leaf && n_node > 0) {
            do {
                if (e_node) {
                    const BvhNode &n = gnodes[cur];
                    float tn0, tn1;
                    const float wide = widen(tmax);
                    const bool h0 = box_test_fast(n.lo0, n.hi0, r, wide, tn0), h1 = box_test_fast(n.lo1, n.hi1, r, wide, tn1);
                    const int32_t c0 = n.child0, c1 = n.child1;
                    const bool second_first = tn1 < tn0;
                    int32_t nxt = h0 ? c0 : c1;
                    if (h0 && h1) {
                        stack[sp * MIW_BLOCK] = second_first ? c0 : c1; ++sp;
                        nxt = second_first ? c1 : c0;
                    } else if (!(h0 || h1)) {
                        nxt = MIW_WALK_DONE;
                        if (sp != 0) { --sp; nxt = stack[sp * MIW_BLOCK]; }
                    }
                    if (nxt < 0 && nxt != MIW_WALK_DONE) {          // a leaf: its triangles become the lane's range
                        const uint32_t code = (uint32_t) ~nxt;
                        tri_i = code >> 4; tri_end = tri_i + (code & 15u) + 1u;
                        nxt = MIW_WALK_DONE;
                        if (sp != 0) { --sp; nxt = stack[sp * MIW_BLOCK]; }
                    }
                    cur = nxt;
                }
                has_range = tri_i < tri_end;
                e_node = has_ray && cur >= 0 && !has_range;
                const int now = count(e_node);
                n_leaf = count(has_ray && has_range);
                if (now == 0 || now < n_leaf || count(has_ray && !has_range && cur == MIW_WALK_DONE) >= MIW_STREAM_BATCH) break;
            } while (true);
        } else if (n_leaf > 0) {
            do {
                if (e_leaf) {
                    const Tri &tr = gtris[tri_i];
                    float t, u, v;
                    if (prim_intersect<true>(tr, ctx, o, d, mint, maxt, t, u, v)) {
                        if (any_hit) { found = true; tri_end = 0; cur = MIW_WALK_DONE; sp = 0; }
                        else if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                            best.t = t; best.u = u; best.v = v; best.tri = tri_i; best.prim = tr.prim;
                            tmax = t;
                        }
                    }
                    ++tri_i;
                    if (tri_i >= tri_end && cur < 0 && cur != MIW_WALK_DONE) {   // the stack handed over another leaf
                        const uint32_t code = (uint32_t) ~cur;
                        tri_i = code >> 4; tri_end = tri_i + (code & 15u) + 1u;
                        cur = MIW_WALK_DONE;
                        if (sp != 0) { --sp; cur = stack[sp * MIW_BLOCK]; }
                    }
                }
                has_range = tri_i < tri_end;
                e_leaf = has_ray && has_range;
                const int now = count(e_leaf);
                n_node = count(has_ray && cur >= 0 && !has_range);
                if (now == 0 || now <= n_node || count(has_ray && !has_range && cur == MIW_WALK_DONE) >= MIW_STREAM_BATCH) break;
            } while (true);
        }
    }
}

// Files every lane whose extension ray was just traced under the BSDF type of the surface it hit (the material sort of
// k_trace<closest>, wavefront_kernels.h), workgroup by workgroup over its own list slice.
__global__ __launch_bounds__(MIW_BLOCK) void k_sort_hits(SceneView sc, LaneQueues Q, WorkLists io, uint32_t *next_segment) {
    __shared__ uint32_t s_cnt[WL_KEYS];
    const uint32_t seg = blockIdx.x * MIW_BLOCK, *cnt = io.count + blockIdx.x * WL_LISTS;
    const uint32_t n = cnt[WL_E];
    // the "no surface" list already holds the lanes k_shade parked there (samples waiting for a shadow ray)
    if (threadIdx.x < WL_KEYS) s_cnt[threadIdx.x] = threadIdx.x == WL_KEYS - 1 ? cnt[WL_SHADE0 + WL_KEYS - 1] : 0u;
    __syncthreads();
    const bool mine = threadIdx.x < n;
    uint32_t lane = 0, key = WL_KEYS - 1;
    if (mine) {
        lane = io.list[WL_E][seg + threadIdx.x];
        const uint32_t tri = f2u(Q.hit[lane].w);
        if (tri != MIW_MISS) { key = sc.bsdfs[sc.shapes[sc.tris[tri].shape].bsdf].type; if (key > 2u) key = 2u; }
    }
#pragma unroll
    for (uint32_t k = 0; k < WL_KEYS; ++k)
        wave_append(mine && key == k, io.list[WL_SHADE0 + k] + seg, &s_cnt[k], lane);
    __syncthreads();
    if (threadIdx.x < WL_KEYS) io.count[blockIdx.x * WL_LISTS + WL_SHADE0 + threadIdx.x] = s_cnt[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_segment = 0;      // the stream of the next iteration starts at segment 0
}
