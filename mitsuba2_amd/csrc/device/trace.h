// Scene queries of the device kernels: what a workgroup stages in LDS (BVH top, triangle packets, leaf boxes), the
// fast slab test, the LDS-stack BVH walk and the paired extension + shadow query of the resident plan (trace2).
// Part of the single translation unit csrc/miwave.hip (included there, in this order; not a stand-alone header).
// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------

struct TraceLds {           // what a trace workgroup finds in its dynamic LDS
    uint32_t nodes_staged;  // first `nodes_staged` BVH nodes (breadth-first = top of tree)
    uint32_t tris_staged;   // first `tris_staged` triangles (all of them or none)
    uint32_t brute;         // 1: tiny scene — LDS holds edge-form triangle packets (+ leaf boxes), no BVH walk
    uint32_t leaves;        // brute: number of LeafBox records staged behind the packets
    uint32_t stack;         // 1: tree walk with a per-lane LDS stack (MIW_STACK_ENTRIES x 256 dwords) at stack16
    uint32_t stack16;       // uint4 offset of the stack area in dynamic LDS
    uint32_t shade_num, shade_den;   // k_path_phased's shade vote: shade once n_shade * shade_num >= shade_den * (lanes of the busier walk body)
    uint32_t node_exit, tri_exit;    // k_path_phased: a walk loop hands over once the lanes still in it are outnumbered 2 : 1 by the other walk body's, or
                            // node_exit : 1 / tri_exit : 1 by the lanes waiting for another body (phased_kernel.h)
    uint32_t queues;        // render kernels fed from the shared pixel queue: 1 queue, or 8 (one per XCD; resident_kernel.h: QueueWork)
    uint32_t tail_prio;     // 1: least-progress-first wave priorities (QueueWork::tick) — shards of about one pixel per resident lane
    uint32_t thr16;         // render kernels that log 16-byte records: uint4 offset of the 256 phase thresholds (film.h) in dynamic LDS
    uint32_t tab16;         // packet kernels and the phase machine: uint4 offset of the scene's small tables in dynamic LDS (stage_tables)
    uint32_t tab_words[10]; // dwords of: shapes, bsdfs, emitters, emit_tri, emit_vnorm, emit_pmf, emit_cdf; packet kernels also: tris, tri_vn, tri_uv (0: absent)
    uint32_t env_top_count, env_top_base, env_top_words;   // the environment warp's top levels behind the tables (envmap.h: EnvTop); 0: none
    uint32_t pool_claim_min; // k_path_pooled: a walk loop looks for new jobs only when this many lanes of a slot are empty
    uint32_t pool16, stat16; // k_path_pooled (pooled_kernel.h): uint4 offsets of the workgroup's walk-job records (5 x 16 B per lane) and of their status bytes
};

// Padded bounding box of one BVH leaf (consecutive triangles in leaf order) + their 64-bit candidate mask, 32 B = 2 x b128.
// The two planes of an axis sit next to each other — lo.x hi.x lo.y hi.y | lo.z hi.z mask — so that one packed fma
// (v_pk_fma_f32: two IEEE fmas per lane in one issue slot) turns a register pair straight out of the ds_read_b128 into the
// two slab distances of that axis (leaf_box_test below).
struct alignas(16) LeafBox { float p[6]; uint32_t mask_lo, mask_hi; };   // p = lo.x hi.x lo.y hi.y lo.z hi.z; mask: one bit per triangle of the leaf (leaf order)
static_assert(sizeof(LeafBox) == 32, "LeafBox must be 32 bytes");

// Triangle packet for the brute-force sweep: p0, e1, e2, prim (48 B = 3 x b128).
struct alignas(16) TriPacket { float p0[3], e1[3], e2[3]; uint32_t prim; uint32_t pad[2]; };
static_assert(sizeof(TriPacket) == 48, "TriPacket must be 48 bytes");

#ifndef MIW_LDS_TABLES
#define MIW_LDS_TABLES 1            /* 1: packet kernels and the phase machine read the scene's small tables from LDS (stage_tables) */
#endif
#ifndef MIW_OCTANT_BOXES
#define MIW_OCTANT_BOXES 1          /* 1: the leaf boxes of a tiny scene are staged once per ray octant, entry / exit plane of every axis side by side (leaf_box_test_octant); 0: one copy, min / max per axis */
#endif
#define MIW_LEAF_BOX_COPIES (MIW_OCTANT_BOXES ? 8u : 1u)
__device__ __forceinline__ void stage_to_lds(const SceneView &sc, TraceLds cfg, uint4 *smem) {
    if (cfg.brute) {
        TriPacket *dst = reinterpret_cast<TriPacket *>(smem);
        for (uint32_t i = threadIdx.x; i < sc.tri_count; i += blockDim.x) {
            const Tri &t = sc.tris[i];
            V3 p0 = ld3(t.p0), e1 = ld3(t.p1) - p0, e2 = ld3(t.p2) - p0;
            TriPacket k;
            k.p0[0] = p0.x; k.p0[1] = p0.y; k.p0[2] = p0.z; k.e1[0] = e1.x; k.e1[1] = e1.y; k.e1[2] = e1.z;
            k.e2[0] = e2.x; k.e2[1] = e2.y; k.e2[2] = e2.z; k.prim = t.prim; k.pad[0] = k.pad[1] = 0;
            dst[i] = k;
        }
        // leaf boxes of the SAH tree behind the packets (k_path_resident's candidate filter)
        uint4 *dst_b = smem + sc.tri_count * (sizeof(TriPacket) / 16);
        const uint4 *src_b = reinterpret_cast<const uint4 *>(sc.leaf_boxes);
#if MIW_OCTANT_BOXES
        // eight copies per leaf, one per ray octant o = (d.x < 0) | (d.y < 0) << 1 | (d.z < 0) << 2, copy o at slot 8 * leaf + o: per
        // axis the plane the ray ENTERS through first, then the one it leaves through. A lane reads the copy of its ray's octant —
        // eight different 32-byte slots of one 256-byte row: every LDS bank once, no conflict.
        for (uint32_t i = threadIdx.x; i < cfg.leaves * 8u; i += blockDim.x) {
            const LeafBox b = reinterpret_cast<const LeafBox *>(sc.leaf_boxes)[i >> 3];
            const uint32_t o = i & 7u;
            LeafBox k;
            for (int a = 0; a < 3; ++a) { const bool neg = (o >> a) & 1u; k.p[2 * a] = b.p[2 * a + (neg ? 1 : 0)]; k.p[2 * a + 1] = b.p[2 * a + (neg ? 0 : 1)]; }
            k.mask_lo = b.mask_lo; k.mask_hi = b.mask_hi;
            reinterpret_cast<LeafBox *>(dst_b)[i] = k;
        }
        (void) src_b;
#else
        for (uint32_t i = threadIdx.x; i < cfg.leaves * (sizeof(LeafBox) / 16); i += blockDim.x) dst_b[i] = src_b[i];
#endif
        // per-packet vertex bounds grown by accept_pad (shape.h: the accept rule), 24 B each, behind the boxes
        float *dst_t = reinterpret_cast<float *>(dst_b + cfg.leaves * MIW_LEAF_BOX_COPIES * (sizeof(LeafBox) / 16));
        const float *src_t = reinterpret_cast<const float *>(sc.tri_bounds);
        for (uint32_t i = threadIdx.x; i < sc.tri_count * 6u; i += blockDim.x) dst_t[i] = src_t[i];
        __syncthreads();
        return;
    }
    const uint4 *src_n = reinterpret_cast<const uint4 *>(sc.nodes);
    uint32_t n16 = cfg.nodes_staged * (sizeof(BvhNode) / 16);
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) smem[i] = src_n[i];
    const uint4 *src_t = reinterpret_cast<const uint4 *>(sc.tris);
    uint32_t t16 = cfg.tris_staged * (sizeof(Tri) / 16);
    uint4 *dst_t = smem + n16;
    for (uint32_t i = threadIdx.x; i < t16; i += blockDim.x) dst_t[i] = src_t[i];
    __syncthreads();
}

// The phase thresholds + bin table of the 16-byte sample record (film.h) behind everything else in dynamic LDS: the class search
// of a finished sample is two dependent rounds of reads per axis, which belong in LDS (64-cycle round trips), not in L1.
__device__ __forceinline__ const float *stage_thresholds(uint4 *smem, TraceLds cfg, const float *thr_global) {
    float *t = reinterpret_cast<float *>(smem + cfg.thr16);
    if (thr_global) {
        for (uint32_t i = threadIdx.x; i < MIW_FC_TABLE; i += blockDim.x) t[i] = thr_global[i];
        __syncthreads();
    }
    return t;
}

// The scene's small tables — shape, BSDF and emitter records, the emitters' face tables — copied behind everything else in dynamic
// LDS, and the kernel's SceneView pointed at the copies (round 4). What path_step reads between two scene queries is a handful
// of DEPENDENT lookups (triangle -> shape -> BSDF record; emitter record -> CDF -> face); from global memory each is a round trip
// through the CU's request path, which the tree walks keep full (a shade run of the phase machine: ~30 k cycles on the material
// balls, ~82 k on the interior for ~2 k instruction slots). From LDS they are ds_reads. Unconditional for the
// kernels that call it (the pointers must not be a choice between address spaces): the host launches those kernels only when
// the tables fit (mi_render / mi_bvh_build), the lock-step tree kernels read the tables from global memory as before.
template <bool WithTris>     // packet scenes (<= 64 triangles): the triangle records the shading reads, their vertex normals and texture coordinates too
__device__ __forceinline__ void stage_tables(SceneView &sc, const TraceLds &cfg, uint4 *smem) {
    constexpr int N = WithTris ? 10 : 7;
    uint32_t *dst = reinterpret_cast<uint32_t *>(smem + cfg.tab16);
    const uint32_t *src[10] = { reinterpret_cast<const uint32_t *>(sc.shapes), reinterpret_cast<const uint32_t *>(sc.bsdfs), reinterpret_cast<const uint32_t *>(sc.emitters),
                                reinterpret_cast<const uint32_t *>(sc.emit_tri), reinterpret_cast<const uint32_t *>(sc.emit_vnorm),
                                reinterpret_cast<const uint32_t *>(sc.emit_pmf), reinterpret_cast<const uint32_t *>(sc.emit_cdf),
                                reinterpret_cast<const uint32_t *>(sc.tris), reinterpret_cast<const uint32_t *>(sc.tri_vn), reinterpret_cast<const uint32_t *>(sc.tri_uv) };
    uint32_t *at[10];
    uint32_t off = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const uint32_t n = cfg.tab_words[k];
        at[k] = dst + off;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[off + i] = src[k][i];
        off += (n + 3u) & ~3u;                                  // every table starts on a 16-byte boundary
    }
    // the smallest levels of the environment map's sampling hierarchy (the first steps of hier2d_sample's chain of dependent reads)
    uint32_t *env_at = dst + off;
    if (cfg.env_top_words) {
        const uint32_t *lv = reinterpret_cast<const uint32_t *>(sc.env->levels) + cfg.env_top_base;
        for (uint32_t i = threadIdx.x; i < cfg.env_top_words; i += blockDim.x) env_at[i] = lv[i];
    }
    sc.env_top = reinterpret_cast<const float *>(env_at); sc.env_top_count = cfg.env_top_count; sc.env_top_base = cfg.env_top_base;
    __syncthreads();
    sc.shapes = reinterpret_cast<const ShapeRec *>(at[0]); sc.bsdfs = reinterpret_cast<const BsdfRec *>(at[1]); sc.emitters = reinterpret_cast<const EmitterRec *>(at[2]);
    sc.emit_tri = reinterpret_cast<const float *>(at[3]);
    sc.emit_vnorm = reinterpret_cast<const float *>(at[4]);          // (read only for emitters whose flags say so; 0 words when absent)
    sc.emit_pmf = reinterpret_cast<const float *>(at[5]); sc.emit_cdf = reinterpret_cast<const float *>(at[6]);
    if (WithTris) {                                                  // (tri_vn / tri_uv: read only for shapes whose flags say so)
        sc.tris = reinterpret_cast<const Tri *>(at[7]); sc.tri_vn = reinterpret_cast<const float *>(at[8]); sc.tri_uv = reinterpret_cast<const float *>(at[9]);
    }
}

// Candidate-box test shared by the tiny-scene filter (trace2) and the stack traversal below: the slab
// test of bvh.h re-expressed for speed — v_rcp_f32 for 1/d, t = fma(plane, inv_d, -o*inv_d), hardware
// min/max (v_min3/v_max3). It only has to stay CONSERVATIVE, not bit-reproducible: boxes are padded by
// 1e-5 x the scene extent (bvh_build.h), orders of magnitude above the rounding differences between the
// two forms, and every hit is decided by the exact Moeller-Trumbore test.
struct FastRay { V3 inv_d, neg_o_inv_d; float mint; };
__device__ __forceinline__ FastRay fast_ray(V3 o, V3 d, float mint) {
    FastRay r;
    float dx = abs_(d.x) < 1e-30f ? mulsign(1e-30f, d.x) : d.x,
          dy = abs_(d.y) < 1e-30f ? mulsign(1e-30f, d.y) : d.y,
          dz = abs_(d.z) < 1e-30f ? mulsign(1e-30f, d.z) : d.z;
    r.inv_d = v3(__builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));
    r.neg_o_inv_d = v3(-(o.x * r.inv_d.x), -(o.y * r.inv_d.y), -(o.z * r.inv_d.z));
    r.mint = mint;
    return r;
}
__device__ __forceinline__ bool box_test_fast(const float *lo, const float *hi, const FastRay &r, float tmax_wide, float &tn_out) {
    float t0x = __builtin_fmaf(lo[0], r.inv_d.x, r.neg_o_inv_d.x), t1x = __builtin_fmaf(hi[0], r.inv_d.x, r.neg_o_inv_d.x),
          t0y = __builtin_fmaf(lo[1], r.inv_d.y, r.neg_o_inv_d.y), t1y = __builtin_fmaf(hi[1], r.inv_d.y, r.neg_o_inv_d.y),
          t0z = __builtin_fmaf(lo[2], r.inv_d.z, r.neg_o_inv_d.z), t1z = __builtin_fmaf(hi[2], r.inv_d.z, r.neg_o_inv_d.z);
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                               __builtin_fmaxf(__builtin_fminf(t0z, t1z), r.mint));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
    // widened like bvh.h's box_test, plus slack for the fma-form rounding
    tf = __builtin_fmaf(abs_(tf), 2e-6f, tf);
    tn_out = tn;
    return tn <= tf && tn <= tmax_wide;
}
// The same test on a LeafBox, three packed fmas per ray instead of six scalar ones (same IEEE fma per plane: same distances).
typedef float miw_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool leaf_box_test(const LeafBox &b, const FastRay &r, float tmax_wide) {
    const miw_f2 px = { b.p[0], b.p[1] }, py = { b.p[2], b.p[3] }, pz = { b.p[4], b.p[5] };
    const miw_f2 tx = __builtin_elementwise_fma(px, (miw_f2) { r.inv_d.x, r.inv_d.x }, (miw_f2) { r.neg_o_inv_d.x, r.neg_o_inv_d.x }),
                 ty = __builtin_elementwise_fma(py, (miw_f2) { r.inv_d.y, r.inv_d.y }, (miw_f2) { r.neg_o_inv_d.y, r.neg_o_inv_d.y }),
                 tz = __builtin_elementwise_fma(pz, (miw_f2) { r.inv_d.z, r.inv_d.z }, (miw_f2) { r.neg_o_inv_d.z, r.neg_o_inv_d.z });
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)),
                                     __builtin_fmaxf(__builtin_fminf(tz.x, tz.y), r.mint));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fmaxf(tz.x, tz.y));
    tf = __builtin_fmaf(abs_(tf), 2e-6f, tf);
    return tn <= tf && tn <= tmax_wide;
}
// The same test on the copy of a LeafBox staged for the ray's octant (stage_to_lds): p = entry.x exit.x entry.y exit.y entry.z exit.z,
// so the packed fma of an axis yields (entry distance, exit distance) and the six min / max of the test above fall away. t(plane) =
// fma(plane, inv_d, -o * inv_d) is monotone in the plane with the sign of inv_d, so these ARE the minima / maxima of the other form.
__device__ __forceinline__ bool leaf_box_test_octant(const LeafBox &b, const FastRay &r, float tmax_wide) {
    const miw_f2 px = { b.p[0], b.p[1] }, py = { b.p[2], b.p[3] }, pz = { b.p[4], b.p[5] };
    const miw_f2 tx = __builtin_elementwise_fma(px, (miw_f2) { r.inv_d.x, r.inv_d.x }, (miw_f2) { r.neg_o_inv_d.x, r.neg_o_inv_d.x }),
                 ty = __builtin_elementwise_fma(py, (miw_f2) { r.inv_d.y, r.inv_d.y }, (miw_f2) { r.neg_o_inv_d.y, r.neg_o_inv_d.y }),
                 tz = __builtin_elementwise_fma(pz, (miw_f2) { r.inv_d.z, r.inv_d.z }, (miw_f2) { r.neg_o_inv_d.z, r.neg_o_inv_d.z });
    const float tn = __builtin_fmaxf(__builtin_fmaxf(tx.x, ty.x), __builtin_fmaxf(tz.x, r.mint));
    float tf = __builtin_fminf(__builtin_fminf(tx.y, ty.y), tz.y);
    tf = __builtin_fmaf(abs_(tf), 2e-6f, tf);
    return tn <= tf && tn <= tmax_wide;
}
__device__ __forceinline__ uint32_t ray_octant(const FastRay &r) {
    return (r.inv_d.x < 0.f ? 1u : 0u) | (r.inv_d.y < 0.f ? 2u : 0u) | (r.inv_d.z < 0.f ? 4u : 0u);
}
__device__ __forceinline__ float widen(float t) { return __builtin_fmaf(abs_(t), 2e-6f, t); }

// Stack traversal of the BVH2 for scenes that do not fit LDS: the per-lane stack lives in LDS
// (entry-major, one dword per lane per entry: conflict-free), the top of the tree is read from LDS
// and the rest through L1/L2. Same observable result as bvh_intersect (== brute force, ties to the
// smaller primitive id); used when the tree depth fits MIW_STACK_ENTRIES, otherwise the stackless
// trail walk of bvh.h runs.
#ifndef MIW_STACK_ENTRIES
#define MIW_STACK_ENTRIES 32
#endif
#ifndef MIW_WALK
#define MIW_WALK 1                /* 0: one loop, node or leaf per iteration; 1: while-while; 2: while-while + one postponed leaf per lane */
#endif
#ifndef MIW_LDS_TOP
#define MIW_LDS_TOP 0             /* 1: the first 255 nodes of a tree that does not fit LDS are staged there — a flat-address select per node visit; measured 4 - 12 % slower than plain global loads (the top of the tree lives in L1 / L2 anyway) */
#endif
#ifndef MIW_TREE_WAVES
#define MIW_TREE_WAVES 3          /* waves per SIMD the tree-walk kernel is compiled for; 4 (<= 128 VGPRs) spills 23 registers and measured 10-20 % slower on C3 / C4 */
#endif
template <bool AnyHit, bool Analytic, typename NodeAt, typename TriAt>
__device__ __forceinline__ bool bvh_intersect_stack(NodeAt node_at, TriAt tri_at, int32_t *stack /* + threadIdx.x */,
                                                    V3 o, V3 d, float mint, float maxt, Hit &best, PrimCtx ctx) {
    best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    const FastRay r = fast_ray(o, d, mint);
    float tmax = maxt;
    int32_t cur = 0, sp = 0;
    for (;;) {
        if (cur >= 0) {
            const BvhNode &n = node_at(cur);
            float tn0, tn1;
            const float wide = widen(tmax);
            const bool h0 = box_test_fast(n.lo0, n.hi0, r, wide, tn0), h1 = box_test_fast(n.lo1, n.hi1, r, wide, tn1);
            const int32_t c0 = n.child0, c1 = n.child1;
            if (h0 && h1) {
                const bool second_first = tn1 < tn0;
                stack[sp * MIW_BLOCK] = second_first ? c0 : c1; ++sp;
                cur = second_first ? c1 : c0;
                continue;
            }
            if (h0 || h1) { cur = h0 ? c0 : c1; continue; }
        } else {
            const uint32_t code = (uint32_t) ~cur, first = code >> 4, count = (code & 15u) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const Tri &tr = tri_at(first + i);
                float t, u, v;
                if (prim_intersect<Analytic>(tr, ctx, o, d, mint, maxt, t, u, v)) {
                    if (AnyHit) { best.t = 0.f; best.tri = first + i; best.prim = tr.prim; return true; }
                    if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                        best.t = t; best.u = u; best.v = v; best.tri = first + i; best.prim = tr.prim;
                        tmax = t;
                    }
                }
            }
        }
        if (sp == 0) return best.tri != MIW_MISS;
        --sp; cur = stack[sp * MIW_BLOCK];
    }
}

// The same traversal as a WHILE-WHILE loop (MIW_WALK >= 1, the default): a wavefront's lanes sit at different places of
// their walks, and in the loop above every wave-iteration in which ANY lane holds a leaf pays the whole leaf body
// (up to 4 Moeller-Trumbore tests) next to the node body, although only ~1 lane in 8 is at a leaf. Here each lane
// first descends through inner nodes until it holds a leaf (the wave leaves the node loop when every lane does),
// then all lanes test their leaf's triangles together. MIW_WALK == 2 postpones ONE leaf per lane: a lane that
// reaches its first leaf parks it and keeps descending until it reaches a second one, so that it does not idle
// while its neighbours still descend (speculative traversal; tmax is then one leaf late for the nodes in between,
// which only costs box tests — the result is the same: every triangle Moeller-Trumbore accepts is tested).
// Debug builds (-DMIW_WALK_STATS=1): lane-steps and wave-steps of the node and the triangle loop, per ray kind,
// summed into g_walk_stats (printed by mi_render under MIW_DEBUG): lane-steps / (64 x wave-steps) = SIMT efficiency.
#if defined(MIW_WALK_STATS)
#define MIW_WALK_STATS_DECL unsigned long long ws_lane_[2] = { 0, 0 }; float ws_wave_[2] = { 0.f, 0.f }
#define MIW_WALK_STATS_STEP(k) do { ws_lane_[k]++; ws_wave_[k] += 1.f / (float) __popcll(__ballot(1)); } while (0)
#define MIW_WALK_STATS_FLUSH(base) do { atomicAdd(&g_walk_stats[(base)], ws_lane_[0]); atomicAdd(&g_walk_stats[(base) + 1], ws_lane_[1]); \
        atomicAdd(&g_walk_statsf[(base)], ws_wave_[0]); atomicAdd(&g_walk_statsf[(base) + 1], ws_wave_[1]); \
        atomicAdd(&g_walk_stats[(base) + 2], 1ull); } while (0)
#else
#define MIW_WALK_STATS_DECL do { } while (0)
#define MIW_WALK_STATS_STEP(k) do { } while (0)
#define MIW_WALK_STATS_FLUSH(base) do { } while (0)
#endif
#define MIW_WALK_DONE ((int32_t) 0x80000000)      /* not a leaf code: ~DONE >> 4 is past every triangle index */
template <bool AnyHit, bool Analytic, int Postpone, typename NodeAt, typename TriAt>
__device__ __forceinline__ bool bvh_intersect_ww(NodeAt node_at, TriAt tri_at, int32_t *stack /* + threadIdx.x */,
                                                 V3 o, V3 d, float mint, float maxt, Hit &best, PrimCtx ctx) {
    best.t = MIW_INFINITY; best.u = best.v = 0.f; best.tri = MIW_MISS; best.prim = 0xffffffffu;
    const FastRay r = fast_ray(o, d, mint);
    float tmax = maxt;
    int32_t cur = 0, sp = 0, parked = MIW_WALK_DONE;
    MIW_WALK_STATS_DECL;
    for (;;) {
        if (Postpone && cur < 0 && cur != MIW_WALK_DONE && parked == MIW_WALK_DONE) {
            parked = cur;
            if (sp == 0) cur = MIW_WALK_DONE; else { --sp; cur = stack[sp * MIW_BLOCK]; }
        }
        while (cur >= 0) {                                       // ---- node phase: descend to the next leaf
            MIW_WALK_STATS_STEP(0);
            const BvhNode &n = node_at(cur);
            float tn0, tn1;
            const float wide = widen(tmax);
            const bool h0 = box_test_fast(n.lo0, n.hi0, r, wide, tn0), h1 = box_test_fast(n.lo1, n.hi1, r, wide, tn1);
            const int32_t c0 = n.child0, c1 = n.child1;
            const bool second_first = tn1 < tn0;
            int32_t next = h0 ? c0 : c1;
            if (h0 && h1) {
                stack[sp * MIW_BLOCK] = second_first ? c0 : c1; ++sp;
                next = second_first ? c1 : c0;
            } else if (!(h0 || h1)) {
                if (sp == 0) next = MIW_WALK_DONE; else { --sp; next = stack[sp * MIW_BLOCK]; }
            }
            if (Postpone && next < 0 && next != MIW_WALK_DONE && parked == MIW_WALK_DONE) {
                parked = next;                                   // first leaf: park it, keep descending
                if (sp == 0) next = MIW_WALK_DONE; else { --sp; next = stack[sp * MIW_BLOCK]; }
            }
            cur = next;
        }
        int32_t leaf;                                            // ---- leaf phase
        if (Postpone) {
            if (parked == MIW_WALK_DONE) break;                  // nothing parked: cur is DONE as well
            leaf = parked; parked = MIW_WALK_DONE;
        } else {
            if (cur == MIW_WALK_DONE) break;
            leaf = cur;
            if (sp == 0) cur = MIW_WALK_DONE; else { --sp; cur = stack[sp * MIW_BLOCK]; }
        }
        const uint32_t code = (uint32_t) ~leaf, first = code >> 4, count = (code & 15u) + 1u;
        bool found = false;
        for (uint32_t i = 0; i < count; ++i) {
            MIW_WALK_STATS_STEP(1);
            const Tri &tr = tri_at(first + i);
            float t, u, v;
            if (prim_intersect<Analytic>(tr, ctx, o, d, mint, maxt, t, u, v)) {
                if (AnyHit) { best.t = 0.f; best.tri = first + i; best.prim = tr.prim; found = true; break; }
                if (t < best.t || (t == best.t && tr.prim < best.prim)) {
                    best.t = t; best.u = u; best.v = v; best.tri = first + i; best.prim = tr.prim;
                    tmax = t;
                }
            }
        }
        if (AnyHit && found) break;
    }
    MIW_WALK_STATS_FLUSH(AnyHit ? 4 : 0);
    return best.tri != MIW_MISS;
}

// per-packet vertex bounds of a tiny scene, staged behind the leaf boxes (stage_to_lds)
__device__ __forceinline__ const TriBounds *packet_bounds(const SceneView &sc, TraceLds cfg, const uint4 *smem) {
    return reinterpret_cast<const TriBounds *>(smem + sc.tri_count * (sizeof(TriPacket) / 16) + cfg.leaves * MIW_LEAF_BOX_COPIES * (sizeof(LeafBox) / 16));
}
// Tiny scenes without a candidate filter: every lane sweeps every packet — wave-uniform LDS addresses
// (broadcast reads, no bank conflicts), no divergence. Same accept rule as bvh.h: min t, ties -> smaller prim id.
template <bool AnyHit>
__device__ __forceinline__ bool trace_brute(const SceneView &sc, TraceLds cfg, const uint4 *smem, V3 o, V3 d, float mint, float maxt, Hit &h) {
    const TriPacket *pk = reinterpret_cast<const TriPacket *>(smem);
    const TriBounds *tb = packet_bounds(sc, cfg, smem);
    h.t = MIW_INFINITY; h.u = h.v = 0.f; h.tri = MIW_MISS; h.prim = 0xffffffffu;
    bool any = false;
    for (uint32_t i = 0; i < sc.tri_count; ++i) {
        const TriPacket &k = pk[i];
        float t, u, v;
        bool hit = ray_intersect_triangle_edges(ld3(k.p0), ld3(k.e1), ld3(k.e2), o, d, mint, maxt, t, u, v) &&
                   hit_in_bounds(tb[i], o, d, t);
        if (AnyHit) {
            any = any || hit;
        } else if (hit && (t < h.t || (t == h.t && k.prim < h.prim))) {
            h.t = t; h.u = u; h.v = v; h.tri = i; h.prim = k.prim;
        }
    }
    if (AnyHit) { if (any) { h.t = 0.f; h.tri = 0; } return any; }
    return h.tri != MIW_MISS;
}

template <bool AnyHit, bool Analytic = true>
__device__ __forceinline__ bool trace_one(const SceneView &sc, TraceLds cfg, const uint4 *smem,
                                          V3 o, V3 d, float mint, float maxt, Hit &h) {
    if (cfg.brute) return trace_brute<AnyHit>(sc, cfg, smem, o, d, mint, maxt, h);
    RayPrep r = ray_prepare(o, d, mint, maxt);
    const BvhNode *lnodes = reinterpret_cast<const BvhNode *>(smem);
    const Tri *ltris = reinterpret_cast<const Tri *>(smem + cfg.nodes_staged * (sizeof(BvhNode) / 16));
    const BvhNode *gnodes = sc.nodes;
    const Tri *gtris = sc.tris;
    if (cfg.nodes_staged >= sc.node_count && cfg.tris_staged >= sc.tri_count) {
        // whole scene is LDS resident: pure ds_read traversal
        auto node_at = [lnodes](int32_t i) -> const BvhNode & { return lnodes[i]; };
        auto tri_at  = [ltris](uint32_t i) -> const Tri & { return ltris[i]; };
        return bvh_intersect<AnyHit>(node_at, tri_at, r, h, prim_ctx(sc));
    } else {
#if MIW_LDS_TOP
        uint32_t ns = cfg.nodes_staged;
        auto node_at = [lnodes, gnodes, ns](int32_t i) -> const BvhNode & {
            return (uint32_t) i < ns ? lnodes[i] : gnodes[i];
        };
#else
        auto node_at = [gnodes](int32_t i) -> const BvhNode & { return gnodes[i]; };   // global loads only (top of the tree lives in L1 / L2)
#endif
        auto tri_at = [gtris](uint32_t i) -> const Tri & { return gtris[i]; };
        if (cfg.stack) {
            int32_t *stack = reinterpret_cast<int32_t *>(const_cast<uint4 *>(smem) + cfg.stack16) + threadIdx.x;
#if MIW_WALK == 0
            return bvh_intersect_stack<AnyHit, Analytic>(node_at, tri_at, stack, o, d, mint, maxt, h, prim_ctx(sc));
#else
            return bvh_intersect_ww<AnyHit, Analytic, MIW_WALK == 2>(node_at, tri_at, stack, o, d, mint, maxt, h, prim_ctx(sc));
#endif
        }
        return bvh_intersect<AnyHit>(node_at, tri_at, r, h, prim_ctx(sc));
    }
}

// The resident plan's paired query: extension ray E and shadow ray S leave the same vertex
// (same origin, same mint). Tiny scenes (packets in LDS) are resolved in two phases:
//   1. a wave-uniform pass over the SAH leaves' padded boxes (broadcast LDS reads, the
//      conservative slab test of bvh.h — a triangle Moeller-Trumbore accepts is never culled)
//      leaves every lane two 64-bit candidate masks, one bit per triangle;
//   2. every lane pops its own candidates (E's first, then S's) and runs the exact
//      Moeller-Trumbore test on that packet (per-lane LDS address). The wave iterates
//      max-over-lanes(candidates) times instead of 2 x tri_count.
// Results are those of the full sweep: closest hit with ties to the smaller primitive id,
// "any triangle passes" for S.
// `Tiny` selects the code that is compiled in: the two-phase LDS query (tiny scenes) or the tree walks —
// one kernel per scene class keeps each one's register budget (occupancy) to what it needs.
template <int Tiny, bool Analytic = true>
__device__ __forceinline__ void trace2(const SceneView &sc, TraceLds cfg, const uint4 *smem,
                                       V3 o, float mint, V3 dE, float maxtE, bool hasE,
                                       V3 dS, float maxtS, bool hasS, F4 &hit_out, bool &occ_out) {
    Hit h; h.t = MIW_INFINITY; h.u = h.v = 0.f; h.tri = MIW_MISS; h.prim = 0xffffffffu;
    bool occ = false;
    if (Tiny && cfg.leaves) {
        // candidate masks: 32 bits when the scene has <= 32 triangles (Tiny == 2), else 64
        using Mask = typename std::conditional<Tiny == 2, uint32_t, unsigned long long>::type;
        auto lowest = [](Mask m) -> uint32_t {
            return Tiny == 2 ? (uint32_t) __ffs((int) (uint32_t) m) - 1u : (uint32_t) __ffsll((long long) m) - 1u;
        };
        const TriPacket *pk = reinterpret_cast<const TriPacket *>(smem);
        const LeafBox *lb = reinterpret_cast<const LeafBox *>(smem + sc.tri_count * (sizeof(TriPacket) / 16));
        const TriBounds *tb = packet_bounds(sc, cfg, smem);
        const FastRay rE = fast_ray(o, dE, mint), rS = fast_ray(o, dS, mint);
        const float wideE = widen(maxtE), wideS = widen(maxtS);
        Mask mE = 0, mS = 0;
#if MIW_OCTANT_BOXES
        const LeafBox *lbE = lb + ray_octant(rE), *lbS = lb + ray_octant(rS);   // each ray reads the copies of its own octant
        for (uint32_t i = 0; i < cfg.leaves; ++i) {
            const LeafBox &bE = lbE[8u * i], &bS = lbS[8u * i];
            const Mask bits = Tiny == 2 ? (Mask) bE.mask_lo : (Mask) (bE.mask_lo | ((unsigned long long) bE.mask_hi << 32));
            if (leaf_box_test_octant(bE, rE, wideE)) mE |= bits;
            if (leaf_box_test_octant(bS, rS, wideS)) mS |= bits;
        }
#else
        for (uint32_t i = 0; i < cfg.leaves; ++i) {
            const LeafBox &b = lb[i];                              // wave-uniform address
            const Mask bits = Tiny == 2 ? (Mask) b.mask_lo : (Mask) (b.mask_lo | ((unsigned long long) b.mask_hi << 32));
            if (leaf_box_test(b, rE, wideE)) mE |= bits;
            if (leaf_box_test(b, rS, wideS)) mS |= bits;
        }
#endif
        if (!hasE) mE = 0;
        if (!hasS) mS = 0;
        MIW_SECTION(1);
        uint32_t s_tri = 0; float s_t = 0.f;
        MIW_WALK_STATS_DECL;                                   // debug builds: candidate tests per ray and the loops' SIMT efficiency
        while (mE != 0) {                                      // closest hit of E over its candidates
            MIW_WALK_STATS_STEP(0);
            const uint32_t i = lowest(mE);
            mE &= mE - 1;
            const TriPacket &k = pk[i];
            float t, u, v;
            if (ray_intersect_triangle_edges(ld3(k.p0), ld3(k.e1), ld3(k.e2), o, dE, mint, maxtE, t, u, v) &&
                (t < h.t || (t == h.t && k.prim < h.prim))) { h.t = t; h.u = u; h.v = v; h.tri = i; h.prim = k.prim; }
        }
        MIW_SECTION(2);
        while (mS != 0) {                                      // any hit of S
            MIW_WALK_STATS_STEP(1);
            const uint32_t i = lowest(mS);
            mS &= mS - 1;
            const TriPacket &k = pk[i];
            float t, u, v;
            if (ray_intersect_triangle_edges(ld3(k.p0), ld3(k.e1), ld3(k.e2), o, dS, mint, maxtS, t, u, v)) {
                occ = true; mS = 0; s_tri = i; s_t = t;
            }
        }
        // The accept rule of shape.h, applied lazily: the loops above ran the bare Moeller-Trumbore test; only the
        // winners are checked against their triangle's bounds. A phantom (about one query in 10^9) sends its lane
        // through the full sweep with the rule inside, which is what the rule means.
        if (h.tri != MIW_MISS && !hit_in_bounds(tb[h.tri], o, dE, h.t)) trace_brute<false>(sc, cfg, smem, o, dE, mint, maxtE, h);
        if (occ && !hit_in_bounds(tb[s_tri], o, dS, s_t)) { Hit hs; occ = trace_brute<true>(sc, cfg, smem, o, dS, mint, maxtS, hs); }
        MIW_SECTION(3);
#if defined(MIW_WALK_STATS)
        atomicAdd(&g_walk_stats[1], ws_lane_[0]); atomicAdd(&g_walk_statsf[1], ws_wave_[0]); if (hasE) atomicAdd(&g_walk_stats[2], 1ull);
        atomicAdd(&g_walk_stats[5], ws_lane_[1]); atomicAdd(&g_walk_statsf[5], ws_wave_[1]); if (hasS) atomicAdd(&g_walk_stats[6], 1ull);
#endif
#if defined(MIW_VERIFY_FILTER)
        {   // debug builds: every filtered query against the full sweep; mismatching rays go to g_verify
            Hit hb; bool occ_b = false;
            if (hasE) trace_brute<false>(sc, cfg, smem, o, dE, mint, maxtE, hb); else { hb.tri = MIW_MISS; hb.t = MIW_INFINITY; }
            if (hasS) { Hit hs; occ_b = trace_brute<true>(sc, cfg, smem, o, dS, mint, maxtS, hs); }
            const bool badE = hasE && (hb.tri != h.tri || f2u(hb.t) != f2u(h.t)), badS = hasS && occ_b != occ;
            if (badE || badS) {
                const uint32_t k = atomicAdd(&g_verify_n, 1u);
                if (k < 16u) {
                    float *r = g_verify + k * 16;
                    r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = mint;
                    const V3 d = badE ? dE : dS;
                    r[4] = d.x; r[5] = d.y; r[6] = d.z; r[7] = badE ? maxtE : maxtS;
                    r[8] = badE ? 1.f : 2.f; r[9] = u2f(badE ? hb.tri : (uint32_t) occ_b); r[10] = u2f(badE ? h.tri : (uint32_t) occ);
                    r[11] = hb.t; r[12] = h.t;
                }
            }
        }
#endif
    } else if (Tiny) {                                         // tiny scene, filter switched off (MI_BVH_NO_LEAF_FILTER)
        if (hasE) trace_brute<false>(sc, cfg, smem, o, dE, mint, maxtE, h);
        if (hasS) { Hit hs; occ = trace_brute<true>(sc, cfg, smem, o, dS, mint, maxtS, hs); }
    } else {
        if (hasE) trace_one<false, Analytic>(sc, cfg, smem, o, dE, mint, maxtE, h);
        if (hasS) { Hit hs; occ = trace_one<true, Analytic>(sc, cfg, smem, o, dS, mint, maxtS, hs); }
    }
    hit_out.x = h.t; hit_out.y = h.u; hit_out.z = h.v; hit_out.w = u2f(h.tri);
    occ_out = occ;
}
