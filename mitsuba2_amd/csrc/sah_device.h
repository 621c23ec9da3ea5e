// Device run of the level-by-level binned-SAH builder (sah_levels.h): mi_bvh_build quality 0. One WORKGROUP per candidate of the
// current level — 1024 threads while a level holds few, large candidates (the top of the tree), one wavefront once it holds many —
// running the steps of sah_levels.h with their reductions in LDS:
//   k_sah_prims     per triangle: padded box + box centre (48-byte record), the identity index array
//   k_sah_decide    per candidate: box + centroid box of its range (wave shuffles, then ordered-uint LDS atomics), 3 x 16 bins in
//                   LDS (min / max / count atomics), one thread per axis runs sah_sweep_axis, thread 0 runs sah_decide
//   hipcub::DeviceScan::ExclusiveSum over the split flags  -> the breadth-first number of every new inner node
//   k_sah_apply     per candidate: sah_link (box + child reference into the parent's BvhNode), stable partition of its range into
//                   the next level's index array (ballots + a prefix over the workgroup's wavefronts), the two child candidates
//   k_sah_heights   per level, bottom-up: the BVH2 heights the 4-wide collapse asks for
//   k_sah_gather    triangles / vertex normals into leaf order
// The host reads two words back per level (inner nodes created, need_host). Same tree as the host builder of bvh_build.h, node
// for node (the decisions are sah_levels.h's functions; the CPU tier compares that restatement with the recursion, the GPU tier
// compares node counts, depth and films).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "sah_levels.h"
#include "lbvh_device.h"

namespace miw {

struct alignas(16) SahPrim { float lo[3], hi[3], cen[3]; uint32_t pad[3]; };
static_assert(sizeof(SahPrim) == 48, "SahPrim must be 48 bytes");
struct SahState { uint32_t n_inner, need_host; };

__global__ void k_sah_prims(const Tri *tris, uint32_t n, float pad, SahPrim *prim, uint32_t *idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SahBox b; float c[3];
    sah_prim(tris[i], pad, b, c);
    SahPrim p;
    for (int a = 0; a < 3; ++a) { p.lo[a] = b.lo[a]; p.hi[a] = b.hi[a]; p.cen[a] = c[a]; p.pad[a] = 0u; }
    prim[i] = p; idx[i] = i;
}

template <int BS>
__global__ __launch_bounds__(BS) void k_sah_decide(const SahCand *cand, uint32_t n_cand, const uint32_t *idx, const SahPrim *prim, uint32_t level,
                                                    uint32_t max_leaf, SahDecision *dec, uint32_t *flags, SahState *state) {
    __shared__ uint32_t s_box[12];                                            // box lo, box hi, centroid lo, centroid hi (ordered uints)
    __shared__ uint32_t s_lo[3][MIW_SAH_BINS][3], s_hi[3][MIW_SAH_BINS][3], s_cnt[3][MIW_SAH_BINS];
    __shared__ float s_cost[3]; __shared__ int s_bin[3];
    const uint32_t j = blockIdx.x;
    if (j >= n_cand) return;
    const SahCand c = cand[j];
    const uint32_t t = threadIdx.x, o_inf = lbvh_f2o(MIW_INFINITY), o_ninf = lbvh_f2o(-MIW_INFINITY);
    if (t < 12) s_box[t] = (t % 6u) < 3u ? o_inf : o_ninf;
    for (uint32_t k = t; k < 3u * MIW_SAH_BINS; k += BS) {
        const uint32_t a = k / MIW_SAH_BINS, b = k % MIW_SAH_BINS;
        for (int q = 0; q < 3; ++q) { s_lo[a][b][q] = o_inf; s_hi[a][b][q] = o_ninf; }
        s_cnt[a][b] = 0u;
    }
    __syncthreads();
    // ---- the range's padded box and centroid box ----
    float v[12] = { MIW_INFINITY, MIW_INFINITY, MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY,
                    MIW_INFINITY, MIW_INFINITY, MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY };
    for (uint32_t i = c.first + t; i < c.first + c.count; i += BS) {
        const SahPrim p = prim[idx[i]];
        for (int a = 0; a < 3; ++a) {
            v[a] = fminf(v[a], p.lo[a]); v[3 + a] = fmaxf(v[3 + a], p.hi[a]);
            v[6 + a] = fminf(v[6 + a], p.cen[a]); v[9 + a] = fmaxf(v[9 + a], p.cen[a]);
        }
    }
    for (int q = 0; q < 12; ++q)
        for (int off = 32; off > 0; off >>= 1) {
            const float w = __shfl_xor(v[q], off, 64);
            v[q] = (q % 6) < 3 ? fminf(v[q], w) : fmaxf(v[q], w);
        }
    if ((t & 63u) == 0u)
        for (int q = 0; q < 12; ++q) { if ((q % 6) < 3) atomicMin(&s_box[q], lbvh_f2o(v[q])); else atomicMax(&s_box[q], lbvh_f2o(v[q])); }
    __syncthreads();
    SahBox box, cbox;
    for (int a = 0; a < 3; ++a) { box.lo[a] = lbvh_o2f(s_box[a]); box.hi[a] = lbvh_o2f(s_box[3 + a]); cbox.lo[a] = lbvh_o2f(s_box[6 + a]); cbox.hi[a] = lbvh_o2f(s_box[9 + a]); }
    // ---- 16 bins per swept axis ----
    bool swept[3]; float scale[3];
    for (int a = 0; a < 3; ++a) { swept[a] = sah_axis_swept(cbox, a, c.count, level); scale[a] = swept[a] ? MIW_SAH_BINS / (cbox.hi[a] - cbox.lo[a]) : 0.f; }
    if (swept[0] || swept[1] || swept[2])
        for (uint32_t i = c.first + t; i < c.first + c.count; i += BS) {
            const SahPrim p = prim[idx[i]];
            for (int a = 0; a < 3; ++a) {
                if (!swept[a]) continue;
                const int b = sah_bin(p.cen[a], cbox.lo[a], scale[a]);
                for (int q = 0; q < 3; ++q) { atomicMin(&s_lo[a][b][q], lbvh_f2o(p.lo[q])); atomicMax(&s_hi[a][b][q], lbvh_f2o(p.hi[q])); }
                atomicAdd(&s_cnt[a][b], 1u);
            }
        }
    __syncthreads();
    if (t < 3u) {                                                             // one thread per axis: the sweep
        float cost = MIW_INFINITY; int bin = -1;
        if (swept[t]) {
            SahBox bb[MIW_SAH_BINS]; uint32_t bc[MIW_SAH_BINS];
            for (int b = 0; b < MIW_SAH_BINS; ++b) {
                for (int q = 0; q < 3; ++q) { bb[b].lo[q] = lbvh_o2f(s_lo[t][b][q]); bb[b].hi[q] = lbvh_o2f(s_hi[t][b][q]); }
                bc[b] = s_cnt[t][b];
            }
            sah_sweep_axis(bb, bc, cost, bin);
        }
        s_cost[t] = cost; s_bin[t] = bin;
    }
    __syncthreads();
    if (t == 0u) {
        const float cost[3] = { s_cost[0], s_cost[1], s_cost[2] }; const int bin[3] = { s_bin[0], s_bin[1], s_bin[2] };
        SahDecision d;
        const int r = sah_decide(box, cbox, c.count, max_leaf, cost, bin, d);
        if (r == 2 || (level == 0u && r == 0)) atomicOr(&state->need_host, 1u);
        if (r == 1) for (uint32_t b = 0; b <= d.bin; ++b) d.n_left += s_cnt[d.axis][b];
        dec[j] = d; flags[j] = d.split;
    }
}

__global__ void k_sah_totals(const uint32_t *flags, const uint32_t *rank, uint32_t n_cand, SahState *state) {
    if (blockIdx.x == 0 && threadIdx.x == 0) state->n_inner = rank[n_cand - 1u] + flags[n_cand - 1u];
}

template <int BS>
__global__ __launch_bounds__(BS) void k_sah_apply(const SahCand *cand, uint32_t n_cand, const uint32_t *idx, uint32_t *idx_next, const SahPrim *prim,
                                                   const SahDecision *dec, const uint32_t *rank, uint32_t base, BvhNode *nodes, SahCand *cand_next) {
    constexpr int NW = BS / 64;
    __shared__ uint32_t s_l[NW], s_v[NW];
    const uint32_t j = blockIdx.x;
    if (j >= n_cand) return;
    const SahCand c = cand[j]; const SahDecision d = dec[j];
    const uint32_t t = threadIdx.x, w = t >> 6;
    const int32_t me = (int32_t) (base + rank[j]);
    if (t == 0u) {
        sah_link(nodes, c, d, me);
        if (d.split) {
            cand_next[2u * rank[j]] = SahCand{ c.first, d.n_left, me, 0u };
            cand_next[2u * rank[j] + 1u] = SahCand{ c.first + d.n_left, c.count - d.n_left, me, 1u };
        }
    }
    if (!d.split) { for (uint32_t i = c.first + t; i < c.first + c.count; i += BS) idx_next[i] = idx[i]; return; }
    uint32_t l_base = c.first, r_base = c.first + d.n_left;
    for (uint32_t i0 = 0; i0 < c.count; i0 += BS) {                           // stable partition, BS positions per trip
        const bool valid = i0 + t < c.count;
        const uint32_t id = valid ? idx[c.first + i0 + t] : 0u;
        const bool left = valid && sah_bin(prim[id].cen[d.axis], d.clo, d.scale) <= (int) d.bin;
        const unsigned long long bl = __ballot(left), bv = __ballot(valid);
        if ((t & 63u) == 0u) { s_l[w] = (uint32_t) __popcll(bl); s_v[w] = (uint32_t) __popcll(bv); }
        __syncthreads();
        uint32_t pl = 0, pv = 0, tl = 0, tv = 0;
        for (int k = 0; k < NW; ++k) { const uint32_t a = s_l[k], b = s_v[k]; if ((uint32_t) k < w) { pl += a; pv += b; } tl += a; tv += b; }
        const unsigned long long below = (1ull << (t & 63u)) - 1ull;
        const uint32_t rl = (uint32_t) __popcll(bl & below), rv = (uint32_t) __popcll(bv & below);
        if (valid) idx_next[left ? l_base + pl + rl : r_base + (pv - pl) + (rv - rl)] = id;
        l_base += tl; r_base += tv - tl;
        __syncthreads();
    }
}

__global__ void k_sah_heights(const BvhNode *nodes, uint32_t first, uint32_t end, uint32_t *height) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    const BvhNode &nd = nodes[i];
    const uint32_t h0 = nd.child0 >= 0 ? height[nd.child0] : 0u, h1 = nd.child1 >= 0 ? height[nd.child1] : 0u;
    height[i] = 1u + (h0 > h1 ? h0 : h1);
}

__global__ void k_sah_gather(const Tri *tris_in, const float *vn_in, const uint32_t *idx, uint32_t n, Tri *tris_out, float *vn_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = idx[i];
    tris_out[i] = tris_in[src];
    if (vn_in) for (int k = 0; k < 9; ++k) vn_out[(size_t) i * 9 + k] = vn_in[(size_t) src * 9 + k];
}

} // namespace miw
