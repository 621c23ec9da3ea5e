// Device run of the level-by-level binned-SAH builder (sah_levels.h): mi_bvh_build quality 0. One WORKGROUP per candidate of the
// current level — 1024 threads for a candidate of more than 8192 triangles or while a level holds at most 512 candidates, one
// wavefront otherwise (a level that mixes sizes is launched twice, each launch skipping the other's candidates) —
// running the steps of sah_levels.h with their reductions in LDS:
//   k_sah_prims     per triangle: padded box + box centre (48-byte record), the identity index array
//   k_sah_decide    per candidate: box + centroid box of its range (wave shuffles, then ordered-uint LDS atomics), 3 x 16 bins in
//                   LDS (min / max / count atomics), one thread per axis runs sah_sweep_axis, thread 0 runs sah_decide
//   rocprim::exclusive_scan over the split flags  -> the breadth-first number of every new inner node
//   k_sah_apply     per candidate: sah_link (box + child reference into the parent's BvhNode), stable partition of its range into
//                   the next level's index array (ballots + a prefix over the workgroup's wavefronts), the two child candidates
//   k_sah_heights   per level, bottom-up: the BVH2 heights the 4-wide collapse asks for
//   k_sah_gather    triangles / vertex normals into leaf order
// The host reads two words back per level (inner nodes created, need_host). Same tree as the host builder of bvh_build.h, node
// for node (the decisions are sah_levels.h's functions; the CPU tier compares that restatement with the recursion, the GPU tier
// compares node counts, depth and films).
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "sah_levels.h"
#include "lbvh_device.h"

namespace miw {

struct alignas(16) SahPrim { float lo[3], hi[3], cen[3]; uint32_t pad[3]; };
static_assert(sizeof(SahPrim) == 48, "SahPrim must be 48 bytes");
struct SahState { uint32_t n_inner, need_host, max_count; };     // max_count: the largest candidate of the NEXT level (k_sah_apply)
#define MIW_SAH_BIG 2048u          /* candidates above this many triangles get a 1024-thread workgroup whatever the level holds */

__global__ void k_sah_prims(const Tri *tris, uint32_t n, float pad, SahPrim *prim, uint32_t *idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SahBox b; float c[3];
    sah_prim(tris[i], pad, b, c);
    SahPrim p;
    for (int a = 0; a < 3; ++a) { p.lo[a] = b.lo[a]; p.hi[a] = b.hi[a]; p.cen[a] = c[a]; p.pad[a] = 0u; }
    prim[i] = p; idx[i] = i;
}

// the end of a candidate's decision, from its bins in LDS: one thread per axis runs the sweep, thread 0 decides (shared by
// k_sah_decide and k_sah_huge_finish; every thread of the workgroup calls it)
__device__ __forceinline__ void sah_decide_tail(uint32_t t, const bool swept[3], const SahBox &box, const SahBox &cbox, uint32_t count, uint32_t level, uint32_t max_leaf,
                                                uint32_t (*s_lo)[MIW_SAH_BINS][3], uint32_t (*s_hi)[MIW_SAH_BINS][3], uint32_t (*s_cnt)[MIW_SAH_BINS],
                                                float (*s_ra)[MIW_SAH_BINS], uint32_t (*s_rc)[MIW_SAH_BINS], float *s_cost, int *s_bin,
                                                SahDecision *dec_out, uint32_t *flag_out, SahState *state) {
    if (t < 3u) {                                                             // one thread per axis: the sweep
        float cost = MIW_INFINITY; int bin = -1;
        if (swept[t])
            sah_sweep_axis([&](int b) { SahBox x; for (int q = 0; q < 3; ++q) { x.lo[q] = lbvh_o2f(s_lo[t][b][q]); x.hi[q] = lbvh_o2f(s_hi[t][b][q]); } return x; },
                           [&](int b) { return s_cnt[t][b]; }, s_ra[t], s_rc[t], cost, bin);
        s_cost[t] = cost; s_bin[t] = bin;
    }
    __syncthreads();
    if (t == 0u) {
        const float cost[3] = { s_cost[0], s_cost[1], s_cost[2] }; const int bin[3] = { s_bin[0], s_bin[1], s_bin[2] };
        SahDecision d;
        const int r = sah_decide(box, cbox, count, max_leaf, cost, bin, d);
        if (r == 2 || (level == 0u && r == 0)) atomicOr(&state->need_host, 1u);
        if (r == 1) for (uint32_t b = 0; b <= d.bin; ++b) d.n_left += s_cnt[d.axis][b];
        *dec_out = d; *flag_out = d.split;
    }
}

template <int BS>
__global__ __launch_bounds__(BS) void k_sah_decide(const SahCand *cand, uint32_t n_cand, const uint32_t *idx, const SahPrim *prim, uint32_t level,
                                                    uint32_t max_leaf, SahDecision *dec, uint32_t *flags, SahState *state, uint32_t count_lo, uint32_t count_hi) {
    __shared__ uint32_t s_box[12];                                            // box lo, box hi, centroid lo, centroid hi (ordered uints)
    __shared__ uint32_t s_lo[3][MIW_SAH_BINS][3], s_hi[3][MIW_SAH_BINS][3], s_cnt[3][MIW_SAH_BINS];
    // (the sixteen wavefronts of a 1024-thread workgroup bin into their OWN copy first — 16 bins x 3 axes is what 1024 threads' atomics
    // would otherwise queue up on — and fold the copies into s_lo / s_hi / s_cnt afterwards)
    constexpr int NWB = BS / 64;
    __shared__ uint32_t w_lo[NWB > 1 ? NWB : 1][3][MIW_SAH_BINS][3], w_hi[NWB > 1 ? NWB : 1][3][MIW_SAH_BINS][3], w_cnt[NWB > 1 ? NWB : 1][3][MIW_SAH_BINS];
    __shared__ float s_cost[3]; __shared__ int s_bin[3];
    __shared__ float s_ra[3][MIW_SAH_BINS]; __shared__ uint32_t s_rc[3][MIW_SAH_BINS];     // the sweep's suffix areas / counts
    const uint32_t j = blockIdx.x;
    if (j >= n_cand) return;
    const SahCand c = cand[j];
    if (c.count == 0u) {                                                      // (cannot happen: sah_sweep_axis never leaves a side empty) — still a defined record, and the host builder's turn
        if (threadIdx.x == 0u) { SahDecision d; memset(&d, 0, sizeof d); dec[j] = d; flags[j] = 0u; atomicOr(&state->need_host, 1u); }
        return;
    }
    if (c.count <= count_lo || c.count > count_hi) return;                    // (the other launch of this level takes it: a level is run twice when it mixes sizes)
    const uint32_t t = threadIdx.x, o_inf = lbvh_f2o(MIW_INFINITY), o_ninf = lbvh_f2o(-MIW_INFINITY);
    if (t < 12) s_box[t] = (t % 6u) < 3u ? o_inf : o_ninf;
    for (uint32_t k = t; k < 3u * MIW_SAH_BINS; k += BS) {
        const uint32_t a = k / MIW_SAH_BINS, b = k % MIW_SAH_BINS;
        for (int q = 0; q < 3; ++q) { s_lo[a][b][q] = o_inf; s_hi[a][b][q] = o_ninf; }
        s_cnt[a][b] = 0u;
    }
    if (NWB > 1)
        for (uint32_t k = t; k < (uint32_t) NWB * 3u * MIW_SAH_BINS; k += BS) {
            const uint32_t ww = k / (3u * MIW_SAH_BINS), a = (k / MIW_SAH_BINS) % 3u, b = k % MIW_SAH_BINS;
            for (int q = 0; q < 3; ++q) { w_lo[ww][a][b][q] = o_inf; w_hi[ww][a][b][q] = o_ninf; }
            w_cnt[ww][a][b] = 0u;
        }
    __syncthreads();
    // ---- the range's padded box and centroid box ----
    float v[12] = { MIW_INFINITY, MIW_INFINITY, MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY,
                    MIW_INFINITY, MIW_INFINITY, MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY };
    // (a thread's trip is idx -> record, two dependent loads a microsecond each: four independent chains per trip, or the top
    // levels — one workgroup on 0.9 M triangles — are nothing but that latency)
    const uint32_t end = c.first + c.count;
    for (uint32_t i0 = c.first + t; i0 < end; i0 += 4u * BS) {
        uint32_t id[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t i = i0 + (uint32_t) k * BS; id[k] = idx[i < end ? i : end - 1u]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const SahPrim p = prim[id[k]];                                    // (a clamped slot repeats the range's last triangle: min / max do not mind)
            for (int a = 0; a < 3; ++a) {
                v[a] = fminf(v[a], p.lo[a]); v[3 + a] = fmaxf(v[3 + a], p.hi[a]);
                v[6 + a] = fminf(v[6 + a], p.cen[a]); v[9 + a] = fmaxf(v[9 + a], p.cen[a]);
            }
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
        for (uint32_t i0 = c.first + t; i0 < end; i0 += 4u * BS) {
            uint32_t id[4]; SahPrim p4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const uint32_t i = i0 + (uint32_t) k * BS; id[k] = idx[i < end ? i : end - 1u]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) p4[k] = prim[id[k]];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i0 + (uint32_t) k * BS >= end) continue;                  // (a clamped slot must not be counted twice)
                const SahPrim &p = p4[k];
                for (int a = 0; a < 3; ++a) {
                    if (!swept[a]) continue;
                    const int b = sah_bin(p.cen[a], cbox.lo[a], scale[a]);
                    uint32_t *lo = NWB > 1 ? w_lo[t >> 6][a][b] : s_lo[a][b], *hi = NWB > 1 ? w_hi[t >> 6][a][b] : s_hi[a][b];
                    for (int q = 0; q < 3; ++q) { atomicMin(&lo[q], lbvh_f2o(p.lo[q])); atomicMax(&hi[q], lbvh_f2o(p.hi[q])); }
                    atomicAdd(NWB > 1 ? &w_cnt[t >> 6][a][b] : &s_cnt[a][b], 1u);
                }
            }
        }
    __syncthreads();
    if (NWB > 1) {                                                            // fold the wavefronts' copies (min / max / sum: order-free)
        for (uint32_t k = t; k < 3u * MIW_SAH_BINS; k += BS) {
            const uint32_t a = k / MIW_SAH_BINS, b = k % MIW_SAH_BINS;
            uint32_t n = 0u, l3[3] = { o_inf, o_inf, o_inf }, h3[3] = { o_ninf, o_ninf, o_ninf };
            for (int ww = 0; ww < NWB; ++ww) {
                n += w_cnt[ww][a][b];
                for (int q = 0; q < 3; ++q) { l3[q] = l3[q] < w_lo[ww][a][b][q] ? l3[q] : w_lo[ww][a][b][q]; h3[q] = h3[q] > w_hi[ww][a][b][q] ? h3[q] : w_hi[ww][a][b][q]; }
            }
            s_cnt[a][b] = n;
            for (int q = 0; q < 3; ++q) { s_lo[a][b][q] = l3[q]; s_hi[a][b][q] = h3[q]; }
        }
        __syncthreads();
    }
    sah_decide_tail(t, swept, box, cbox, c.count, level, max_leaf, s_lo, s_hi, s_cnt, s_ra, s_rc, s_cost, s_bin, &dec[j], &flags[j], state);
}

// ---- candidates of more than MIW_SAH_HUGE triangles: the same decision from SEVERAL workgroups (round 5) ----
// One workgroup per candidate leaves the top of the tree to one CU: level 0 of the 0.9 M-triangle interior binned 911 362 triangles
// on one CU in 14.4 ms, the 13 dispatches that still held such candidates took 33.7 of the builder's 41 ms (profiles/r04_c4_tree_*).
// Every reduction of the decision is order-free (min / max over ordered uints, counts), so a candidate's range can be cut into
// slices of MIW_SAH_SLICE triangles, one 1024-thread workgroup each, folding into one record per candidate in global memory with
// the same ordered-uint atomics — the bins come out bit for bit as one workgroup computes them, and so does the tree:
//   k_sah_huge_init   the records of the level's candidates (inf / -inf / 0)
//   k_sah_huge_box    per slice: padded box + centroid box  -> record (12 atomics per workgroup)
//   k_sah_huge_bins   per slice: 3 x 16 bins in LDS (per-wavefront copies, folded) -> record (336 atomics per workgroup)
//   k_sah_huge_finish per candidate: the record into LDS, then the tail every candidate runs (sweep, sah_decide)
// Grid = (candidates of the level, slices of its largest candidate): workgroups without a slice, and all workgroups of a
// candidate at or below MIW_SAH_HUGE (which k_sah_decide takes), retire at once.
#define MIW_SAH_HUGE 32768u
#define MIW_SAH_SLICE 16384u
#define MIW_SAH_HUGE_CANDS 4096u    /* levels with more candidates than this keep the one-workgroup kernels (their records: 1.4 KB each) */
struct SahHuge { uint32_t box[12]; uint32_t lo[3][MIW_SAH_BINS][3], hi[3][MIW_SAH_BINS][3], cnt[3][MIW_SAH_BINS]; };

__global__ void k_sah_huge_init(SahHuge *hg, uint32_t n_cand) {
    constexpr uint32_t W = sizeof(SahHuge) / 4u;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand * W) return;
    const uint32_t k = i % W, o_inf = lbvh_f2o(MIW_INFINITY), o_ninf = lbvh_f2o(-MIW_INFINITY);
    uint32_t v;
    if (k < 12u) v = (k % 6u) < 3u ? o_inf : o_ninf;
    else if (k < 12u + 3u * MIW_SAH_BINS * 3u) v = o_inf;
    else if (k < 12u + 6u * MIW_SAH_BINS * 3u) v = o_ninf;
    else v = 0u;
    reinterpret_cast<uint32_t *>(hg)[i] = v;
}

__global__ __launch_bounds__(1024) void k_sah_huge_box(const SahCand *cand, uint32_t n_cand, const uint32_t *idx, const SahPrim *prim, SahHuge *hg) {
    __shared__ uint32_t s_box[12];
    const uint32_t j = blockIdx.x;
    if (j >= n_cand) return;
    const SahCand c = cand[j];
    const uint32_t s0 = blockIdx.y * MIW_SAH_SLICE;
    if (c.count <= MIW_SAH_HUGE || s0 >= c.count) return;
    const uint32_t t = threadIdx.x, o_inf = lbvh_f2o(MIW_INFINITY), o_ninf = lbvh_f2o(-MIW_INFINITY);
    if (t < 12) s_box[t] = (t % 6u) < 3u ? o_inf : o_ninf;
    __syncthreads();
    float v[12] = { MIW_INFINITY, MIW_INFINITY, MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY,
                    MIW_INFINITY, MIW_INFINITY, MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY };
    const uint32_t begin = c.first + s0, end = c.first + (s0 + MIW_SAH_SLICE < c.count ? s0 + MIW_SAH_SLICE : c.count);
    for (uint32_t i0 = begin + t; i0 < end; i0 += 4u * 1024u) {
        uint32_t id[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t i = i0 + (uint32_t) k * 1024u; id[k] = idx[i < end ? i : end - 1u]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const SahPrim p = prim[id[k]];
            for (int a = 0; a < 3; ++a) {
                v[a] = fminf(v[a], p.lo[a]); v[3 + a] = fmaxf(v[3 + a], p.hi[a]);
                v[6 + a] = fminf(v[6 + a], p.cen[a]); v[9 + a] = fmaxf(v[9 + a], p.cen[a]);
            }
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
    if (t < 12u) { if ((t % 6u) < 3u) atomicMin(&hg[j].box[t], s_box[t]); else atomicMax(&hg[j].box[t], s_box[t]); }
}

__global__ __launch_bounds__(1024) void k_sah_huge_bins(const SahCand *cand, uint32_t n_cand, const uint32_t *idx, const SahPrim *prim, uint32_t level, SahHuge *hg) {
    constexpr int NWB = 16;
    __shared__ uint32_t w_lo[NWB][3][MIW_SAH_BINS][3], w_hi[NWB][3][MIW_SAH_BINS][3], w_cnt[NWB][3][MIW_SAH_BINS];
    const uint32_t j = blockIdx.x;
    if (j >= n_cand) return;
    const SahCand c = cand[j];
    const uint32_t s0 = blockIdx.y * MIW_SAH_SLICE;
    if (c.count <= MIW_SAH_HUGE || s0 >= c.count) return;
    const uint32_t t = threadIdx.x, o_inf = lbvh_f2o(MIW_INFINITY), o_ninf = lbvh_f2o(-MIW_INFINITY);
    for (uint32_t k = t; k < (uint32_t) NWB * 3u * MIW_SAH_BINS; k += 1024u) {
        const uint32_t ww = k / (3u * MIW_SAH_BINS), a = (k / MIW_SAH_BINS) % 3u, b = k % MIW_SAH_BINS;
        for (int q = 0; q < 3; ++q) { w_lo[ww][a][b][q] = o_inf; w_hi[ww][a][b][q] = o_ninf; }
        w_cnt[ww][a][b] = 0u;
    }
    __syncthreads();
    SahBox cbox;
    for (int a = 0; a < 3; ++a) { cbox.lo[a] = lbvh_o2f(hg[j].box[6 + a]); cbox.hi[a] = lbvh_o2f(hg[j].box[9 + a]); }
    bool swept[3]; float scale[3];
    for (int a = 0; a < 3; ++a) { swept[a] = sah_axis_swept(cbox, a, c.count, level); scale[a] = swept[a] ? MIW_SAH_BINS / (cbox.hi[a] - cbox.lo[a]) : 0.f; }
    if (!(swept[0] || swept[1] || swept[2])) return;
    const uint32_t begin = c.first + s0, end = c.first + (s0 + MIW_SAH_SLICE < c.count ? s0 + MIW_SAH_SLICE : c.count);
    for (uint32_t i0 = begin + t; i0 < end; i0 += 4u * 1024u) {
        uint32_t id[4]; SahPrim p4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t i = i0 + (uint32_t) k * 1024u; id[k] = idx[i < end ? i : end - 1u]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) p4[k] = prim[id[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + (uint32_t) k * 1024u >= end) continue;
            const SahPrim &p = p4[k];
            for (int a = 0; a < 3; ++a) {
                if (!swept[a]) continue;
                const int b = sah_bin(p.cen[a], cbox.lo[a], scale[a]);
                uint32_t *lo = w_lo[t >> 6][a][b], *hi = w_hi[t >> 6][a][b];
                for (int q = 0; q < 3; ++q) { atomicMin(&lo[q], lbvh_f2o(p.lo[q])); atomicMax(&hi[q], lbvh_f2o(p.hi[q])); }
                atomicAdd(&w_cnt[t >> 6][a][b], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t k = t; k < 3u * MIW_SAH_BINS; k += 1024u) {                  // fold the wavefronts' copies, then into the candidate's record
        const uint32_t a = k / MIW_SAH_BINS, b = k % MIW_SAH_BINS;
        uint32_t n = 0u, l3[3] = { o_inf, o_inf, o_inf }, h3[3] = { o_ninf, o_ninf, o_ninf };
        for (int ww = 0; ww < NWB; ++ww) {
            n += w_cnt[ww][a][b];
            for (int q = 0; q < 3; ++q) { l3[q] = l3[q] < w_lo[ww][a][b][q] ? l3[q] : w_lo[ww][a][b][q]; h3[q] = h3[q] > w_hi[ww][a][b][q] ? h3[q] : w_hi[ww][a][b][q]; }
        }
        if (n) {
            atomicAdd(&hg[j].cnt[a][b], n);
            for (int q = 0; q < 3; ++q) { atomicMin(&hg[j].lo[a][b][q], l3[q]); atomicMax(&hg[j].hi[a][b][q], h3[q]); }
        }
    }
}

__global__ __launch_bounds__(64) void k_sah_huge_finish(const SahCand *cand, uint32_t n_cand, uint32_t level, uint32_t max_leaf, const SahHuge *hg,
                                                        SahDecision *dec, uint32_t *flags, SahState *state) {
    __shared__ uint32_t s_lo[3][MIW_SAH_BINS][3], s_hi[3][MIW_SAH_BINS][3], s_cnt[3][MIW_SAH_BINS];
    __shared__ float s_cost[3]; __shared__ int s_bin[3];
    __shared__ float s_ra[3][MIW_SAH_BINS]; __shared__ uint32_t s_rc[3][MIW_SAH_BINS];
    const uint32_t j = blockIdx.x;
    if (j >= n_cand) return;
    const SahCand c = cand[j];
    if (c.count <= MIW_SAH_HUGE) return;
    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k < 3u * MIW_SAH_BINS; k += 64u) {
        const uint32_t a = k / MIW_SAH_BINS, b = k % MIW_SAH_BINS;
        for (int q = 0; q < 3; ++q) { s_lo[a][b][q] = hg[j].lo[a][b][q]; s_hi[a][b][q] = hg[j].hi[a][b][q]; }
        s_cnt[a][b] = hg[j].cnt[a][b];
    }
    __syncthreads();
    SahBox box, cbox;
    for (int a = 0; a < 3; ++a) { box.lo[a] = lbvh_o2f(hg[j].box[a]); box.hi[a] = lbvh_o2f(hg[j].box[3 + a]); cbox.lo[a] = lbvh_o2f(hg[j].box[6 + a]); cbox.hi[a] = lbvh_o2f(hg[j].box[9 + a]); }
    bool swept[3];
    for (int a = 0; a < 3; ++a) swept[a] = sah_axis_swept(cbox, a, c.count, level);
    sah_decide_tail(t, swept, box, cbox, c.count, level, max_leaf, s_lo, s_hi, s_cnt, s_ra, s_rc, s_cost, s_bin, &dec[j], &flags[j], state);
}

__global__ void k_sah_totals(const uint32_t *flags, const uint32_t *rank, uint32_t n_cand, SahState *state) {
    if (blockIdx.x == 0 && threadIdx.x == 0) state->n_inner = rank[n_cand - 1u] + flags[n_cand - 1u];
}

template <int BS>
__global__ __launch_bounds__(BS) void k_sah_apply(const SahCand *cand, uint32_t n_cand, const uint32_t *idx, uint32_t *idx_next, const SahPrim *prim,
                                                   const SahDecision *dec, const uint32_t *rank, uint32_t base, BvhNode *nodes, SahCand *cand_next,
                                                   SahState *state, uint32_t count_lo, uint32_t count_hi, uint32_t max_nodes) {
    constexpr int NW = BS / 64;
    __shared__ uint32_t s_l[NW], s_v[NW];
    const uint32_t j = blockIdx.x;
    if (j >= n_cand) return;
    const SahCand c = cand[j];
    if (c.count <= count_lo || c.count > count_hi) return;
    const SahDecision d = dec[j];
    const uint32_t t = threadIdx.x, w = t >> 6;
    const int32_t me = (int32_t) (base + rank[j]);
    // n triangles make at most n - 1 inner nodes and n candidates per level: a record or a candidate beyond that can only come from a
    // decision that left one side empty (sah_sweep_axis never does) — refuse to write it and hand the scene to the host builder
    if (d.split && ((uint32_t) me + 1u >= max_nodes || 2u * rank[j] + 1u >= max_nodes)) { if (t == 0u) atomicOr(&state->need_host, 1u); return; }
    if (t == 0u) {
        sah_link(nodes, c, d, me);
        if (d.split) {
            cand_next[2u * rank[j]] = SahCand{ c.first, d.n_left, me, 0u };
            cand_next[2u * rank[j] + 1u] = SahCand{ c.first + d.n_left, c.count - d.n_left, me, 1u };
            const uint32_t larger = d.n_left > c.count - d.n_left ? d.n_left : c.count - d.n_left;
            if (larger > MIW_SAH_BIG) atomicMax(&state->max_count, larger);
        }
    }
    if (!d.split) { for (uint32_t i = c.first + t; i < c.first + c.count; i += BS) idx_next[i] = idx[i]; return; }
    // stable partition, 4 BS positions per trip: a thread takes four consecutive positions (four independent idx -> centroid load
    // chains), a wavefront 256, the workgroup's wavefronts consecutive runs; ranks from four ballots + a prefix over the wavefronts
    uint32_t l_base = c.first, r_base = c.first + d.n_left;
    for (uint32_t i0 = 0; i0 < c.count; i0 += 4u * BS) {
        uint32_t id[4]; bool valid[4], left[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t p = i0 + 4u * t + (uint32_t) k; valid[k] = p < c.count; id[k] = valid[k] ? idx[c.first + p] : 0u; }
#pragma unroll
        for (int k = 0; k < 4; ++k) left[k] = valid[k] && sah_bin(prim[id[k]].cen[d.axis], d.clo, d.scale) <= (int) d.bin;
        const unsigned long long below = (1ull << (t & 63u)) - 1ull;
        uint32_t rl = 0, rv = 0, wl = 0, wv = 0;                              // lefts / valid positions before this thread in its wavefront; the wavefront's totals
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long bl = __ballot(left[k]), bv = __ballot(valid[k]);
            rl += (uint32_t) __popcll(bl & below); rv += (uint32_t) __popcll(bv & below);
            wl += (uint32_t) __popcll(bl); wv += (uint32_t) __popcll(bv);
        }
        if ((t & 63u) == 0u) { s_l[w] = wl; s_v[w] = wv; }
        __syncthreads();
        uint32_t pl = 0, pv = 0, tl = 0, tv = 0;
        for (int k = 0; k < NW; ++k) { const uint32_t a = s_l[k], b = s_v[k]; if ((uint32_t) k < w) { pl += a; pv += b; } tl += a; tv += b; }
        uint32_t at_l = l_base + pl + rl, at_r = r_base + (pv - pl) + (rv - rl);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (valid[k]) { if (left[k]) idx_next[at_l++] = id[k]; else idx_next[at_r++] = id[k]; }
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
