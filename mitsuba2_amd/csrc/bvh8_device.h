// Device-side collapse of the device-built SAH BVH2 (sah_device.h) into the 8-wide quantised tree of miw/bvh8.h, so that a
// quality-0 mi_bvh_build never leaves the GPU. Three steps, all of them the functions the host builder (bvh8_build.h) runs:
//   k_bvh8_dp       bottom-up over the BVH2's levels (the builder numbers its nodes breadth-first, so the children of a level's
//                   nodes lie in later levels): bvh8_dp_node — the dynamic programme that picks the topology;
//   k_bvh8_level    top-down, level by level (like bvh4_device.h): the frontier of level L — BVH2 nodes that become 8-wide nodes —
//                   is turned into the Bvh8Nodes of that level and the frontier of level L + 1. One thread per node runs
//                   bvh8_collapse_node; the inner children of a node get consecutive numbers and the triangles of its leaf
//                   slots consecutive positions in the tree's own triangle order (both handed out per wavefront: one prefix sum
//                   over the lanes' counts + one atomic), and every leaf slot writes its run of the permutation;
//   k_bvh8_gather   triangle records and vertex normals into that order.
// Only the ORDER of the nodes inside a level and of the nodes' triangle runs is the device's own (atomics); the nodes are the
// host builder's. Stands where the reference's GPU path builds and compacts its acceleration structure on the device
// (include/mitsuba/render/optix/shapes.h:72-228).
#pragma once
#include <hip/hip_runtime.h>
#include "bvh8_build.h"

namespace miw {

struct Bvh8Levels {                      // device bookkeeping, zeroed before the first level (count[0] = 1 is set by the launcher)
    uint32_t count[64], start[65];
    uint32_t tri_next, failed;
};

__global__ void k_bvh8_dp(const BvhNode *n2, Bvh8Dp *dp, uint32_t first, uint32_t end) {
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    bvh8_dp_node(n2, dp, (int32_t) i);
}

__global__ __launch_bounds__(256) void k_bvh8_level(const BvhNode *n2, const Bvh8Dp *dp, const int32_t *in, int32_t *out, Bvh8Levels *lv,
                                                    Bvh8Node *nodes8, uint32_t *perm, uint32_t level, uint32_t max_nodes, uint32_t max_tris) {
    const uint32_t n_in = lv->count[level];
    if (blockIdx.x * blockDim.x >= n_in) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, start = lv->start[level], start_next = start + n_in;
    if (i == 0) lv->start[level + 1] = start_next;             // read by the next launch only
    const bool live = i < n_in;
    Bvh8Node n; int32_t kid_ref[8], leaf_code[8];
    uint32_t inner = 0, ntri = 0;
    if (live) {
        const int nt = bvh8_collapse_node(n2, dp, in[i], 8, n, kid_ref, leaf_code);
        if (nt < 0) { atomicOr(&lv->failed, 1u); for (int s = 0; s < 8; ++s) { kid_ref[s] = -1; leaf_code[s] = 0; } }
        else ntri = (uint32_t) nt;
        for (int s = 0; s < 8; ++s) inner += kid_ref[s] >= 0 ? 1u : 0u;
    }
    // slots of the next level and positions in the triangle order: exclusive prefix sums over the wavefront, one atomic each
    uint32_t incl = inner, incl_t = ntri;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = (uint32_t) __shfl_up((int) incl, o, 64), b = (uint32_t) __shfl_up((int) incl_t, o, 64);
        if ((threadIdx.x & 63u) >= (uint32_t) o) { incl += a; incl_t += b; }
    }
    const uint32_t total = (uint32_t) __shfl((int) incl, 63, 64), total_t = (uint32_t) __shfl((int) incl_t, 63, 64);
    uint32_t base = 0, base_t = 0;
    if ((threadIdx.x & 63u) == 63u) {
        if (total) base = atomicAdd(&lv->count[level + 1], total);
        if (total_t) base_t = atomicAdd(&lv->tri_next, total_t);
    }
    base = (uint32_t) __shfl((int) base, 63, 64); base_t = (uint32_t) __shfl((int) base_t, 63, 64);
    if (!live) return;
    uint32_t slot = base + incl - inner;
    const uint32_t tri0 = base_t + incl_t - ntri;
    if (start_next + slot + inner > max_nodes || tri0 + ntri > max_tris || start + i >= max_nodes) { atomicOr(&lv->failed, 2u); return; }
    n.child_base = start_next + slot; n.tri_base = tri0;
    for (int s = 0; s < 8; ++s) {
        if (kid_ref[s] >= 0) out[slot++] = kid_ref[s];
        else if (leaf_code[s]) {
            const uint32_t code = (uint32_t) ~leaf_code[s], first = code >> 4, count = (code & 15u) + 1u;
            const uint32_t at = tri0 + ((n.meta[s >> 2] >> (8 * (s & 3))) & 31u);
            for (uint32_t j = 0; j < count; ++j) perm[at + j] = first + j;
        }
    }
    nodes8[start + i] = n;
}

__global__ void k_bvh8_gather(const Tri *tris_in, const float *vn_in, const uint32_t *perm, uint32_t n, Tri *tris_out, float *vn_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = perm[i];
    tris_out[i] = tris_in[src];
    if (vn_in) for (int k = 0; k < 9; ++k) vn_out[(size_t) i * 9 + k] = vn_in[(size_t) src * 9 + k];
}

} // namespace miw
