// Device PLOC build (mi_bvh_build quality = 0 since round 4): the per-element steps of ploc_build.h, one thread per element.
// Counterpart in the reference: its GPU mode builds and compacts the acceleration structure on the device
// (include/mitsuba/render/optix/shapes.h:72-167); the CPU variants build the kd-tree (src/librender/scene_native.inl:3-10).
//
//   k_lbvh_bounds / k_lbvh_morton (lbvh_device.h) + rocPRIM radix sort (via hipCUB)   Morton order of the triangles
//   k_lbvh_leaves                      triangles / vertex normals gathered into Morton order, padded leaf boxes
//   k_ploc_init                        clusters = the leaves, counts 1, no parents
//   per round (no read-back; the cluster count lives on the device, a round is four launches):
//     k_ploc_partner                   every cluster's cheapest partner within `radius` positions       (ploc_partner)
//     k_ploc_flags                     survives / creates a node, packed for ONE prefix sum            (ploc_flags)
//     hipcub::DeviceScan::ExclusiveSum
//     k_ploc_apply                     mutual pairs become nodes, survivors are compacted in order      (ploc_apply)
//   k_ploc_offsets                     every node's first triangle in leaf order                        (ploc_offset)
//   k_ploc_scatter                     triangles / vertex normals into leaf order
//   k_ploc_emit                        inner nodes -> BvhNode records (subtrees of <= max_leaf triangles become leaves)
// The host reads the cluster count back every few rounds to know when to stop (rounds after the last merge retire at once).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "ploc_build.h"
#include "lbvh_device.h"

namespace miw {

static_assert(sizeof(PlocBox) == sizeof(LbvhBox), "k_lbvh_leaves writes the leaf boxes");

struct PlocState { uint32_t m, created; };           // clusters alive, nodes created so far

__global__ void k_ploc_init(uint32_t n, uint32_t *cluster, uint32_t *count, int32_t *parent, PlocState *state) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cluster[i] = (n - 1u) + i;
    if (i < 2u * n - 1u) { count[i] = 1u; parent[i] = -1; }
    if (i == 0) { state[0].m = n; state[0].created = 0u; state[1].m = n; state[1].created = 0u; }
}

__global__ void k_ploc_partner(const PlocState *state, const uint32_t *cluster, const PlocBox *box, uint32_t radius, uint32_t *partner) {
    const uint32_t m = state->m, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < 2u || i >= m) return;
    partner[i] = ploc_partner(cluster, box, m, i, radius);
}

// (covers [0, bound): positions past the live clusters get 0, so that the scan over `bound` items sees zeros there)
__global__ void k_ploc_flags(const PlocState *state, const uint32_t *partner, uint32_t bound, unsigned long long *flags) {
    const uint32_t m = state->m, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound) return;
    flags[i] = (m >= 2u && i < m) ? ploc_flags(partner, i) : 0ull;
}

__global__ void k_ploc_apply(const PlocState *state, PlocState *state_next, const uint32_t *cluster, const uint32_t *partner,
                             const unsigned long long *scan, uint32_t n, uint32_t max_leaf, PlocTree T, uint32_t *cluster_next) {
    const uint32_t m = state->m, created = state->created, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < 2u) { if (i == 0) *state_next = *state; return; }
    if (i >= m) return;
    ploc_apply(cluster, partner, scan, i, created, n, max_leaf, T, cluster_next);
    if (i == m - 1u) {                                          // the round's totals: the last position's prefix + its own flags
        const unsigned long long tot = scan[i] + ploc_flags(partner, i);
        state_next->m = (uint32_t) tot; state_next->created = created + (uint32_t) (tot >> 32);
    }
}

__global__ void k_ploc_offsets(PlocTree T, uint32_t n, uint32_t *offset) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < 2u * n - 1u) offset[v] = ploc_offset(T, v);
}

__global__ void k_ploc_scatter(const Tri *tris_sorted, const float *vn_sorted, const uint32_t *offset, uint32_t n, Tri *tris_out, float *vn_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t at = offset[(n - 1u) + i];
    tris_out[at] = tris_sorted[i];
    if (vn_sorted) for (int k = 0; k < 9; ++k) vn_out[(size_t) at * 9 + k] = vn_sorted[(size_t) i * 9 + k];
}

__global__ void k_ploc_emit(PlocTree T, const uint32_t *offset, uint32_t n, uint32_t max_leaf, BvhNode *nodes) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n - 1u) nodes[id] = ploc_emit(T, offset, id, n, max_leaf);
}

} // namespace miw
