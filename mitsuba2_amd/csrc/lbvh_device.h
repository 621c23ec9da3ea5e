// Device LBVH build (mi_bvh_build quality = 0): Morton codes -> radix sort -> Karras' parallel
// radix tree -> bottom-up box fitting, all on the GPU. Stands where ShapeKDTree::build() stands in
// the reference (src/librender/scene_native.inl:3-10); like the host SAH builder it only has to
// deliver a tree whose traversal result equals brute force (bvh.h), so none of the kd-tree's
// structure is reproduced. Output format = the SAH builder's: 64-byte BVH2 nodes holding both child
// boxes + parent links, root = node 0, up to 2 triangles per leaf by default (fat leaves, below), triangles permuted into leaf order.
// The 4-wide tree of the phase machine is collapsed from it on the device as well (bvh4_device.h).
//
// Kernels (hand-written; the key sort itself is rocPRIM's device radix sort via hipCUB — a plain
// library primitive):
//   k_lbvh_bounds   scene bounds by wave + workgroup reduction, one atomic min/max per workgroup
//   k_lbvh_morton   30-bit Morton code of each triangle centroid | triangle index (64-bit unique keys)
//   k_lbvh_leaves   gather triangles / vertex normals into sorted order, padded leaf boxes
//   k_lbvh_tree     Karras 2012: one thread per inner node finds its key range and split
//   k_lbvh_fit      bottom-up: every leaf climbs; the second thread to reach a node (agent-scope
//                   acq_rel counter) merges the child boxes and height and continues
//   k_lbvh_emit     inner nodes -> BvhNode records
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "miw/scene.h"
#include "miw/bvh.h"

namespace miw {

struct LbvhBox { float lo[3], hi[3]; };

__device__ __forceinline__ uint32_t lbvh_expand_bits(uint32_t v) {   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// order-preserving float <-> uint (for atomicMin/Max on floats of either sign)
__device__ __forceinline__ uint32_t lbvh_f2o(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lbvh_o2f(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ void k_lbvh_bounds(const Tri *tris, uint32_t n, uint32_t *bounds /* lo xyz, hi xyz as ordered uints */) {
    float lo[3] = { MIW_INFINITY, MIW_INFINITY, MIW_INFINITY }, hi[3] = { -MIW_INFINITY, -MIW_INFINITY, -MIW_INFINITY };
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Tri &t = tris[i];
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], fminf(t.p0[a], fminf(t.p1[a], t.p2[a])));
            hi[a] = fmaxf(hi[a], fmaxf(t.p0[a], fmaxf(t.p1[a], t.p2[a])));
        }
    }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64));
        }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) { atomicMin(bounds + a, lbvh_f2o(lo[a])); atomicMax(bounds + 3 + a, lbvh_f2o(hi[a])); }
}

__global__ void k_lbvh_morton(const Tri *tris, uint32_t n, const uint32_t *bounds, uint64_t *keys) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Tri &t = tris[i];
    uint32_t code = 0;
    for (int a = 0; a < 3; ++a) {
        float lo = lbvh_o2f(bounds[a]), hi = lbvh_o2f(bounds[3 + a]);
        float c = (fminf(t.p0[a], fminf(t.p1[a], t.p2[a])) + fmaxf(t.p0[a], fmaxf(t.p1[a], t.p2[a]))) * 0.5f;
        float ext = hi - lo;
        float u = ext > 0.f ? (c - lo) / ext : 0.f;
        uint32_t q = (uint32_t) fminf(fmaxf(u * 1024.f, 0.f), 1023.f);
        code |= lbvh_expand_bits(q) << (2 - a);
    }
    keys[i] = ((uint64_t) code << 32) | i;
}

__global__ void k_lbvh_leaves(const Tri *tris_in, const float *vn_in, const uint64_t *keys, uint32_t n, float pad,
                              Tri *tris_out, float *vn_out, LbvhBox *boxes /* [n-1 inner][n leaves] */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = (uint32_t) keys[i];
    Tri t = tris_in[src];
    tris_out[i] = t;
    if (vn_in) for (int k = 0; k < 9; ++k) vn_out[(size_t) i * 9 + k] = vn_in[(size_t) src * 9 + k];
    LbvhBox b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(t.p0[a], fminf(t.p1[a], t.p2[a])) - pad;
        b.hi[a] = fmaxf(t.p0[a], fmaxf(t.p1[a], t.p2[a])) + pad;
    }
    boxes[(n - 1) + i] = b;
}

// length of the common prefix of keys i and j (keys are unique), -1 outside the array
__device__ __forceinline__ int lbvh_delta(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long) (keys[i] ^ keys[j]));
}

struct LbvhLinks { int32_t left, right, parent; };   // child >= 0: inner node, < 0: ~leaf index

// Fat leaves: an inner node of the radix tree covers a contiguous run of the sorted triangles, so a subtree of at most
// `max_leaf` triangles can stand in the emitted BVH2 as ONE leaf (first = start of the run, count = its length): a tree of
// single-triangle leaves has twice the nodes and one more level. Measured on the 0.9 M-triangle interior (Msamples/s; host
// SAH 369): 1 triangle per leaf 311, 2 -> 352, 4 -> 334, 8 -> 302; the material balls, whose few huge wall triangles Morton
// order serves badly whatever the leaf size, stay 25 % behind their SAH tree (676 / 660 / 638 / 616 against 900). span[i] / first[i] = length and start of node i's run; the nodes inside a fat leaf stay in the arrays, unreferenced.
__global__ void k_lbvh_tree(const uint64_t *keys, int n, LbvhLinks *inner, int32_t *leaf_parent, uint32_t *span, uint32_t *first) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    // direction of the range, then its other end by doubling + binary search (Karras 2012, fig. 4)
    const int d = lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    // split position
    const int dnode = lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) {
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    span[i] = (uint32_t) (hi - lo + 1); first[i] = (uint32_t) lo;
    const int32_t left = lo == gamma ? ~gamma : gamma, right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    inner[i].left = left; inner[i].right = right;
    if (i == 0) inner[0].parent = -1;
    if (left >= 0) inner[left].parent = i; else leaf_parent[~left] = i;
    if (right >= 0) inner[right].parent = i; else leaf_parent[~right] = i;
}

__global__ void k_lbvh_fit(const LbvhLinks *inner, const int32_t *leaf_parent, int n, LbvhBox *boxes, uint32_t *arrivals,
                           uint32_t *height /* [n-1] */, const uint32_t *span, uint32_t max_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t node = leaf_parent[i];
    while (node >= 0) {
        // the first thread to arrive leaves; the second sees both children's results
        if (__hip_atomic_fetch_add(arrivals + node, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
        const int32_t l = inner[node].left, r = inner[node].right;
        const LbvhBox a = boxes[l >= 0 ? l : (n - 1) + ~l], b = boxes[r >= 0 ? r : (n - 1) + ~r];
        LbvhBox m;
        for (int k = 0; k < 3; ++k) { m.lo[k] = fminf(a.lo[k], b.lo[k]); m.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
        boxes[node] = m;
        const uint32_t hl = (l >= 0 && span[l] > max_leaf) ? height[l] : 0u, hr = (r >= 0 && span[r] > max_leaf) ? height[r] : 0u;   // fat leaves count as leaves
        height[node] = 1u + (hl > hr ? hl : hr);
        node = inner[node].parent;
    }
}

__global__ void k_lbvh_emit(const LbvhLinks *inner, const LbvhBox *boxes, int n, BvhNode *nodes, const uint32_t *span, const uint32_t *first, uint32_t max_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int32_t l = inner[i].left, r = inner[i].right;
    const LbvhBox a = boxes[l >= 0 ? l : (n - 1) + ~l], b = boxes[r >= 0 ? r : (n - 1) + ~r];
    BvhNode nd;
    for (int k = 0; k < 3; ++k) { nd.lo0[k] = a.lo[k]; nd.hi0[k] = a.hi[k]; nd.lo1[k] = b.lo[k]; nd.hi1[k] = b.hi[k]; }
    nd.child0 = l >= 0 ? (span[l] > max_leaf ? l : bvh_leaf_code(first[l], span[l])) : bvh_leaf_code((uint32_t) ~l, 1);
    nd.child1 = r >= 0 ? (span[r] > max_leaf ? r : bvh_leaf_code(first[r], span[r])) : bvh_leaf_code((uint32_t) ~r, 1);
    nd.parent = inner[i].parent; nd.pad = 0;
    nodes[i] = nd;
}

} // namespace miw
