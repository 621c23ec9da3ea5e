// Device-side collapse of a device-built BVH2 (lbvh_device.h) into the 4-wide quantised tree of miw/bvh4.h — so that a
// quality-0 mi_bvh_build never leaves the GPU (round 2 read the LBVH back and collapsed it on the host: 14 ms of device
// build followed by 272 ms of read-back + host work at 0.9 M triangles). Stands where the reference's GPU path builds and
// compacts its acceleration structure on the device (include/mitsuba/render/optix/shapes.h:72-228).
//
// The collapse is top-down (a node's fan-out depends on the stack budget its ancestors left over), so it runs level by
// level: k_bvh4_level turns the frontier of level L — items (BVH2 node, budget left) — into the Bvh4Nodes of that level and
// the frontier of level L + 1. One thread per item runs bvh4_collapse_node (bvh4_build.h), the very function the host
// builder runs, so a node is the same node whichever side built it; only the ORDER of the nodes inside a level is the
// device's own (slots are handed out per wavefront: one prefix sum over the lanes' inner-child counts + one atomic).
// Node indices: level L occupies [start[L], start[L] + count[L]), the children of item i of level L sit at
// start[L + 1] + slot. The launcher enqueues one launch per possible level (the LDS-stack budget bounds the height: <= 32)
// without reading anything back in between; a launch whose level is empty retires at once.
// Heights of the BVH2 subtrees (the fit test of the collapse) come from k_lbvh_fit, which already computes them.
#pragma once
#include <hip/hip_runtime.h>
#include "bvh4_build.h"

namespace miw {

struct Bvh4Item { int32_t ref; uint32_t budget; };
struct Bvh4Levels {                      // device bookkeeping, zeroed before the first level (count[0] = 1 is set by the launcher)
    uint32_t count[64], start[65];
    uint32_t stack_bound, failed;
};

__global__ __launch_bounds__(256) void k_bvh4_level(const BvhNode *n2, const uint32_t *h2, const Bvh4Item *in, Bvh4Item *out, Bvh4Levels *lv,
                                                    Bvh4Node *nodes4, uint32_t level, uint32_t stack_budget, int max_fan) {
    const uint32_t n_in = lv->count[level];
    if (blockIdx.x * blockDim.x >= n_in) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, start = lv->start[level], start_next = start + n_in;
    if (i == 0) lv->start[level + 1] = start_next;             // read by the next launch only
    const bool live = i < n_in;
    Bvh4Node n; int32_t kid_ref[4] = { -1, -1, -1, -1 };
    int k = 0; uint32_t budget = 0, inner = 0;
    if (live) {
        const Bvh4Item it = in[i];
        budget = it.budget;
        k = bvh4_collapse_node(n2, h2, it.ref, it.budget, max_fan, n, kid_ref);
        if (k < 0) { atomicOr(&lv->failed, 1u); k = 0; }
        for (int c = 0; c < k; ++c) inner += kid_ref[c] >= 0 ? 1u : 0u;
    }
    // slots of the next level: exclusive prefix sum of `inner` over the wavefront, one atomic per wavefront
    uint32_t incl = inner;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t) __shfl_up((int) incl, o, 64); if ((threadIdx.x & 63u) >= (uint32_t) o) incl += t; }
    const uint32_t total = (uint32_t) __shfl((int) incl, 63, 64);
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 63u && total) base = atomicAdd(&lv->count[level + 1], total);
    base = (uint32_t) __shfl((int) base, 63, 64);
    if (!live) return;
    uint32_t slot = base + incl - inner;
    for (int c = 0; c < k; ++c)
        if (kid_ref[c] >= 0) {
            n.child[c] = (int32_t) (start_next + slot);
            Bvh4Item o; o.ref = kid_ref[c]; o.budget = budget - (uint32_t) (k - 1);
            out[slot++] = o;
        }
    nodes4[start + i] = n;
    // pushes along the path down to and including this node = the walk's worst case if it ends here
    atomicMax(&lv->stack_bound, (stack_budget - budget) + (uint32_t) (k > 0 ? k - 1 : 0));
}

} // namespace miw
