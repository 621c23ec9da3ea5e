// N-GPU frames inside ONE process: device films and their reduce. Part of the single translation unit csrc/miwave.hip.
//
// pixel tiles shard over the GPUs of a node, every context renders its blocks into a private full-size
// film on its own device (mi_render, film_on_device = 1), and one sum onto the root's film closes the frame. The reference has
// one process and one call (include/mitsuba/render/integrator.h:42, Integrator::render(scene, sensor)); this is what lets the
// C++ host layer keep that contract over several GPUs without torch.distributed (mitsuba2_amd/dist.py stays the one-process-per-
// GPU alternative).
//   * all contexts on DISTINCT devices: RCCL (librccl, dlopen'ed on first use so that a single-GPU user never loads it) —
//     communicators from ncclCommInitAll, cached per device list; one grouped ncclReduce(sum, float32) over xGMI, in place on the root;
//   * otherwise (several contexts on one device: tests, or RCCL missing): the partial films are added onto the root's in rank
//     order by a device kernel (a film on another device goes through a staging buffer on the root's device first).
// Float32 association: a texel under one block only has ONE non-zero partial (exact); texels under a block border sum <= 4 block
// partials, whose association the reduce does not promise (RCCL's ring order; rank order in the fallback) — the same <= 1 ulp
// per border texel as the reference's thread-timing-dependent Film::put order (src/samplers/independent.cpp:36-40).
#pragma once
#include <dlfcn.h>
#include <atomic>
#include <map>
#include <mutex>
#include <rccl/rccl.h>

namespace miw {

__global__ void k_film_add(float *dst, const float *src, uint64_t n) {
    const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
}

struct RcclApi {
    void *lib = nullptr; bool tried = false;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(bool allowed = true) {
        if (!allowed) return false;                                                  // option MIW_RCCL=0 of the calling context: the device-add path (A/B runs)
        if (tried) return lib != nullptr;
        tried = true;
        for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll)) dlsym(lib, "ncclCommInitAll");
        Reduce = (decltype(Reduce)) dlsym(lib, "ncclReduce");
        GroupStart = (decltype(GroupStart)) dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd)) dlsym(lib, "ncclGroupEnd");
        CommDestroy = (decltype(CommDestroy)) dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString)) dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !Reduce || !GroupStart || !GroupEnd) { dlclose(lib); lib = nullptr; }
        return lib != nullptr;
    }
};
static RcclApi g_rccl;
static std::mutex g_rccl_mutex;
static std::map<std::vector<int>, std::vector<ncclComm_t>> g_rccl_comms;    // one communicator set per device list, kept while a context lives
static std::atomic<int> g_live_contexts{ 0 };
// mi_destroy of the LAST context releases the communicators (while HIP is certainly still up: a static destructor at process exit
// could run after the runtime's own teardown). A later mi_film_reduce builds them again.
static void rccl_release_comms() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.CommDestroy)
        for (auto &kv : g_rccl_comms) for (ncclComm_t cm : kv.second) (void) g_rccl.CommDestroy(cm);
    g_rccl_comms.clear();
}

} // namespace miw
