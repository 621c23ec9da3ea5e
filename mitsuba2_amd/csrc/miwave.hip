// libmiwave.so — gfx950 kernels and the C ABI declared in include/miwave.h. One translation unit:
//   miw/*.h                      leaf arithmetic shared with the CPU checker (float32, fixed operation order)
//   device/trace.h               LDS staging, packet sweep + leaf filter, LDS-stack BVH walk, trace2
//   device/wavefront_kernels.h   plan 1: k_init_lanes, k_trace<closest|any>, k_shade over SoA queues in HBM
//   device/resident_kernel.h     plan 2: k_init_pixels, the pixel queue, k_path_resident (path / direct)
//   device/phased_kernel.h       plan 2 over a tree: k_path_phased (wave-level phase machine: node steps / triangle tests / shade)
//   device/stream_trace.h        plan 1 over a tree: k_trace_stream (persistent walk kernel, dynamic ray fetch), k_sort_hits
//   device/film_kernels.h        k_film_resolve, k_film_groups (16-byte class records), k_film_blocks (24-byte position log), k_film_merge
//   device/eval_kernels.h        k_trace_soa (mi_trace), k_eval (mi_eval)
//   lbvh_device.h, bvh4_device.h device LBVH builder, level-synchronous collapse into the 4-wide tree
// followed here by the host side: context, scene upload, BVH build, mi_trace, mi_render, mi_eval, mi_selftest.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <string>
#include <vector>
#include <chrono>
#include <atomic>
#include <cmath>
#include <map>
#include <string>
#include <algorithm>

#include "../../include/miwave.h"
// Section clock (debug builds, -DMIW_SECTION_PROFILE=1): wall cycles per wavefront between markers, summed
// into g_sections and printed by mi_render when MIW_DEBUG is set. Not compiled into the product library.
#if defined(MIW_SECTION_PROFILE)
__device__ unsigned long long g_sections[16];
#endif
#if defined(MIW_PHASE_STATS)
__device__ unsigned long long g_phase_stats[15];   // k_path_phased: per bucket (node trip, triangle trip, walk end, shade, vote) runs, lanes, wall cycles
#endif
#if defined(MIW_WALK_STATS)
__device__ unsigned long long g_walk_stats[8];     // per ray kind (closest 0.., any 4..): node lane-steps, triangle lane-steps, rays
__device__ float g_walk_statsf[8];                 //                                      node wave-steps, triangle wave-steps
#endif
#if defined(MIW_VERIFY_FILTER)
__device__ unsigned int g_verify_n;
__device__ float g_verify[16 * 16];
#endif
#if defined(MIW_SECTION_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned long long *miw_sec_buf() { __shared__ unsigned long long b[4][16]; return &b[threadIdx.x >> 6][0]; }
#define MIW_SECTION(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); unsigned long long *b_ = miw_sec_buf(); \
        if ((threadIdx.x & 63u) == (unsigned) __ffsll((long long) __ballot(1)) - 1u) { b_[i] += now_ - b_[15]; b_[15] = now_; } } while (0)
#endif
#include "miw/base.h"
#include "miw/rng.h"
#include "miw/warp.h"
#include "miw/special.h"
#include "miw/shape.h"
#include "miw/bsdf.h"
#include "miw/scene.h"
#include "miw/film.h"
#include "miw/bvh.h"
#include "miw/path.h"
#include "miw/direct.h"
#include "rect_build.h"
#include "miw/film_gather.h"
#include "texture_build.h"
#include "bvh_build.h"
#include "bvh4_build.h"
#include "envmap_build.h"
#include "film_classes.h"
#include "lbvh_device.h"
#include "sah_device.h"
#include "bvh4_device.h"
#include "bvh8_device.h"
#include "film_reduce.h"

using namespace miw;

static_assert(sizeof(BvhNode) == 64, "BvhNode must be one 64-byte line");
static_assert(sizeof(Tri) == 48, "Tri must be 48 bytes");
static_assert(sizeof(BsdfRec) == 128 && sizeof(mi_bsdf) == 128, "bsdf record layout");
static_assert(sizeof(TexRec) == sizeof(mi_texture), "texture record layout");

#define MIW_BLOCK 256
#ifndef MIW_POOL_NW
#define MIW_POOL_NW 12              /* wavefronts per workgroup of k_path_pooled (device/pooled_kernel.h): one workgroup per CU, three wavefronts per SIMD */
#endif
#define MIW_CNT_SHARDS 1024        /* power of two */
#define MIW_BRUTE_MAX_LEAF 2        /* triangles per leaf box of the resident plan's candidate filter */
#define MIW_BRUTE_MAX_TRIS 64       /* <= this many triangles: LDS brute-force sweep instead of the BVH */
#define MIW_PLACE_PIECES 4          /* placed pixel queues: 64-lane pieces per SIMD queue (= wavefronts per SIMD of the packet kernel) */

#include "device/trace.h"
#include "device/wavefront_kernels.h"
#include "device/resident_kernel.h"
#include "device/phased_kernel.h"
#if !MIW_SPECTRAL
#include "device/pooled_kernel.h"      /* (experimental, opt-in: the scalar_rgb library only) */
#endif
#include "device/stream_trace.h"
#include "device/film_kernels.h"
#include "device/eval_kernels.h"

__global__ void k_iota(uint32_t *out, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = i; }

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

template <typename T> struct DevBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t resize(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void) hipFree(p);
        p = nullptr; n = 0;
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void **) &p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const std::vector<T> &v, hipStream_t s) {
        hipError_t e = resize(std::max<size_t>(v.size(), 1));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; n = 0; }
};

// a DevBuf local to one call: freed on every return path (HIP_TRY returns early)
template <typename T> struct TmpBuf : DevBuf<T> { TmpBuf() = default; TmpBuf(const TmpBuf &) = delete; ~TmpBuf() { this->release(); } };

// ---- options --------------------------------------------------------------------------------
// Every switch of the library (A/B runs, tests, fallbacks). The ENVIRONMENT is read ONCE, by mi_create, into the context's own copy;
// after that the caller's environment does not matter to this context: a switch is changed with mi_set_option(ctx, name, value)
// (value NULL: back to unset). INTEGRATION.md section 5 lists them; mi_render_cfg::debug_* carry the ones a caller may want per render.
struct KnobInfo { const char *name, *what; };
static const KnobInfo g_knobs[] = {
    { "MIW_DEBUG", "1: diagnostics on stderr (LDS layout, builder statistics, phase statistics of -DMIW_PHASE_STATS builds)" },
    { "MIW_DEBUG_ALLOC", "1: mi_bvh_build times its uploads and allocations" },
    { "MIW_DEVICE_BUILDER", "mi_bvh_build quality 0: sah (default) | lbvh: which device builder runs" },
    { "MIW_SAH_HUGE", "0: the device SAH sweep bins every candidate in one workgroup (round 4's form)" },
    { "MIW_LBVH_LEAF", "radix tree: triangles per leaf" },
    { "MIW_MAX_LEAF", "host SAH builder: triangles per leaf" },
    { "MIW_NO_STACK", "1: no LDS-stack walk (stackless BVH2 walk, wavefront plan)" },
    { "MIW_BVH4", "0: no 4-wide tree (the phase machine steps through the BVH2)" },
    { "MIW_BVH4_FAN", "2 | 3 | 4: widest node the 4-wide collapse may build" },
    { "MIW_BVH4_HOST", "1: the 4-wide tree is collapsed on the host from the read-back BVH2" },
    { "MIW_BVH8", "0: mi_bvh_build skips the 8-wide tree / mi_render keeps the 4-wide walk" },
    { "MIW_STACK8_FULL", "1: the 8-wide walk keeps the 16-entry LDS column instead of one entry per tree level" },
    { "MIW_ENV_TOP", "0: the environment warp's top levels are not staged in LDS" },
    { "MIW_PHASED", "0: tree scenes take the lock-step kernel instead of the phase machine" },
    { "MIW_PHASED_WAVES", "3 | 4: wavefronts per SIMD the phase machine is launched for" },
    { "MIW_TRIO", "0: MATS_PLAIN kernels where the MATS_TRIO class would do" },
    { "MIW_SHADE_VOTE", "num:den of the phase machine's shade vote" },
    { "MIW_LOOP_EXIT", "node:triangle hand-over ratios of the phase machine's walk loops" },
    { "MIW_POOLED", "1: tree scenes take k_path_pooled (walk jobs pooled across the workgroup; experimental, slower: DESIGN.md)" },
    { "MIW_POOL_SHAPE", "12x1 | 8x2: wavefronts per workgroup x pixels per lane of k_path_pooled" },
    { "MIW_POOL_VOTE", "shade_min:walk_min:node_min:tri_min:claim_min of k_path_pooled" },
    { "MIW_TAIL_PRIO", "0: no least-progress-first wave priorities" },
    { "MIW_WG_PER_CU", "workgroups per CU of the packet kernels' persistent grid" },
    { "MIW_PLACE", "0: shards of about one pixel per lane skip the measuring launch + placed queues" },
    { "MIW_JOB_CHUNK", "smallest chunk of a full frame's pixel jobs, in samples (a power of two; default 64, the phase machine 32): the pixels' sample streams are cut into halving chunks drawn chunk-major from one queue; 0: a job = all the samples of a pixel" },
    { "MIW_JOB_CHUNK_FORCE", "1: chunk jobs whatever the frame's size (tests of the hand-over between lanes)" },
    { "MIW_PACKET_SHARD4", "0: a plain-diffuse packet job of at most four wavefronts of pixels per SIMD keeps the five-wavefront (96-register) kernel" },
    { "MIW_PLACE_MEASURE", "divisor: the measuring launch runs spp / divisor samples" },
    { "MIW_PLACE_SPREAD", "0 | 1: placed pieces = consecutive sorted lanes | one lane of every cost stratum" },
    { "MIW_STREAM", "0: plan 1 walks with one kernel per list slice instead of the persistent stream kernel" },
    { "MIW_FILM_LEGACY", "1: the 24-byte position log + k_film_blocks for every filter" },
    { "MIW_FILM_LANES", "0 | 1 | 2: k_film_lanes off / over the tile-interleaved log / over [lane][sample]" },
    { "MIW_FILM_QUADS", "0 | 24 | 28 | 42 | 44: k_film_quads group shape (0: off)" },
    { "MIW_FILM_COLUMNS", "0 | 42 | 44 | 82: k_film_columns group shape" },
    { "MIW_FILM_GROUP", "2 | 3 | 4: k_film_groups group shape" },
    { "MIW_FL_NT", "0: k_film_lanes reads the log with plain instead of streaming loads" },
    { "MIW_FILM_OVERLAP", "1: part of the film replay of a full Cornell-class frame is queued on three more streams beside the path kernel, sets of 64-tile groups each behind the flags their last pixels raise (default 0: measured +0 - 1 %); 2: without the waits (timing probe: the film is the previous frame's)" },
    { "MIW_FILM_OVERLAP_SET", "groups of 64 tiles per launch of that replay (default 4)" },
    { "MIW_FILM_OVERLAP_ROOM", "workgroups the path kernel's grid is short of filling the device while the replay runs beside it (default 32: one per CU on 32 CUs)" },
    { "MIW_FILM_OVERLAP_LEAN", "0: the launches of that replay are the 168-register kernel (default 1: the 128-register form)" },
    { "MIW_FILM_OVERLAP_BESIDE", "groups of 64 tiles replayed beside the path kernel; the rest in one launch after it (default: three quarters)" },
    { "MIW_FILM_OVERLAP_PRIO", "s_setprio of the replay wavefronts that run beside the path kernel, 0 .. 3 (default 0)" },
    { "MIW_FQ_U", "2 | 4 | 8: records per trip of k_film_quads" },
    { "MIW_RCCL", "0: mi_film_reduce never uses RCCL (device add)" },
    { "MIW_RCCL_FORCE", "1: mi_film_reduce takes the RCCL branch for one context too (tests)" }
};
struct Options {
    std::map<std::string, std::string> v;
    void from_env() { for (const KnobInfo &k : g_knobs) if (const char *e = getenv(k.name)) v[k.name] = e; }
    static bool known(const char *name) { for (const KnobInfo &k : g_knobs) if (!strcmp(k.name, name)) return true; return false; }
    const char *get(const char *name) const { auto it = v.find(name); return it == v.end() ? nullptr : it->second.c_str(); }
};

struct mi_ctx {
    int device = 0;
    Options opt;
    hipStream_t stream = nullptr;
    std::string error;
    std::atomic<int> cancel{0};

    // host copy of the scene (input order)
    std::vector<Tri> tris_in;
    std::vector<float> tri_vn_in;       // 9 per tri or empty
    std::vector<float> tri_uv_in;       // 6 per face (texture coordinates, texture_build.h) or empty
    std::vector<ShapeRec> shapes;
    std::vector<AnalyticRec> rects;                 // analytic rectangles
    std::vector<BsdfRec> bsdfs; bool diffuse_only = false;   // every record one-sided smooth diffuse
    bool trio = false;                                       // every record diffuse / dielectric / roughconductor: MATS_TRIO kernels
    bool textured = false;                                   // texture coordinates, bitmaps or an "extended" plugin: MATS_ALL kernels
    std::vector<float> bsdf_tables; DevBuf<float> d_bsdf_tables;
    std::vector<EmitterRec> emitters;
    std::vector<float> emit_tri, emit_vnorm, emit_pmf, emit_cdf;
    bool have_scene = false, have_bvh = false;

    // device scene
    DevBuf<BvhNode> d_nodes; DevBuf<Tri> d_tris; DevBuf<float> d_tri_vn, d_tri_uv, d_bitmap_data; DevBuf<BitmapRec> d_bitmaps; uint32_t bitmap_count = 0;
    DevBuf<ShapeRec> d_shapes; DevBuf<BsdfRec> d_bsdfs; DevBuf<EmitterRec> d_emitters; DevBuf<AnalyticRec> d_rects;
    DevBuf<float> d_emit_tri, d_emit_vnorm, d_emit_pmf, d_emit_cdf;
    DevBuf<LeafBox> d_leaf_boxes; DevBuf<TriBounds> d_tri_bounds; DevBuf<Bvh4Node> d_nodes4; uint32_t nodes4_count = 0, nodes4_stack = 0;
    DevBuf<Bvh8Node> d_nodes8; DevBuf<Tri> d_tris8; DevBuf<float> d_tri_vn8; uint32_t nodes8_count = 0, nodes8_depth = 0;   // the 8-wide tree (miw/bvh8.h) and the triangles / vertex normals in its order
    DevBuf<float> d_env_data, d_env_levels; DevBuf<EnvmapRec> d_env; EnvmapRec env_host{}; size_t env_levels_total = 0;   // (host copy of the record: level offsets for the LDS staging)
    bool have_env = false;
    SceneView view{};
    TraceLds lds_cfg{}; size_t lds_bytes = 0;

    // render state
    DevBuf<F4> q_tp, q_res, q_ray_o, q_ray_d, q_hit, q_sh_d, q_sh_c;
    DevBuf<U4> q_st; DevBuf<F2> q_pos; DevBuf<uint32_t> q_pixel, q_sh_vis;
    DevBuf<double> d_accum; DevBuf<unsigned char> d_out;
    DevBuf<F2> q_log_pos; DevBuf<F4> q_log_val;         // 24-byte sample log (filters without phase classes)
    DevBuf<U4> q_log_rec;                               // 16-byte sample log (miw/film.h: phase classes)
    FilmClasses classes; FilmRec classes_of{}; bool classes_valid = false;   // the class tables of the last filter rendered with (ok or refused), and that filter
    DevBuf<float> d_fc_thr, d_fc_w;
    DevBuf<uint32_t> d_next_pixel; int cu_count = 256;
    DevBuf<uint32_t> d_lane_cost, d_cost_sorted, d_lane_iota, d_lane_sorted, d_piece_list, d_simd_ids;   // placed pixel queues of small shards (resident_kernel.h: QueueWork)
    DevBuf<unsigned char> d_place_tmp;
    DevBuf<uint32_t> d_lists, d_list_counts;    // wavefront plan: 2 parities x WL_LISTS lists / counters
    DevBuf<uint32_t> d_block_ids, d_tile_list; DevBuf<int32_t> d_block_tile; DevBuf<float> d_tiles;
    DevBuf<Counters> d_cnt;
    Counters *h_cnt = nullptr;          // pinned
    std::vector<uint32_t> h_piece_list, h_simd_ids;   // the placed launch's dealing (host side): uploaded without waiting, so they live here
    // film replay beside the render (round 6): a second stream, fork / join events, per group of 64 tiles the pixels done / expected (device) and
    // the flag its last pixel raises (host-coherent: what hipStreamWaitValue32 polls)
    hipStream_t stream2[3] = { nullptr, nullptr, nullptr }; hipEvent_t ev_fork = nullptr, ev_join[3] = { nullptr, nullptr, nullptr };   // (kernels of ONE stream run one after the other: three replay streams)
    DevBuf<uint32_t> d_group_done, d_group_expected; uint32_t *h_group_flag = nullptr; uint32_t group_flag_cap = 0;
    std::vector<uint32_t> h_group_expected; std::vector<int32_t> h_block_tile;
    int overlap_state = -1;             // -1: not tried; 0: the runtime refused (hipStreamWaitValue32 / the allocations); 1: available
    bool replay_enqueued = false;       // this frame's k_film_lanes launches already sit in the replay streams (assemble_film only joins)
    uint32_t replay_launches = 0;

    std::vector<hipEvent_t> ev_pool;
    mi_counters counters{};
};

static mi_status fail(mi_ctx *c, mi_status code, const char *fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->error = buf;
    return code;
}
#define HIP_TRY(c, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(c, MI_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static thread_local std::string g_global_error;

extern "C" {

int32_t mi_spectrum_channels(void) { return MIW_SPEC_N; }

mi_status mi_device_count(int32_t *count) {
    if (!count) return MI_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; g_global_error = hipGetErrorString(e); return MI_ERR_DEVICE; }
    *count = n;
    return MI_OK;
}

mi_status mi_create(int32_t device, mi_ctx **out) {
    if (!out) return MI_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { g_global_error = "no HIP device visible"; return MI_ERR_DEVICE; }
    if (device < 0 || device >= n) { g_global_error = "device index out of range"; return MI_ERR_INVALID; }
    if (hipSetDevice(device) != hipSuccess) { g_global_error = "hipSetDevice failed"; return MI_ERR_DEVICE; }
    mi_ctx *c = new mi_ctx();
    c->device = device;
    c->opt.from_env();                                           // the only place the library reads the environment
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->cu_count = prop.multiProcessorCount; }
    if (hipHostMalloc((void **) &c->h_cnt, sizeof(Counters) * MIW_CNT_SHARDS) != hipSuccess) { delete c; g_global_error = "hipHostMalloc failed"; return MI_ERR_DEVICE; }
    *out = c;
    g_live_contexts.fetch_add(1);
    return MI_OK;
}

void mi_destroy(mi_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipDeviceSynchronize();
    c->d_nodes.release(); c->d_tris.release(); c->d_tri_vn.release(); c->d_tri_uv.release(); c->d_bitmap_data.release(); c->d_bitmaps.release(); c->d_bsdf_tables.release(); c->d_shapes.release(); c->d_rects.release(); c->d_bsdfs.release();
    c->d_emitters.release(); c->d_leaf_boxes.release(); c->d_tri_bounds.release(); c->d_nodes4.release(); c->d_nodes8.release(); c->d_tris8.release(); c->d_tri_vn8.release(); c->d_env_data.release(); c->d_env_levels.release(); c->d_env.release(); c->d_emit_tri.release(); c->d_emit_vnorm.release(); c->d_emit_pmf.release(); c->d_emit_cdf.release();
    c->q_tp.release(); c->q_res.release(); c->q_ray_o.release(); c->q_ray_d.release(); c->q_hit.release();
    c->q_sh_d.release(); c->q_sh_c.release(); c->q_st.release(); c->q_pos.release(); c->q_pixel.release(); c->q_sh_vis.release();
    c->d_accum.release(); c->d_out.release(); c->d_next_pixel.release(); c->d_lane_cost.release(); c->d_cost_sorted.release(); c->d_lane_iota.release(); c->d_lane_sorted.release(); c->d_place_tmp.release(); c->d_piece_list.release(); c->d_simd_ids.release(); c->d_lists.release(); c->d_list_counts.release(); c->d_block_ids.release(); c->d_tile_list.release(); c->d_cnt.release();
    c->q_log_pos.release(); c->q_log_val.release(); c->q_log_rec.release(); c->d_fc_thr.release(); c->d_fc_w.release(); c->d_block_tile.release(); c->d_tiles.release();
    for (hipEvent_t e : c->ev_pool) (void) hipEventDestroy(e);
    c->d_group_done.release(); c->d_group_expected.release();
    if (c->h_group_flag) (void) hipHostFree(c->h_group_flag);
    if (c->ev_fork) (void) hipEventDestroy(c->ev_fork);
    for (int i = 0; i < 3; ++i) { if (c->ev_join[i]) (void) hipEventDestroy(c->ev_join[i]); if (c->stream2[i]) (void) hipStreamDestroy(c->stream2[i]); }
    if (c->h_cnt) (void) hipHostFree(c->h_cnt);
    delete c;
    if (g_live_contexts.fetch_sub(1) == 1) rccl_release_comms();
}

mi_status mi_set_stream(mi_ctx *c, void *s) { if (!c) return MI_ERR_INVALID; c->stream = (hipStream_t) s; return MI_OK; }

const char *mi_last_error(mi_ctx *c) { return c ? c->error.c_str() : g_global_error.c_str(); }

mi_status mi_set_option(mi_ctx *c, const char *name, const char *value) {
    if (!c || !name) return MI_ERR_INVALID;
    if (!Options::known(name)) return fail(c, MI_ERR_INVALID, "mi_set_option: unknown option %s", name);
    if (value) c->opt.v[name] = value; else c->opt.v.erase(name);
    return MI_OK;
}
const char *mi_get_option(mi_ctx *c, const char *name) { return c && name ? c->opt.get(name) : nullptr; }
int32_t mi_option_count(void) { return (int32_t) (sizeof(g_knobs) / sizeof(g_knobs[0])); }
const char *mi_option_name(int32_t i) { return i >= 0 && i < mi_option_count() ? g_knobs[i].name : nullptr; }
const char *mi_option_help(int32_t i) { return i >= 0 && i < mi_option_count() ? g_knobs[i].what : nullptr; }

mi_status mi_cancel(mi_ctx *c) { if (!c) return MI_ERR_INVALID; c->cancel.store(1); return MI_OK; }

mi_status mi_get_counters(mi_ctx *c, mi_counters *out) {
    if (!c || !out) return MI_ERR_INVALID;
    *out = c->counters;
    return MI_OK;
}

// ---- N-GPU frames inside one process: device films and their reduce (film_reduce.h) ------------------------------
mi_status mi_film_alloc(mi_ctx *c, uint64_t count, void **film) {
    if (!c || !film) return MI_ERR_INVALID;
    *film = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    float *p = nullptr;
    HIP_TRY(c, hipMalloc((void **) &p, std::max<uint64_t>(count, 1) * sizeof(float)));
    hipError_t e = hipMemsetAsync(p, 0, count * sizeof(float), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void) hipFree(p); return fail(c, MI_ERR_DEVICE, "mi_film_alloc: %s", hipGetErrorString(e)); }
    *film = p;
    return MI_OK;
}
void mi_film_free(mi_ctx *c, void *film) {
    if (!c || !film) return;
    (void) hipSetDevice(c->device);
    (void) hipFree(film);
}
mi_status mi_film_download(mi_ctx *c, const void *film, float *host, uint64_t count) {
    if (!c || !film || !host) return MI_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(host, film, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MI_OK;
}
mi_status mi_film_reduce(mi_ctx *const *ctxs, void *const *films, int32_t n, uint64_t count, int32_t root, int32_t *how) {
    if (how) *how = MI_REDUCE_NONE;
    if (!ctxs || !films || n < 1 || root < 0 || root >= n) return MI_ERR_INVALID;
    for (int32_t i = 0; i < n; ++i) if (!ctxs[i] || !films[i]) return MI_ERR_INVALID;
    mi_ctx *r = ctxs[root];
    // everything the contexts still have in flight on their streams lands first
    for (int32_t i = 0; i < n; ++i) { HIP_TRY(r, hipSetDevice(ctxs[i]->device)); HIP_TRY(r, hipStreamSynchronize(ctxs[i]->stream)); }
    // (one context: nothing to add — unless MIW_RCCL_FORCE=1 asks for the RCCL branch anyway: a communicator of one rank, an in-place
    // reduce that leaves the film as it is. That is how the branch — dlopen, ncclCommInitAll, the grouped ncclReduce, the stream
    // waits — runs on a one-GPU box: tests/test_multi_gpu.py)
    const bool force_rccl = r->opt.get("MIW_RCCL_FORCE") && atoi(r->opt.get("MIW_RCCL_FORCE")) != 0;
    if (count == 0 || (n == 1 && !force_rccl)) return MI_OK;
    std::vector<int> devs(n);
    bool distinct = true;
    for (int32_t i = 0; i < n; ++i) { devs[i] = ctxs[i]->device; for (int32_t j = 0; j < i; ++j) distinct = distinct && devs[j] != devs[i]; }
    if (distinct) {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (g_rccl.load(!(r->opt.get("MIW_RCCL") && atoi(r->opt.get("MIW_RCCL")) == 0))) {
            auto it = g_rccl_comms.find(devs);
            if (it == g_rccl_comms.end()) {
                std::vector<ncclComm_t> comms(n);
                const ncclResult_t rc = g_rccl.CommInitAll(comms.data(), n, devs.data());
                if (rc != ncclSuccess) return fail(r, MI_ERR_DEVICE, "mi_film_reduce: ncclCommInitAll: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
                it = g_rccl_comms.emplace(devs, comms).first;
            }
            // (no return between GroupStart and GroupEnd: a failure is remembered, the group is always closed, then reported — ADVICE r05)
            ncclResult_t rc = g_rccl.GroupStart();
            hipError_t he = hipSuccess;
            for (int32_t i = 0; i < n && rc == ncclSuccess && he == hipSuccess; ++i) {
                he = hipSetDevice(devs[i]);
                if (he == hipSuccess) rc = g_rccl.Reduce(films[i], films[i], (size_t) count, ncclFloat, ncclSum, root, it->second[i], ctxs[i]->stream);
            }
            const ncclResult_t re = g_rccl.GroupEnd();
            if (rc == ncclSuccess) rc = re;
            if (he != hipSuccess) return fail(r, MI_ERR_DEVICE, "mi_film_reduce: hipSetDevice: %s", hipGetErrorString(he));
            if (rc != ncclSuccess) return fail(r, MI_ERR_DEVICE, "mi_film_reduce: ncclReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
            for (int32_t i = 0; i < n; ++i) { HIP_TRY(r, hipSetDevice(devs[i])); HIP_TRY(r, hipStreamSynchronize(ctxs[i]->stream)); }
            HIP_TRY(r, hipSetDevice(r->device));
            if (how) *how = MI_REDUCE_RCCL;
            return MI_OK;
        }
    }
    if (n == 1) return MI_OK;                                   // (MIW_RCCL_FORCE without RCCL)
    // rank-ordered device add onto the root's film (contexts sharing a device, or no RCCL)
    HIP_TRY(r, hipSetDevice(r->device));
    TmpBuf<float> stage;
    const dim3 blk(256), grd((unsigned) ((count + 255) / 256));
    for (int32_t i = 0; i < n; ++i) {
        if (i == root) continue;
        const float *src = (const float *) films[i];
        if (ctxs[i]->device != r->device) {
            HIP_TRY(r, stage.resize(count));
            HIP_TRY(r, hipMemcpyPeerAsync(stage.p, r->device, films[i], ctxs[i]->device, count * sizeof(float), r->stream));
            src = stage.p;
        }
        hipLaunchKernelGGL(k_film_add, grd, blk, 0, r->stream, (float *) films[root], src, count);
    }
    HIP_TRY(r, hipGetLastError());
    HIP_TRY(r, hipStreamSynchronize(r->stream));
    if (how) *how = MI_REDUCE_DEVICE_ADD;
    return MI_OK;
}

// ---- scene upload ---------------------------------------------------------------------
mi_status mi_scene_upload(mi_ctx *c, const mi_scene_desc *s) {
    if (!c || !s) return MI_ERR_INVALID;
    c->have_scene = c->have_bvh = false;
    if (s->face_count && (!s->vertex_positions || !s->faces)) return fail(c, MI_ERR_INVALID, "scene: null geometry pointers");
    if (!s->shapes || s->shape_count == 0) return fail(c, MI_ERR_INVALID, "scene: no shapes");
    if (!s->bsdfs || s->bsdf_count == 0) return fail(c, MI_ERR_INVALID, "scene: no bsdfs");
    if (s->face_count >= (1u << 27) - 16u) return fail(c, MI_ERR_INVALID, "scene: too many faces");   // leaf codes (first << 4 | count - 1) must stay clear of MIW_BVH4_ABSENT / MIW_WALK_DONE

    c->tris_in.assign(s->face_count, Tri{});
    std::vector<uint32_t> face_shape(s->face_count, 0xffffffffu);
    bool any_normals = false;
    c->shapes.resize(s->shape_count);
    for (uint32_t i = 0; i < s->shape_count; ++i) {
        const mi_shape &sh = s->shapes[i];
        if (sh.bsdf >= s->bsdf_count) return fail(c, MI_ERR_INVALID, "shape %u: bsdf index out of range", i);
        if (sh.emitter >= (int32_t) s->emitter_count) return fail(c, MI_ERR_INVALID, "shape %u: emitter index out of range", i);
        if (s->envmap && s->envmap->emitter_index > s->emitter_count) return fail(c, MI_ERR_INVALID, "envmap: emitter_index out of range");
        if ((uint64_t) sh.first_face + sh.face_count > s->face_count) return fail(c, MI_ERR_INVALID, "shape %u: face range out of bounds", i);
        if ((sh.flags & MI_SHAPE_HAS_NORMALS) && !s->vertex_normals) return fail(c, MI_ERR_INVALID, "shape %u: HAS_NORMALS without vertex_normals", i);
        for (uint32_t f = sh.first_face; f < sh.first_face + sh.face_count; ++f) {
            if (face_shape[f] != 0xffffffffu) return fail(c, MI_ERR_INVALID, "face %u belongs to two shapes", f);
            face_shape[f] = i;
        }
        // emitter ids index the combined list: the environment map sits at envmap->emitter_index
        int32_t emitter_id = sh.emitter;
        if (emitter_id >= 0 && s->envmap && (uint32_t) emitter_id >= s->envmap->emitter_index) emitter_id += 1;
        ShapeRec r; r.bsdf = sh.bsdf; r.emitter = emitter_id; r.flags = sh.flags & (MI_SHAPE_HAS_NORMALS | MI_SHAPE_HAS_TEXCOORDS); r.pad = 0;
        if (sh.flags & (MI_SHAPE_RECTANGLE | MI_SHAPE_SPHERE)) {
            if (sh.face_count != 1) return fail(c, MI_ERR_INVALID, "shape %u: an analytic shape is one primitive (face_count == 1)", i);
            if (sh.flags & (MI_SHAPE_HAS_NORMALS | MI_SHAPE_HAS_TEXCOORDS)) return fail(c, MI_ERR_INVALID, "shape %u: an analytic shape has no vertex normals / texture coordinates", i);
            if ((sh.flags & MI_SHAPE_RECTANGLE) && (sh.flags & MI_SHAPE_SPHERE)) return fail(c, MI_ERR_INVALID, "shape %u: rectangle and sphere", i);
        }
        c->shapes[i] = r;
        any_normals = any_normals || (r.flags & 1u);
    }
    // analytic rectangles: record + two bounding triangles (the second one appended behind the faces)
    c->rects.clear();
    if (s->rectangle_count && !s->rectangles) return fail(c, MI_ERR_INVALID, "scene: rectangle_count without rectangles");
    std::vector<int32_t> shape_rect(s->shape_count, -1);
    for (uint32_t k = 0; k < s->rectangle_count; ++k) {
        const mi_rectangle &q = s->rectangles[k];
        if (q.shape >= s->shape_count || !(s->shapes[q.shape].flags & MI_SHAPE_RECTANGLE) || shape_rect[q.shape] >= 0)
            return fail(c, MI_ERR_INVALID, "rectangle %u: shape %u is not a (single) MI_SHAPE_RECTANGLE shape", k, q.shape);
        shape_rect[q.shape] = (int32_t) k;
        c->rects.push_back(rect_record(q.to_world, q.to_object, q.shape, s->shapes[q.shape].first_face));
        if (!(c->rects.back().inv_area > 0.f) || !isfinite_(c->rects.back().inv_area))
            return fail(c, MI_ERR_INVALID, "rectangle %u: degenerate to_world", k);
    }
    if (s->sphere_count && !s->spheres) return fail(c, MI_ERR_INVALID, "scene: sphere_count without spheres");
    for (uint32_t k = 0; k < s->sphere_count; ++k) {          // spheres follow the rectangles in the analytic table
        const mi_sphere &q = s->spheres[k];
        if (q.shape >= s->shape_count || !(s->shapes[q.shape].flags & MI_SHAPE_SPHERE) || shape_rect[q.shape] >= 0)
            return fail(c, MI_ERR_INVALID, "sphere %u: shape %u is not a (single) MI_SHAPE_SPHERE shape", k, q.shape);
        if (!(q.radius > 0.f) || !isfinite_(q.radius)) return fail(c, MI_ERR_INVALID, "sphere %u: radius must be positive", k);
        shape_rect[q.shape] = (int32_t) c->rects.size();
        c->rects.push_back(sphere_record(q.center, q.radius, q.flip_normals != 0, q.to_world, q.to_object, q.shape, s->shapes[q.shape].first_face));
    }
    for (uint32_t i = 0; i < s->shape_count; ++i)
        if ((s->shapes[i].flags & (MI_SHAPE_RECTANGLE | MI_SHAPE_SPHERE)) && shape_rect[i] < 0)
            return fail(c, MI_ERR_INVALID, "shape %u: no mi_rectangle record", i);
    if (!build_face_texcoords(s, c->tri_uv_in)) return fail(c, MI_ERR_INVALID, "scene: MI_SHAPE_HAS_TEXCOORDS without vertex_texcoords (or a vertex index out of range)");
    c->tris_in.resize((size_t) s->face_count + c->rects.size());
    c->tri_vn_in.clear();
    if (any_normals) c->tri_vn_in.assign(c->tris_in.size() * 9, 0.f);
    for (uint32_t f = 0; f < s->face_count; ++f) {
        if (face_shape[f] == 0xffffffffu) return fail(c, MI_ERR_INVALID, "face %u belongs to no shape", f);
        Tri &t = c->tris_in[f];
        if (shape_rect[face_shape[f]] >= 0) {                  // the rectangle's primitive slot: its two bounding triangles
            Tri two[2];
            analytic_bounding_tris(c->rects[shape_rect[face_shape[f]]], (uint32_t) shape_rect[face_shape[f]], two);
            t = two[0]; c->tris_in[(size_t) s->face_count + shape_rect[face_shape[f]]] = two[1];
            continue;
        }
        for (int k = 0; k < 3; ++k) {
            uint32_t vi = s->faces[3 * f + k];
            if (vi >= s->vertex_count) return fail(c, MI_ERR_INVALID, "face %u: vertex index out of range", f);
            float *dst = k == 0 ? t.p0 : (k == 1 ? t.p1 : t.p2);
            memcpy(dst, s->vertex_positions + 3 * (size_t) vi, 12);
            if (any_normals && (c->shapes[face_shape[f]].flags & 1u))
                memcpy(&c->tri_vn_in[(size_t) f * 9 + 3 * k], s->vertex_normals + 3 * (size_t) vi, 12);
        }
        t.shape = face_shape[f]; t.prim = f; t.pad = 0;
    }
    c->bsdfs.resize(s->bsdf_count);
    for (uint32_t i = 0; i < s->bsdf_count; ++i) {
        const mi_bsdf &b = s->bsdfs[i];
        BsdfRec r; int slot = 0;
        if (s->bsdf_table_floats && !s->bsdf_tables) return fail(c, MI_ERR_INVALID, "scene: bsdf_table_floats without bsdf_tables");
        if (const char *why = bsdf_record_from_abi(b, s->bitmap_count, s->bsdf_table_floats, r, &slot))
            return slot < 0 ? fail(c, MI_ERR_INVALID, "bsdf %u: %s", i, why) : fail(c, MI_ERR_INVALID, "bsdf %u: texture %d: %s", i, slot, why);
        r.back = 0;
        if (b.flags & MI_BSDF_FLAG_TWOSIDED) {                 // twosided.cpp:62-92
            if (b.back >= s->bsdf_count) return fail(c, MI_ERR_INVALID, "bsdf %u: back-side record %u out of range", i, b.back);
            const uint32_t tr = BSDF_Transmission;
            BsdfRec probe; memset(&probe, 0, sizeof probe);
            probe.type = b.type; const uint32_t f0 = bsdf_flags(probe);
            probe.type = s->bsdfs[b.back].type; const uint32_t f1 = probe.type < BSDF_TYPE_COUNT ? bsdf_flags(probe) : 0u;
            if ((f0 | f1) & tr) return fail(c, MI_ERR_INVALID, "bsdf %u: only materials without a transmission component can be nested", i);
            r.back = b.back;
        }
        c->bsdfs[i] = r;
    }
    c->diffuse_only = true;
    for (const BsdfRec &r : c->bsdfs) if (r.type != BSDF_TYPE_DIFFUSE || (r.flags & BSDF_REC_TWOSIDED)) c->diffuse_only = false;
    if (!c->tri_uv_in.empty()) c->diffuse_only = false;          // texture coordinates steer the shading frame (mesh.cpp:492-511)
    for (const BsdfRec &r : c->bsdfs) if (bsdf_uses_bitmap(r)) c->diffuse_only = false;   // the plain-diffuse kernel reads constants only
    {   // bitmap textures: one device buffer of texels, records pointing into it
        std::vector<BitmapRec> recs; uint32_t bad = 0;
        if (const char *why = build_bitmap_table(s, recs, &bad)) return fail(c, MI_ERR_INVALID, "bitmap %u: %s", bad, why);
        std::vector<size_t> first(recs.size() + 1, 0);
        for (size_t i = 0; i < recs.size(); ++i) first[i + 1] = first[i] + (size_t) recs[i].width * recs[i].height * recs[i].channels;
        HIP_TRY(c, c->d_bitmap_data.resize(std::max<size_t>(first.back(), 1)));
        for (size_t i = 0; i < recs.size(); ++i) {
            HIP_TRY(c, hipMemcpyAsync(c->d_bitmap_data.p + first[i], recs[i].data, (first[i + 1] - first[i]) * sizeof(float), hipMemcpyHostToDevice, c->stream));
            recs[i].data = c->d_bitmap_data.p + first[i];
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));            // the caller's arrays may go away after this call
        HIP_TRY(c, c->d_bitmaps.upload(recs, c->stream));
        c->bitmap_count = (uint32_t) recs.size();
    }
    c->trio = true;
    for (const BsdfRec &r : c->bsdfs) if (r.type != BSDF_TYPE_DIFFUSE && r.type != BSDF_TYPE_DIELECTRIC && r.type != BSDF_TYPE_ROUGHCONDUCTOR) c->trio = false;
    c->textured = !c->tri_uv_in.empty();
    for (const BsdfRec &r : c->bsdfs) if (bsdf_uses_bitmap(r) || bsdf_is_extended(r)) c->textured = true;   // -> the MATS_ALL kernels
    c->bsdf_tables.assign(s->bsdf_tables, s->bsdf_tables + s->bsdf_table_floats);
    HIP_TRY(c, c->d_bsdf_tables.upload(c->bsdf_tables, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // emitters: Mesh::build_pmf (mesh.cpp:285-312) + DiscreteDistribution (distr_1d.h:55-87)
    c->emitters.clear(); c->emit_tri.clear(); c->emit_vnorm.clear(); c->emit_pmf.clear(); c->emit_cdf.clear();
    bool any_emit_normals = false;
    auto push_env = [&]() { EmitterRec r; memset(&r, 0, sizeof r); r.type = EMITTER_ENVMAP; r.shape = 0xffffffffu; c->emitters.push_back(r); };
    for (uint32_t i = 0; i < s->emitter_count; ++i) {
        if (s->envmap && s->envmap->emitter_index == i) push_env();
        const mi_emitter &e = s->emitters[i];
        if (e.shape >= s->shape_count) return fail(c, MI_ERR_INVALID, "emitter %u: shape index out of range", i);
        const mi_shape &sh = s->shapes[e.shape];
        if (sh.emitter != (int32_t) i) return fail(c, MI_ERR_INVALID, "emitter %u: shape %u does not point back to it", i, e.shape);
        if (sh.face_count == 0) return fail(c, MI_ERR_INVALID, "emitter %u: cannot create sampling table for an empty mesh", i);
        EmitterRec r; memset(&r, 0, sizeof r);
#if MIW_SPECTRAL
        if (e.radiance_tex.type != MI_TEX_D65 && e.radiance_tex.type != MI_TEX_SRGB_D65 && e.radiance_tex.type != MI_TEX_UNIFORM)
            return fail(c, MI_ERR_INVALID, "emitter %u: the scalar_spectral library needs a spectral radiance record", i);
        memcpy(&r.radiance, &e.radiance_tex, sizeof(TexRec));
#else
        r.radiance.type = TEX_RGB; memcpy(r.radiance.v, e.radiance, 12);
#endif
        if (shape_rect[e.shape] >= 0) {                        // area light on an analytic rectangle: no face tables
            const AnalyticRec &q = c->rects[shape_rect[e.shape]];
            r.shape = e.shape; r.tri_first = (uint32_t) shape_rect[e.shape]; r.tri_count = 0; r.flags = 2u;
            r.valid_lo = r.valid_hi = 0; r.normalization = q.inv_area; r.sum = rcp(q.inv_area);
            c->emitters.push_back(r);
            continue;
        }
        r.shape = e.shape; r.tri_first = (uint32_t) c->emit_pmf.size(); r.tri_count = sh.face_count;
        r.flags = (sh.flags & MI_SHAPE_HAS_NORMALS) ? 1u : 0u;
        any_emit_normals = any_emit_normals || r.flags;
        double sum = 0.0; uint32_t vlo = 0xffffffffu, vhi = 0;
        for (uint32_t k = 0; k < sh.face_count; ++k) {
            const Tri &t = c->tris_in[sh.first_face + k];
            float area = face_area(ld3(t.p0), ld3(t.p1), ld3(t.p2));
            if (area < 0.f) return fail(c, MI_ERR_INVALID, "emitter %u: negative face area", i);
            c->emit_pmf.push_back(area);
            sum += (double) area;
            c->emit_cdf.push_back((float) sum);
            if (area > 0.f) { if (vlo == 0xffffffffu) vlo = k; vhi = k; }
            for (int q = 0; q < 3; ++q) c->emit_tri.push_back(t.p0[q]);
            for (int q = 0; q < 3; ++q) c->emit_tri.push_back(t.p1[q]);
            for (int q = 0; q < 3; ++q) c->emit_tri.push_back(t.p2[q]);
            for (int q = 0; q < 9; ++q)
                c->emit_vnorm.push_back(r.flags ? c->tri_vn_in[(size_t) (sh.first_face + k) * 9 + q] : 0.f);
        }
        if (vlo == 0xffffffffu) return fail(c, MI_ERR_INVALID, "emitter %u: no probability mass found", i);
        r.valid_lo = vlo; r.valid_hi = vhi;
        r.sum = (float) sum; r.normalization = (float) (1.0 / sum);
        c->emitters.push_back(r);
    }
    if (!any_emit_normals) c->emit_vnorm.clear();
    if (s->envmap && s->envmap->emitter_index >= s->emitter_count) push_env();

    HIP_TRY(c, hipSetDevice(c->device));
    c->have_env = false;
#if MIW_SPECTRAL
    if (s->envmap && !s->envmap->density)
        return fail(c, MI_ERR_INVALID, "envmap: the scalar_spectral library needs mi_envmap::density (its texels are model coefficients, include/miwave.h)");
#endif
    if (s->envmap) {
        EnvmapTables t = envmap_build(*s->envmap);
        if (!t.ok) return fail(c, MI_ERR_INVALID, "envmap: needs >= 2x2 texels, a non-zero luminance sum and an invertible to_world");
        HIP_TRY(c, c->d_env_data.upload(t.data, c->stream));
        HIP_TRY(c, c->d_env_levels.upload(t.levels, c->stream));
        t.rec.data = c->d_env_data.p; t.rec.levels = c->d_env_levels.p;
        c->env_host = t.rec; c->env_levels_total = t.levels.size();
        std::vector<EnvmapRec> one(1, t.rec);
        HIP_TRY(c, c->d_env.upload(one, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->have_env = true;
    }
    HIP_TRY(c, c->d_shapes.upload(c->shapes, c->stream));
    HIP_TRY(c, c->d_tri_uv.upload(c->tri_uv_in, c->stream));
    HIP_TRY(c, c->d_rects.upload(c->rects, c->stream));
    HIP_TRY(c, c->d_bsdfs.upload(c->bsdfs, c->stream));
    HIP_TRY(c, c->d_emitters.upload(c->emitters, c->stream));
    HIP_TRY(c, c->d_emit_tri.upload(c->emit_tri, c->stream));
    HIP_TRY(c, c->d_emit_vnorm.upload(c->emit_vnorm, c->stream));
    HIP_TRY(c, c->d_emit_pmf.upload(c->emit_pmf, c->stream));
    HIP_TRY(c, c->d_emit_cdf.upload(c->emit_cdf, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->have_scene = true;
    return MI_OK;
}

// ---- BVH build ------------------------------------------------------------------------
// dwords of the tables stage_tables() copies into LDS (shapes, bsdfs, emitters, emit_tri, emit_vnorm, emit_pmf, emit_cdf), each
// padded to 16 bytes; returns the bytes of the block
#define MIW_LDS_PER_WORKGROUP (40u * 1024u)       /* 160 KB per CU / four workgroups of 256 (four wavefronts per SIMD) */
#define MIW_LDS_STATIC 1024u                      /* static __shared__ of k_path_phased: s_job_pend (QueueWork: a word per lane); s_prog, 16 bytes, inside the granule below */
#define MIW_LDS_GRANULE 512u                      /* LDS is handed out in granules: dynamic + static, rounded up, must stay inside 40 KB */
static size_t lds_table_bytes(const mi_ctx *c, uint32_t words[10], bool with_tris) {
    words[0] = (uint32_t) (c->shapes.size() * sizeof(ShapeRec) / 4); words[1] = (uint32_t) (c->bsdfs.size() * sizeof(BsdfRec) / 4);
    words[2] = (uint32_t) (c->emitters.size() * sizeof(EmitterRec) / 4);
    words[3] = (uint32_t) c->emit_tri.size(); words[4] = (uint32_t) c->emit_vnorm.size(); words[5] = (uint32_t) c->emit_pmf.size(); words[6] = (uint32_t) c->emit_cdf.size();
    // packet scenes: the triangle records (leaf order), their vertex normals and per-face texture coordinates as well
    words[7] = with_tris ? (uint32_t) (c->tris_in.size() * sizeof(Tri) / 4) : 0u;
    words[8] = with_tris ? (uint32_t) c->tri_vn_in.size() : 0u; words[9] = with_tris ? (uint32_t) c->tri_uv_in.size() : 0u;
    size_t total = 0;
    for (int k = 0; k < 10; ++k) total += ((size_t) words[k] + 3u) / 4u * 16u;
    return total;
}

mi_status mi_bvh_build(mi_ctx *c, int32_t quality) {
    if (!c) return MI_ERR_INVALID;
    if (!c->have_scene) return fail(c, MI_ERR_STATE, "mi_bvh_build: no scene uploaded");
    const bool force_tree = (quality & MI_BVH_FORCE_TREE) != 0;
    const int32_t quality_flags = quality;
    quality &= ~(MI_BVH_FORCE_TREE | MI_BVH_NO_LEAF_FILTER | MI_BVH_RADIX_TREE);
    if (quality != 1 && quality != 0) return fail(c, MI_ERR_INVALID, "mi_bvh_build: quality must be 0 or 1");
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(c, hipSetDevice(c->device));
    const float pad_unit = scene_pad_unit(c->tris_in);        // 1e-5 x the largest |coordinate| (bvh_build.h): boxes grow by two units, the accept rule by one
    BvhBuildResult r;                       // host SAH result (nodes kept for the tiny-scene leaf filter)
    uint32_t node_count = 0, tri_count = (uint32_t) c->tris_in.size(), depth = 0;
    bool built_on_device = false;
    // quality 0: the 4-wide tree is collapsed on the device as well (bvh4_device.h); these say whether that happened
    bool wide_on_device = false; uint32_t dev4_nodes = 0, dev4_stack = 0;
    const bool wide_on = !(c->opt.get("MIW_BVH4") && atoi(c->opt.get("MIW_BVH4")) == 0);
    // the 8-wide tree (the phase machine's default since round 5): off with MIW_BVH8=0, and whenever the 4-wide walk is asked for by name
    const bool wide8_on = wide_on && !(c->opt.get("MIW_BVH8") && atoi(c->opt.get("MIW_BVH8")) == 0) && !c->opt.get("MIW_BVH4_FAN");
    uint32_t dev8_nodes = 0, dev8_depth = 0; double ms_bvh8 = 0.0;
    std::vector<uint32_t> sah_level_start;                    // the device SAH builder's level table (first node of every level)
    int max_fan = 4;
    if (const char *e = c->opt.get("MIW_BVH4_FAN")) max_fan = std::min(4, std::max(2, atoi(e)));
    double ms_bvh4 = 0.0;
    uint32_t tab_words_[10];
    // (the packet kernels read the scene's small tables from LDS: a scene of <= 64 triangles whose tables would not fit beside the
    // packets — dozens of unused BSDF records — is walked as a tree instead)
    const bool tiny = c->tris_in.size() <= MIW_BRUTE_MAX_TRIS && !force_tree && c->rects.empty() &&   // packets are triangles only
                      (!MIW_LDS_TABLES || lds_table_bytes(c, tab_words_, true) <= 24u * 1024u);
    if (quality == 0 && !tiny && tri_count >= 2) {
        // ---- device LBVH (lbvh_device.h) ----
        hipStream_t s = c->stream;
        const int n = (int) tri_count;
        TmpBuf<Tri> d_in; TmpBuf<float> d_vn_in; TmpBuf<uint64_t> d_keys, d_keys_sorted; TmpBuf<uint32_t> d_bounds, d_arrivals, d_height, d_span, d_first;
        uint32_t lbvh_leaf = 2u;                               // triangles per fat leaf (measured on the interior: 1 -> 311, 2 -> 352, 4 -> 334, 8 -> 302 Msamples/s; SAH 369); MIW_LBVH_LEAF = 1 .. 16 overrides
        if (const char *e = c->opt.get("MIW_LBVH_LEAF")) lbvh_leaf = (uint32_t) std::min(16, std::max(1, atoi(e)));
        if ((uint32_t) n <= lbvh_leaf) lbvh_leaf = 1u;          // (the root must stay an inner node)
        TmpBuf<LbvhBox> d_boxes; TmpBuf<LbvhLinks> d_inner; TmpBuf<int32_t> d_leaf_parent; TmpBuf<unsigned char> d_tmp;
        auto free_tmp = [&]() { d_in.release(); d_vn_in.release(); d_keys.release(); d_keys_sorted.release(); d_bounds.release();
                                d_arrivals.release(); d_height.release(); d_boxes.release(); d_inner.release(); d_leaf_parent.release(); d_tmp.release();
                                d_span.release(); d_first.release(); };
        // MIW_DEBUG_ALLOC=1: where the set-up time of a build goes (stderr, ms since the previous mark)
        auto lap_t = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!c->opt.get("MIW_DEBUG_ALLOC")) return;
            (void) hipStreamSynchronize(s);
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[miwave]   build set-up: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
            lap_t = now;
        };
        HIP_TRY(c, d_in.upload(c->tris_in, s));
        lap("triangle upload");
        if (!c->tri_vn_in.empty()) HIP_TRY(c, d_vn_in.upload(c->tri_vn_in, s));
        lap("vertex-normal upload");
        HIP_TRY(c, c->d_nodes.resize(n)); HIP_TRY(c, c->d_tris.resize(n)); HIP_TRY(c, d_height.resize(n));
        if (!c->tri_vn_in.empty()) HIP_TRY(c, c->d_tri_vn.resize((size_t) n * 9));
        lap("node / triangle buffers");
        const dim3 blk(256), grd((unsigned) ((n + 255) / 256));
        // Which device builder. Default: the level-by-level binned-SAH sweep (sah_device.h) — the host builder's tree, node for node.
        // MI_BVH_RADIX_TREE / MIW_DEVICE_BUILDER=lbvh: the radix tree over Morton codes of rounds 2 - 3 (lbvh_device.h; measured 7 % / 16 % behind the SAH
        // tree on the interior / the material balls: A/B runs). A scene the sweep hands back (need_host: coincident centroids) takes
        // the host builder, like quality 1.
        enum { DEV_SAH = 0, DEV_LBVH = 1 } dev_builder = (quality_flags & MI_BVH_RADIX_TREE) ? DEV_LBVH : DEV_SAH;
        if (const char *e = c->opt.get("MIW_DEVICE_BUILDER")) {        // only the two names override the flag; anything else leaves the caller's choice alone
            if (!strcmp(e, "lbvh")) dev_builder = DEV_LBVH; else if (!strcmp(e, "sah")) dev_builder = DEV_SAH;
        }
        bool sah_need_host = false;
        if (dev_builder == DEV_SAH) {
            uint32_t max_leaf = 4u;                                        // (bvh_build_sah's default leaf size: the same tree; MIW_MAX_LEAF overrides on both sides)
            if (const char *e = c->opt.get("MIW_MAX_LEAF")) max_leaf = (uint32_t) std::min(16, std::max(1, atoi(e)));
            const uint32_t un = (uint32_t) n;
            const float pad = 2.f * pad_unit;
            TmpBuf<SahPrim> d_prim; TmpBuf<uint32_t> d_ia, d_ib, d_flags, d_rank; TmpBuf<SahCand> d_ca, d_cb; TmpBuf<SahDecision> d_dec; TmpBuf<SahState> d_state;
            TmpBuf<unsigned char> d_scan_tmp; TmpBuf<SahHuge> d_huge;
            HIP_TRY(c, d_huge.resize(MIW_SAH_HUGE_CANDS));
            HIP_TRY(c, d_prim.resize(un)); HIP_TRY(c, d_ia.resize(un)); HIP_TRY(c, d_ib.resize(un)); HIP_TRY(c, d_flags.resize(un)); HIP_TRY(c, d_rank.resize(un));
            HIP_TRY(c, d_ca.resize(un)); HIP_TRY(c, d_cb.resize(un)); HIP_TRY(c, d_dec.resize(un)); HIP_TRY(c, d_state.resize(1));
            size_t scan_bytes = 0;
            HIP_TRY(c, rocprim::exclusive_scan(nullptr, scan_bytes, d_flags.p, d_rank.p, 0u, (size_t) n, rocprim::plus<uint32_t>(), s));
            HIP_TRY(c, d_scan_tmp.resize(scan_bytes + 16));
            lap("builder temporaries");
            HIP_TRY(c, hipStreamSynchronize(s));
            const auto t_setup = std::chrono::steady_clock::now();     // (uploads and allocations behind us)
            hipLaunchKernelGGL(k_sah_prims, grd, blk, 0, s, d_in.p, un, pad, d_prim.p, d_ia.p);
            const SahCand root = { 0u, un, -1, 0u };
            HIP_TRY(c, hipMemcpyAsync(d_ca.p, &root, sizeof root, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemsetAsync(d_state.p, 0, sizeof(SahState), s));
            std::vector<uint32_t> level_start{ 0u };
            uint32_t n_cand = 1u, base = 0u, level = 0u;
            bool big_left = true;                                        // candidates of more than MIW_SAH_BIG triangles in the current level
            uint32_t level_max = un;                                     // the largest candidate of the current level (0: none above MIW_SAH_BIG)
            const bool huge_on = !(c->opt.get("MIW_SAH_HUGE") && atoi(c->opt.get("MIW_SAH_HUGE")) == 0);     // MIW_SAH_HUGE=0: one workgroup per candidate, as round 4 (A/B runs)
            while (n_cand > 0u && !sah_need_host) {
                const SahCand *cur = (level & 1u) ? d_cb.p : d_ca.p; SahCand *nxt = (level & 1u) ? d_ca.p : d_cb.p;
                const uint32_t *ic = (level & 1u) ? d_ib.p : d_ia.p; uint32_t *in = (level & 1u) ? d_ia.p : d_ib.p;
                // few candidates: 1024 threads each. Many: one wavefront each — except those above MIW_SAH_BIG triangles (an unbalanced
                // split leaves a 100 k-triangle candidate next to thousands of small ones twelve levels down; one wavefront on it cost
                // 15 ms), which a second launch of the 1024-thread kernel takes; `big_left` (read back with the level's totals) says
                // whether the level holds any.
                const bool all_big = n_cand <= 512u;
                const uint32_t none = 0u, all = 0xffffffffu;
                // candidates above MIW_SAH_HUGE triangles: sliced over several workgroups (sah_device.h: k_sah_huge_*), the others as before
                const bool huge = huge_on && level_max > MIW_SAH_HUGE && n_cand <= MIW_SAH_HUGE_CANDS;
                if (huge) {
                    const dim3 hgrid(n_cand, (level_max + MIW_SAH_SLICE - 1u) / MIW_SAH_SLICE);
                    const uint32_t words = n_cand * (uint32_t) (sizeof(SahHuge) / 4u);
                    hipLaunchKernelGGL(k_sah_huge_init, dim3((words + 255u) / 256u), dim3(256), 0, s, d_huge.p, n_cand);
                    hipLaunchKernelGGL(k_sah_huge_box, hgrid, dim3(1024), 0, s, cur, n_cand, ic, d_prim.p, d_huge.p);
                    hipLaunchKernelGGL(k_sah_huge_bins, hgrid, dim3(1024), 0, s, cur, n_cand, ic, d_prim.p, level, d_huge.p);
                    hipLaunchKernelGGL(k_sah_huge_finish, dim3(n_cand), dim3(64), 0, s, cur, n_cand, level, max_leaf, d_huge.p, d_dec.p, d_flags.p, d_state.p);
                }
                if (all_big || big_left) hipLaunchKernelGGL(k_sah_decide<1024>, dim3(n_cand), dim3(1024), 0, s, cur, n_cand, ic, d_prim.p, level, max_leaf, d_dec.p, d_flags.p, d_state.p, all_big ? none : MIW_SAH_BIG, huge ? MIW_SAH_HUGE : all);
                if (!all_big) hipLaunchKernelGGL(k_sah_decide<64>, dim3(n_cand), dim3(64), 0, s, cur, n_cand, ic, d_prim.p, level, max_leaf, d_dec.p, d_flags.p, d_state.p, none, big_left ? MIW_SAH_BIG : all);
                HIP_TRY(c, rocprim::exclusive_scan(d_scan_tmp.p, scan_bytes, d_flags.p, d_rank.p, 0u, (size_t) n_cand, rocprim::plus<uint32_t>(), s));
                hipLaunchKernelGGL(k_sah_totals, dim3(1), dim3(1), 0, s, d_flags.p, d_rank.p, n_cand, d_state.p);
                if (all_big || big_left) hipLaunchKernelGGL(k_sah_apply<1024>, dim3(n_cand), dim3(1024), 0, s, cur, n_cand, ic, in, d_prim.p, d_dec.p, d_rank.p, base, c->d_nodes.p, nxt, d_state.p, all_big ? none : MIW_SAH_BIG, all, un);
                if (!all_big) hipLaunchKernelGGL(k_sah_apply<64>, dim3(n_cand), dim3(64), 0, s, cur, n_cand, ic, in, d_prim.p, d_dec.p, d_rank.p, base, c->d_nodes.p, nxt, d_state.p, none, big_left ? MIW_SAH_BIG : all, un);
                SahState h;
                HIP_TRY(c, hipMemcpyAsync(&h, d_state.p, sizeof h, hipMemcpyDeviceToHost, s));
                HIP_TRY(c, hipStreamSynchronize(s));
                HIP_TRY(c, hipGetLastError());
                sah_need_host = h.need_host != 0u || level >= 62u || (uint64_t) base + h.n_inner > (uint64_t) un - 1u;
                big_left = h.max_count > MIW_SAH_BIG; level_max = h.max_count;
                if (big_left) HIP_TRY(c, hipMemsetAsync(&d_state.p->max_count, 0, sizeof(uint32_t), s));     // (counts the next level afresh)
                depth = level;
                base += h.n_inner; level_start.push_back(base);
                n_cand = 2u * h.n_inner; ++level;
            }
            const auto t_levels = std::chrono::steady_clock::now();
            if (!sah_need_host) {
                for (size_t L = level_start.size() - 1; L-- > 0;) {
                    const uint32_t cnt = level_start[L + 1] - level_start[L];
                    if (cnt) hipLaunchKernelGGL(k_sah_heights, dim3((cnt + 255u) / 256u), blk, 0, s, c->d_nodes.p, level_start[L], level_start[L + 1], d_height.p);
                }
                const uint32_t *idx_final = (level & 1u) ? d_ib.p : d_ia.p;
                hipLaunchKernelGGL(k_sah_gather, grd, blk, 0, s, d_in.p, c->tri_vn_in.empty() ? (const float *) nullptr : d_vn_in.p, idx_final, un,
                                   c->d_tris.p, c->tri_vn_in.empty() ? (float *) nullptr : c->d_tri_vn.p);
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipStreamSynchronize(s));
                node_count = base;
                built_on_device = true;
                sah_level_start = level_start;
                c->counters.bvh_builder = 3u;
                if (c->opt.get("MIW_DEBUG")) {
                    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                    fprintf(stderr, "[miwave] device builder: binned SAH by levels, %u triangles, %u inner nodes, depth %u; upload + allocations %.2f ms, %u levels %.2f ms, heights + gather %.2f ms\n",
                            un, node_count, depth, ms(t0, t_setup), level, ms(t_setup, t_levels), ms(t_levels, std::chrono::steady_clock::now()));
                }
            }
        } else {
            HIP_TRY(c, d_keys.resize(n)); HIP_TRY(c, d_keys_sorted.resize(n)); HIP_TRY(c, d_bounds.resize(6));
            HIP_TRY(c, d_arrivals.resize(n)); HIP_TRY(c, d_boxes.resize((size_t) 2 * n));
            HIP_TRY(c, d_inner.resize(n)); HIP_TRY(c, d_leaf_parent.resize(n)); HIP_TRY(c, d_span.resize(n)); HIP_TRY(c, d_first.resize(n));
            const uint32_t init_bounds[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
            HIP_TRY(c, hipMemcpyAsync(d_bounds.p, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemsetAsync(d_arrivals.p, 0, (size_t) n * sizeof(uint32_t), s));
            HIP_TRY(c, hipMemsetAsync(d_height.p, 0, (size_t) n * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_lbvh_bounds, dim3(std::min<unsigned>(grd.x, 1024u)), blk, 0, s, d_in.p, (uint32_t) n, d_bounds.p);
            hipLaunchKernelGGL(k_lbvh_morton, grd, blk, 0, s, d_in.p, (uint32_t) n, d_bounds.p, d_keys.p);
            size_t tmp_bytes = 0;
            HIP_TRY(c, rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.p, d_keys_sorted.p, (size_t) n, 0u, 64u, s));
            HIP_TRY(c, d_tmp.resize(tmp_bytes + 16));
            HIP_TRY(c, rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.p, d_keys_sorted.p, (size_t) n, 0u, 64u, s));
            // box padding (bvh.h, bvh_build.h: scene_pad_unit): 2e-5 x the largest |coordinate|
            uint32_t hb[6];
            HIP_TRY(c, hipMemcpyAsync(hb, d_bounds.p, sizeof hb, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            float m = 0.f;
            for (int k = 0; k < 6; ++k) { uint32_t o = hb[k]; float f = u2f((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); m = std::max(m, std::fabs(f)); }
            const float pad = 2.f * std::max(1e-5f * m, 1e-30f);
            // the radix tree over the Morton order (lbvh_device.h)
            HIP_TRY(c, hipMemsetAsync(d_height.p, 0, (size_t) n * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_lbvh_leaves, grd, blk, 0, s, d_in.p, c->tri_vn_in.empty() ? (const float *) nullptr : d_vn_in.p, d_keys_sorted.p,
                               (uint32_t) n, pad, c->d_tris.p, c->tri_vn_in.empty() ? (float *) nullptr : c->d_tri_vn.p, d_boxes.p);
            hipLaunchKernelGGL(k_lbvh_tree, grd, blk, 0, s, d_keys_sorted.p, n, d_inner.p, d_leaf_parent.p, d_span.p, d_first.p);
            hipLaunchKernelGGL(k_lbvh_fit, grd, blk, 0, s, d_inner.p, d_leaf_parent.p, n, d_boxes.p, d_arrivals.p, d_height.p, d_span.p, lbvh_leaf);
            hipLaunchKernelGGL(k_lbvh_emit, grd, blk, 0, s, d_inner.p, d_boxes.p, n, c->d_nodes.p, d_span.p, d_first.p, lbvh_leaf);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(&depth, d_height.p, sizeof depth, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            c->counters.bvh_builder = 1u;
            if (c->opt.get("MIW_DEBUG")) fprintf(stderr, "[miwave] device builder: LBVH, %u triangles, height %u\n", (uint32_t) n, depth);
        node_count = (uint32_t) (n - 1);
        built_on_device = depth <= MIW_BVH_MAX_DEPTH;     // deeper (many coincident centroids): take the SAH builder
        }
        // ---- the 4-wide tree of the phase machine, collapsed level by level on the device (bvh4_device.h); the heights the
        // collapse's fit test needs are the builder's (k_sah_heights / k_lbvh_fit). MIW_BVH4_HOST=1 keeps round 2's read-back + host collapse (A/B runs) ----
        const uint32_t budget4 = MIW_STACK_ENTRIES - 1;    // one entry of slack: the node body's unconditional stores
        if (built_on_device && wide_on && depth <= budget4 && !c->opt.get("MIW_NO_STACK") && !c->opt.get("MIW_BVH4_HOST")) {
            auto t4 = std::chrono::steady_clock::now();
            TmpBuf<Bvh4Item> fa, fb; TmpBuf<Bvh4Levels> lv;
            HIP_TRY(c, fa.resize(n)); HIP_TRY(c, fb.resize(n)); HIP_TRY(c, lv.resize(1)); HIP_TRY(c, c->d_nodes4.resize(n));
            Bvh4Levels h; memset(&h, 0, sizeof h); h.count[0] = 1;
            const Bvh4Item root = { 0, budget4 };
            HIP_TRY(c, hipMemcpyAsync(lv.p, &h, sizeof h, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(fa.p, &root, sizeof root, hipMemcpyHostToDevice, s));
            const uint32_t levels = std::min<uint32_t>(depth + 1u, 62u);
            for (uint32_t L = 0; L < levels; ++L)
                hipLaunchKernelGGL(k_bvh4_level, grd, blk, 0, s, c->d_nodes.p, d_height.p, (L & 1u) ? fb.p : fa.p, (L & 1u) ? fa.p : fb.p, lv.p,
                                   c->d_nodes4.p, L, budget4, max_fan);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(&h, lv.p, sizeof h, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            for (uint32_t L = 0; L < levels; ++L) dev4_nodes += h.count[L];
            dev4_stack = h.stack_bound;
            wide_on_device = !h.failed && h.count[levels] == 0 && dev4_stack <= budget4 && dev4_nodes <= (uint32_t) n;
            ms_bvh4 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t4).count();
        }
        // ---- the 8-wide tree (miw/bvh8.h) from the same BVH2, on the device as well (bvh8_device.h): the programme bottom-up over the
        // builder's levels, the collapse top-down, triangles + vertex normals gathered into the tree's order. SAH sweep only (the
        // radix tree has no level table: it keeps the 4-wide tree). MIW_BVH8=0 / MIW_BVH4=0 / MIW_BVH4_FAN switch it off (A/B runs). ----
        if (built_on_device && dev_builder == DEV_SAH && wide8_on && !c->opt.get("MIW_NO_STACK")) {
            auto t8 = std::chrono::steady_clock::now();
            TmpBuf<Bvh8Dp> d_dp; TmpBuf<int32_t> ga, gb; TmpBuf<Bvh8Levels> lv8; TmpBuf<uint32_t> d_perm;
            HIP_TRY(c, d_dp.resize(node_count)); HIP_TRY(c, ga.resize(n)); HIP_TRY(c, gb.resize(n)); HIP_TRY(c, lv8.resize(1)); HIP_TRY(c, d_perm.resize(n));
            HIP_TRY(c, c->d_nodes8.resize(n)); HIP_TRY(c, c->d_tris8.resize(n));
            if (!c->tri_vn_in.empty()) HIP_TRY(c, c->d_tri_vn8.resize((size_t) n * 9));
            for (size_t L = sah_level_start.size() - 1; L-- > 0;) {
                const uint32_t cnt = sah_level_start[L + 1] - sah_level_start[L];
                if (cnt) hipLaunchKernelGGL(k_bvh8_dp, dim3((cnt + 255u) / 256u), blk, 0, s, c->d_nodes.p, d_dp.p, sah_level_start[L], sah_level_start[L + 1]);
            }
            Bvh8Levels h8; memset(&h8, 0, sizeof h8); h8.count[0] = 1;
            const int32_t root8 = 0;
            HIP_TRY(c, hipMemcpyAsync(lv8.p, &h8, sizeof h8, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(ga.p, &root8, sizeof root8, hipMemcpyHostToDevice, s));
            const uint32_t levels8 = std::min<uint32_t>(depth + 1u, 62u);
            for (uint32_t L = 0; L < levels8; ++L)
                hipLaunchKernelGGL(k_bvh8_level, grd, blk, 0, s, c->d_nodes.p, d_dp.p, (L & 1u) ? gb.p : ga.p, (L & 1u) ? ga.p : gb.p, lv8.p,
                                   c->d_nodes8.p, d_perm.p, L, (uint32_t) n, (uint32_t) n);
            hipLaunchKernelGGL(k_bvh8_gather, grd, blk, 0, s, c->d_tris.p, c->tri_vn_in.empty() ? (const float *) nullptr : c->d_tri_vn.p, d_perm.p, (uint32_t) n,
                               c->d_tris8.p, c->tri_vn_in.empty() ? (float *) nullptr : c->d_tri_vn8.p);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipMemcpyAsync(&h8, lv8.p, sizeof h8, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            uint32_t nn = 0, dd = 0;
            for (uint32_t L = 0; L < levels8; ++L) { nn += h8.count[L]; dd += h8.count[L] ? 1u : 0u; }
            if (!h8.failed && h8.count[levels8] == 0 && h8.tri_next == (uint32_t) n && dd <= MIW_BVH8_STACK && nn <= (uint32_t) n) { dev8_nodes = nn; dev8_depth = dd; }
            ms_bvh8 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t8).count();
        }
        free_tmp();
    }
    std::vector<float> vn;
    if (!built_on_device) {
        // ---- host binned SAH (bvh_build.h) ----
        // tiny scenes are swept through their SAH leaves' boxes (trace2): leaf size tuned for that filter
        uint32_t max_leaf = tiny ? MIW_BRUTE_MAX_LEAF : 4u;
        if (const char *e = c->opt.get("MIW_MAX_LEAF")) max_leaf = (uint32_t) atoi(e);
        r = bvh_build_sah(c->tris_in, -1.f, max_leaf);
        if (r.depth > MIW_BVH_MAX_DEPTH) return fail(c, MI_ERR_INVALID, "BVH depth %u exceeds the traversal trail", r.depth);
        if (!c->tri_vn_in.empty()) {
            vn.resize(c->tri_vn_in.size());
            for (size_t i = 0; i < r.order.size(); ++i)
                memcpy(&vn[i * 9], &c->tri_vn_in[(size_t) r.order[i] * 9], 36);
        }
        HIP_TRY(c, c->d_nodes.upload(r.nodes, c->stream));
        HIP_TRY(c, c->d_tris.upload(r.tris, c->stream));
        HIP_TRY(c, c->d_tri_vn.upload(vn, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        node_count = (uint32_t) r.nodes.size(); tri_count = (uint32_t) r.tris.size(); depth = r.depth;
    }
    c->counters.bvh_on_device = built_on_device ? 1u : 0u;
    if (!built_on_device) c->counters.bvh_builder = 0u;

    SceneView &v = c->view;
    v.accept_pad = pad_unit;                                  // shape.h: the bounds rule of every triangle hit
    v.tri_bounds = nullptr;
    v.nodes = c->d_nodes.p; v.node_count = node_count;
    v.tris = c->d_tris.p; v.tri_count = tri_count;
    v.tri_vn = c->tri_vn_in.empty() ? nullptr : c->d_tri_vn.p;
    v.tri_uv = c->tri_uv_in.empty() ? nullptr : c->d_tri_uv.p;
    v.bitmaps = c->bitmap_count ? c->d_bitmaps.p : nullptr;
    v.bsdf_tables = c->bsdf_tables.empty() ? nullptr : c->d_bsdf_tables.p;
    v.shapes = c->d_shapes.p; v.shape_count = (uint32_t) c->shapes.size();
    v.bsdfs = c->d_bsdfs.p; v.bsdf_count = (uint32_t) c->bsdfs.size();
    v.emitters = c->d_emitters.p; v.emitter_count = (uint32_t) c->emitters.size();
    scene_view_prepare(v);
    v.emit_tri = c->d_emit_tri.p; v.emit_vnorm = c->emit_vnorm.empty() ? nullptr : c->d_emit_vnorm.p;
    v.emit_pmf = c->d_emit_pmf.p; v.emit_cdf = c->d_emit_cdf.p;
    v.env = c->have_env ? c->d_env.p : nullptr;
    v.env_top = nullptr; v.env_top_count = v.env_top_base = 0;      // (set inside the kernels that stage them: trace.h stage_tables)
    v.rects = c->rects.empty() ? nullptr : c->d_rects.p; v.rect_count = (uint32_t) c->rects.size();

    // LDS plan: whole scene if it fits in 16 KiB (keeps 8 workgroups/CU resident),
    // otherwise the top of the tree only.
    size_t all = (size_t) node_count * sizeof(BvhNode) + (size_t) tri_count * sizeof(Tri);
    c->lds_cfg.brute = 0; c->lds_cfg.leaves = 0; c->lds_cfg.stack = 0; c->lds_cfg.stack16 = 0; v.leaf_boxes = nullptr; v.nodes4 = nullptr; c->nodes4_count = c->nodes4_stack = 0;
    v.nodes8 = nullptr; c->nodes8_count = c->nodes8_depth = 0; bool wide8_on_device = false;
    if (tiny && v.tri_count > 0) {
        // tiny scene (Cornell class): a branch-free sweep over LDS triangle packets beats any tree walk
        c->lds_cfg.brute = 1; c->lds_cfg.nodes_staged = 0; c->lds_cfg.tris_staged = v.tri_count;
        // the SAH leaves (padded boxes, <= 4 triangles each) become the resident plan's candidate filter
        std::vector<LeafBox> leaves;
        auto add_leaf = [&](const float *lo, const float *hi, int32_t child) {
            if (child >= 0 || !(lo[0] <= hi[0])) return;          // inner node, or absent child (inverted box)
            const uint32_t code = (uint32_t) ~child;
            const uint32_t first = code >> 4, count = (code & 15u) + 1u;
            const unsigned long long bits = (count >= 64u ? ~0ull : ((1ull << count) - 1ull)) << first;
            LeafBox b; for (int a = 0; a < 3; ++a) { b.p[2 * a] = lo[a]; b.p[2 * a + 1] = hi[a]; } b.mask_lo = (uint32_t) bits; b.mask_hi = (uint32_t) (bits >> 32);
            leaves.push_back(b);
        };
        for (const BvhNode &n : r.nodes) { add_leaf(n.lo0, n.hi0, n.child0); add_leaf(n.lo1, n.hi1, n.child1); }
        HIP_TRY(c, c->d_leaf_boxes.upload(leaves, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        v.leaf_boxes = c->d_leaf_boxes.p;
        c->lds_cfg.leaves = (quality_flags & MI_BVH_NO_LEAF_FILTER) ? 0u : (uint32_t) leaves.size();
        std::vector<TriBounds> bounds(v.tri_count);               // per packet (leaf order), staged behind the boxes
        for (uint32_t i = 0; i < v.tri_count; ++i)
            bounds[i] = tri_bounds(ld3(r.tris[i].p0), ld3(r.tris[i].p1), ld3(r.tris[i].p2), v.accept_pad);
        HIP_TRY(c, c->d_tri_bounds.upload(bounds, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        v.tri_bounds = c->d_tri_bounds.p;
        c->lds_bytes = v.tri_count * sizeof(TriPacket) + leaves.size() * MIW_LEAF_BOX_COPIES * sizeof(LeafBox) + v.tri_count * sizeof(TriBounds);   // (trace.h: stage_to_lds)
    } else {
        // A whole tree that fits 16 KiB could be walked out of LDS with the stackless trail walk; measured (r02 triangle-count
        // series: 172 triangles, 375 Msamples/s that way against 850 with the LDS-stack walk of the phase machine, whose
        // node fetches hit L1) that only pays for the forced-tree test path of <= 64 triangles, which keeps it covered.
        const bool stack_ok = depth <= MIW_STACK_ENTRIES && !c->opt.get("MIW_NO_STACK");
        const bool resident_tree = all <= 16 * 1024 && (!stack_ok || v.tri_count <= MIW_BRUTE_MAX_TRIS);
        if (resident_tree) { c->lds_cfg.nodes_staged = v.node_count; c->lds_cfg.tris_staged = v.tri_count; }
        else { c->lds_cfg.nodes_staged = MIW_LDS_TOP ? std::min<uint32_t>(v.node_count, 255) : 0u; c->lds_cfg.tris_staged = 0; }
        c->lds_bytes = c->lds_cfg.nodes_staged * sizeof(BvhNode) + c->lds_cfg.tris_staged * sizeof(Tri);
        c->lds_cfg.stack = 0; c->lds_cfg.stack16 = 0;
        if (!resident_tree && stack_ok) {
            c->lds_cfg.stack = 1; c->lds_cfg.stack16 = (uint32_t) (c->lds_bytes / 16);
            c->lds_bytes += (size_t) MIW_STACK_ENTRIES * MIW_BLOCK * sizeof(int32_t);
        }
        // The phase machine (plan 2 over a stack-walked tree) walks the 4-wide quantised collapse of this tree (miw/bvh4.h,
        // bvh4_build.h; +10 % on the material balls and the interior against the BVH2 walk). The BVH2 stays for
        // mi_trace, the scene queries and plan 1; a device-built LBVH is collapsed on the device (bvh4_device.h, above). MIW_BVH4=0 switches it
        // off (A/B runs), MIW_BVH4_FAN = 2..4 caps the fan-out. A tree the collapse refuses (height above the stack budget,
        // coordinates beyond the quantisation range) is rendered by the lock-step kernel.
        if (c->lds_cfg.stack && wide_on && wide_on_device) {
            v.nodes4 = c->d_nodes4.p; c->nodes4_count = dev4_nodes; c->nodes4_stack = dev4_stack;
            if (c->opt.get("MIW_DEBUG")) fprintf(stderr, "[miwave] bvh4 (device): %u nodes (bvh2 %u), stack bound %u, %.2f ms\n", dev4_nodes, node_count, dev4_stack, ms_bvh4);
        } else if (c->lds_cfg.stack && wide_on) {
            auto t4 = std::chrono::steady_clock::now();
            if (built_on_device) {
                r.nodes.resize(node_count);
                HIP_TRY(c, hipMemcpy(r.nodes.data(), c->d_nodes.p, (size_t) node_count * sizeof(BvhNode), hipMemcpyDeviceToHost));
            }
            const Bvh4BuildResult b4 = bvh4_collapse(r.nodes, MIW_STACK_ENTRIES - 1, max_fan);   // one entry of slack: the node body's unconditional stores
            if (b4.ok) {
                HIP_TRY(c, c->d_nodes4.upload(b4.nodes, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                v.nodes4 = c->d_nodes4.p; c->nodes4_count = (uint32_t) b4.nodes.size(); c->nodes4_stack = b4.stack_bound;
            }
            ms_bvh4 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t4).count();
            if (c->opt.get("MIW_DEBUG")) fprintf(stderr, "[miwave] bvh4: %zu nodes (bvh2 %u), depth %u, stack bound %u, ok %d\n", b4.nodes.size(), node_count, b4.depth, b4.stack_bound, (int) b4.ok);
        }
        // The 8-wide tree (miw/bvh8.h; walked instead of the 4-wide one whenever it exists — mi_render, MIW_BVH8=0 at render time keeps
        // the 4-wide walk): collapsed on the device above, or here on the host from the host-built BVH2 (quality 1). c->view keeps
        // the BVH2's triangle order; mi_render hands the phase machine a view whose tris / tri_vn are d_tris8 / d_tri_vn8.
        if (c->lds_cfg.stack && v.nodes4 && dev8_nodes) {
            c->nodes8_count = dev8_nodes; c->nodes8_depth = dev8_depth; wide8_on_device = true;
            if (c->opt.get("MIW_DEBUG")) fprintf(stderr, "[miwave] bvh8 (device): %u nodes (bvh2 %u, bvh4 %u), depth %u, %.2f ms\n", dev8_nodes, node_count, c->nodes4_count, dev8_depth, ms_bvh8);
        } else if (c->lds_cfg.stack && v.nodes4 && wide8_on && !built_on_device) {
            auto t8 = std::chrono::steady_clock::now();
            const Bvh8BuildResult b8 = bvh8_collapse(r.nodes, tri_count);
            if (b8.ok) {
                std::vector<Tri> t8v(b8.perm.size()); std::vector<float> vn8;
                for (size_t i = 0; i < t8v.size(); ++i) t8v[i] = r.tris[b8.perm[i]];
                if (!vn.empty()) { vn8.resize(vn.size()); for (size_t i = 0; i < t8v.size(); ++i) memcpy(&vn8[i * 9], &vn[(size_t) b8.perm[i] * 9], 36); }
                HIP_TRY(c, c->d_nodes8.upload(b8.nodes, c->stream)); HIP_TRY(c, c->d_tris8.upload(t8v, c->stream)); HIP_TRY(c, c->d_tri_vn8.upload(vn8, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                c->nodes8_count = (uint32_t) b8.nodes.size(); c->nodes8_depth = b8.depth;
            }
            ms_bvh8 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t8).count();
            if (c->opt.get("MIW_DEBUG")) fprintf(stderr, "[miwave] bvh8: %zu nodes (bvh2 %u), depth %u, ok %d, %.2f ms\n", b8.nodes.size(), node_count, b8.depth, (int) b8.ok, ms_bvh8);
        }
    }

    // the 8-wide arrays were sized for the worst case (one node per triangle) before the collapse ran: a tree that was refused or is
    // not adopted gives them back, an adopted one keeps the nodes it has (0.9 M triangles: 6.9 MB instead of 73 MB) — ADVICE r05
    if (c->nodes8_count == 0) { c->d_nodes8.release(); c->d_tris8.release(); c->d_tri_vn8.release(); }
    else if (c->d_nodes8.n > 2 * (size_t) c->nodes8_count) {
        DevBuf<Bvh8Node> fit;
        HIP_TRY(c, fit.resize(c->nodes8_count));
        HIP_TRY(c, hipMemcpyAsync(fit.p, c->d_nodes8.p, (size_t) c->nodes8_count * sizeof(Bvh8Node), hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->d_nodes8.release(); c->d_nodes8 = fit;
    }

    c->counters.ms_bvh_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->counters.bvh_nodes = v.node_count; c->counters.bvh_tris = v.tri_count; c->counters.bvh_depth = depth;
    c->counters.bvh4_on_device = (v.nodes4 && wide_on_device) ? 1u : 0u; c->counters.ms_bvh4 = ms_bvh4;
    c->counters.bvh8_nodes = c->nodes8_count; c->counters.bvh8_depth = c->nodes8_depth; c->counters.bvh8_on_device = wide8_on_device ? 1u : 0u; c->counters.ms_bvh8 = ms_bvh8;
    c->have_bvh = true;
    return MI_OK;
}

// ---- mi_trace --------------------------------------------------------------------------
mi_status mi_trace(mi_ctx *c, const mi_rays_soa *rays, const mi_hits_soa *hits, uint64_t n, int32_t any_hit) {
    if (!c || !rays || !hits) return MI_ERR_INVALID;
    if (!c->have_bvh) return fail(c, MI_ERR_STATE, "mi_trace: call mi_scene_upload and mi_bvh_build first");
    if (n == 0) return MI_OK;
    if (!hits->t) return fail(c, MI_ERR_INVALID, "mi_trace: hits->t is required");
    HIP_TRY(c, hipSetDevice(c->device));
    TmpBuf<float> in, out; TmpBuf<uint32_t> outu;
    HIP_TRY(c, in.resize(8 * n)); HIP_TRY(c, out.resize(3 * n)); HIP_TRY(c, outu.resize(2 * n));
    const float *src[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->mint, rays->maxt };
    for (int k = 0; k < 8; ++k) {
        if (!src[k]) return fail(c, MI_ERR_INVALID, "mi_trace: null ray array");
        HIP_TRY(c, hipMemcpyAsync(in.p + k * n, src[k], n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    SoaRays R = { in.p, in.p + n, in.p + 2 * n, in.p + 3 * n, in.p + 4 * n, in.p + 5 * n, in.p + 6 * n, in.p + 7 * n };
    SoaHits H = { out.p, out.p + n, out.p + 2 * n, outu.p, outu.p + n };
    dim3 grid((unsigned) ((n + MIW_BLOCK - 1) / MIW_BLOCK)), block(MIW_BLOCK);
    if (any_hit) hipLaunchKernelGGL(k_trace_soa<true>, grid, block, c->lds_bytes, c->stream, c->view, R, H, n, c->lds_cfg);
    else         hipLaunchKernelGGL(k_trace_soa<false>, grid, block, c->lds_bytes, c->stream, c->view, R, H, n, c->lds_cfg);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(hits->t, out.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (hits->u) HIP_TRY(c, hipMemcpyAsync(hits->u, out.p + n, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (hits->v) HIP_TRY(c, hipMemcpyAsync(hits->v, out.p + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (hits->prim) HIP_TRY(c, hipMemcpyAsync(hits->prim, outu.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    if (hits->shape) HIP_TRY(c, hipMemcpyAsync(hits->shape, outu.p + n, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MI_OK;
}

// ---- the Scene query surface: mi_ray_intersect, mi_sample_emitter_direction, mi_pdf_emitter_direction, mi_emitter_eval ----
// (host arrays in, host arrays out, like mi_trace; temporaries are RAII so that an early HIP_TRY return frees them)
mi_status mi_ray_intersect(mi_ctx *c, const mi_rays_soa *rays, mi_surface_interaction *si, uint64_t n) {
    if (!c || !rays || !si) return MI_ERR_INVALID;
    if (!c->have_bvh) return fail(c, MI_ERR_STATE, "mi_ray_intersect: call mi_scene_upload and mi_bvh_build first");
    if (n == 0) return MI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    TmpBuf<float> in; TmpBuf<mi_surface_interaction> out;
    HIP_TRY(c, in.resize(8 * n)); HIP_TRY(c, out.resize(n));
    const float *src[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->mint, rays->maxt };
    for (int k = 0; k < 8; ++k) {
        if (!src[k]) return fail(c, MI_ERR_INVALID, "mi_ray_intersect: null ray array");
        HIP_TRY(c, hipMemcpyAsync(in.p + k * n, src[k], n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    SoaRays R = { in.p, in.p + n, in.p + 2 * n, in.p + 3 * n, in.p + 4 * n, in.p + 5 * n, in.p + 6 * n, in.p + 7 * n };
    hipLaunchKernelGGL(k_ray_intersect, dim3((unsigned) ((n + MIW_BLOCK - 1) / MIW_BLOCK)), dim3(MIW_BLOCK), c->lds_bytes, c->stream, c->view, R, out.p, n, c->lds_cfg);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(si, out.p, n * sizeof(mi_surface_interaction), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MI_OK;
}

mi_status mi_sample_emitter_direction(mi_ctx *c, int32_t emitter, const float *ref_p, const float *sample, const float *wavelengths,
                                      int32_t test_visibility, mi_direction_sample *ds, float *spec_out, uint64_t n) {
    if (!c || !ref_p || !sample || !ds || !spec_out) return MI_ERR_INVALID;
    if (!c->have_bvh) return fail(c, MI_ERR_STATE, "mi_sample_emitter_direction: call mi_scene_upload and mi_bvh_build first");
    if (emitter >= (int32_t) c->emitters.size()) return fail(c, MI_ERR_INVALID, "mi_sample_emitter_direction: emitter index out of range");
    if (MIW_SPECTRAL && !wavelengths) return fail(c, MI_ERR_INVALID, "mi_sample_emitter_direction: the scalar_spectral library needs wavelengths");
    if (n == 0) return MI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    TmpBuf<float> d_ref, d_smp, d_wl, d_spec; TmpBuf<mi_direction_sample> d_ds;
    HIP_TRY(c, d_ref.resize(3 * n)); HIP_TRY(c, d_smp.resize(2 * n)); HIP_TRY(c, d_spec.resize(MIW_SPEC_N * n)); HIP_TRY(c, d_ds.resize(n));
    HIP_TRY(c, hipMemcpyAsync(d_ref.p, ref_p, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_smp.p, sample, 2 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (MIW_SPECTRAL) {
        HIP_TRY(c, d_wl.resize(4 * n));
        HIP_TRY(c, hipMemcpyAsync(d_wl.p, wavelengths, 4 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_sample_emitter_direction, dim3((unsigned) ((n + MIW_BLOCK - 1) / MIW_BLOCK)), dim3(MIW_BLOCK), c->lds_bytes, c->stream,
                       c->view, emitter, d_ref.p, d_smp.p, d_wl.p, test_visibility, d_ds.p, d_spec.p, n, c->lds_cfg);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(ds, d_ds.p, n * sizeof(mi_direction_sample), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(spec_out, d_spec.p, MIW_SPEC_N * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MI_OK;
}

mi_status mi_pdf_emitter_direction(mi_ctx *c, int32_t emitter, const float *ref_p, const mi_direction_sample *ds, float *pdf, uint64_t n) {
    if (!c || !ref_p || !ds || !pdf) return MI_ERR_INVALID;
    if (!c->have_bvh) return fail(c, MI_ERR_STATE, "mi_pdf_emitter_direction: call mi_scene_upload and mi_bvh_build first");
    if (emitter >= (int32_t) c->emitters.size()) return fail(c, MI_ERR_INVALID, "mi_pdf_emitter_direction: emitter index out of range");
    if (n == 0) return MI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    TmpBuf<float> d_ref, d_pdf; TmpBuf<mi_direction_sample> d_ds;
    HIP_TRY(c, d_ref.resize(3 * n)); HIP_TRY(c, d_pdf.resize(n)); HIP_TRY(c, d_ds.resize(n));
    HIP_TRY(c, hipMemcpyAsync(d_ref.p, ref_p, 3 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_ds.p, ds, n * sizeof(mi_direction_sample), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pdf_emitter_direction, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, c->view, emitter, d_ref.p, d_ds.p, d_pdf.p, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(pdf, d_pdf.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MI_OK;
}

mi_status mi_emitter_eval(mi_ctx *c, const mi_surface_interaction *si, const float *wavelengths, float *spec_out, uint64_t n) {
    if (!c || !si || !spec_out) return MI_ERR_INVALID;
    if (!c->have_bvh) return fail(c, MI_ERR_STATE, "mi_emitter_eval: call mi_scene_upload and mi_bvh_build first");
    if (MIW_SPECTRAL && !wavelengths) return fail(c, MI_ERR_INVALID, "mi_emitter_eval: the scalar_spectral library needs wavelengths");
    if (n == 0) return MI_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    TmpBuf<float> d_wl, d_spec; TmpBuf<mi_surface_interaction> d_si;
    HIP_TRY(c, d_spec.resize(MIW_SPEC_N * n)); HIP_TRY(c, d_si.resize(n));
    HIP_TRY(c, hipMemcpyAsync(d_si.p, si, n * sizeof(mi_surface_interaction), hipMemcpyHostToDevice, c->stream));
    if (MIW_SPECTRAL) {
        HIP_TRY(c, d_wl.resize(4 * n));
        HIP_TRY(c, hipMemcpyAsync(d_wl.p, wavelengths, 4 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(k_emitter_eval, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, c->view, d_si.p, d_wl.p, d_spec.p, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(spec_out, d_spec.p, MIW_SPEC_N * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MI_OK;
}

// ---- render ----------------------------------------------------------------------------
static mi_status fill_params(mi_ctx *c, const mi_render_cfg *cfg, RenderParams &P) {
    if (cfg->crop_w <= 0 || cfg->crop_h <= 0 || cfg->crop_x < 0 || cfg->crop_y < 0)
        return fail(c, MI_ERR_INVALID, "render: bad crop window");
    if (cfg->crop_x + cfg->crop_w > 65535 || cfg->crop_y + cfg->crop_h > 65535)
        return fail(c, MI_ERR_INVALID, "render: film larger than 65535 pixels per side");
    if (cfg->block_size <= 0 || (cfg->block_size & (cfg->block_size - 1)))
        return fail(c, MI_ERR_INVALID, "render: block_size must be a power of two");
    if (cfg->integrator == MI_INTEGRATOR_PATH && cfg->rr_depth <= 0) return fail(c, MI_ERR_INVALID, "\"rr_depth\" must be set to a value greater than zero!");
    if (cfg->integrator == MI_INTEGRATOR_PATH && cfg->max_depth < 0 && cfg->max_depth != -1) return fail(c, MI_ERR_INVALID, "\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    if (cfg->filter_radius <= 0.f || cfg->filter_radius > 4.f) return fail(c, MI_ERR_INVALID, "render: filter radius out of range (0, 4]");
    memset(&P, 0, sizeof P);
    memcpy(P.sensor.sample_to_camera, cfg->sample_to_camera, 64);
    memcpy(P.sensor.to_world, cfg->to_world, 64);
    P.sensor.near_clip = cfg->near_clip; P.sensor.far_clip = cfg->far_clip;
    P.sensor.pp_offset[0] = cfg->principal_point_offset[0]; P.sensor.pp_offset[1] = cfg->principal_point_offset[1];
    P.film.crop_w = cfg->crop_w; P.film.crop_h = cfg->crop_h; P.film.crop_x = cfg->crop_x; P.film.crop_y = cfg->crop_y;
    P.film.block_size = cfg->block_size; P.film.border = cfg->filter_border;
    P.film.radius = cfg->filter_radius;
    P.film.scale_factor = (float) MIW_FILTER_RESOLUTION / cfg->filter_radius;
    memcpy(P.film.lut, cfg->filter_lut, sizeof P.film.lut);
    P.film.warn_negative = cfg->moment_pass ? 0u : 1u;           // integrator.cpp:113: !has_aovs
    P.spp = cfg->spp; P.max_depth = cfg->max_depth; P.rr_depth = cfg->rr_depth;
    if (cfg->integrator == MI_INTEGRATOR_DIRECT) {               // direct.cpp:82-103
        const uint32_t ne = cfg->emitter_samples, nb = cfg->bsdf_samples;
        if (ne + nb == 0) return fail(c, MI_ERR_INVALID, "Must have at least 1 BSDF or emitter sample!");
        P.integrator = INTEG_DIRECT;
        P.direct.emitter_samples = ne; P.direct.bsdf_samples = nb; P.direct.hide_emitters = cfg->hide_emitters ? 1u : 0u;
        P.direct.weight_bsdf = 1.f / (float) nb; P.direct.weight_lum = 1.f / (float) ne;
        P.direct.frac_bsdf = (float) nb / (float) (ne + nb); P.direct.frac_lum = (float) ne / (float) (ne + nb);
    } else if (cfg->integrator != MI_INTEGRATOR_PATH)
        return fail(c, MI_ERR_INVALID, "render: unknown integrator %d", cfg->integrator);
    if (cfg->moment_pass < MI_MOMENT_OFF || cfg->moment_pass > MI_MOMENT_SQUARES) return fail(c, MI_ERR_INVALID, "render: moment_pass must be 0, 1 or 2");
#if MIW_SPECTRAL
    if (cfg->moment_pass) return fail(c, MI_ERR_INVALID, "render: the moment integrator is provided by the scalar_rgb library only");
#endif
    P.moment_pass = (uint32_t) cfg->moment_pass;
    render_params_prepare(P);
    return MI_OK;
}

// ---- per-launch timing of one mi_render: HIP events from the context's pool around every launch, summed per launch class into mi_counters ----
struct LaunchTimer {
    struct Stamp { int cls; size_t e0, e1; };
    mi_ctx *c; mi_counters &K; bool on;
    std::vector<Stamp> stamps; size_t ev_used = 0;
    hipError_t get_event(size_t &idx) {
        if (ev_used == c->ev_pool.size()) {
            hipEvent_t e; hipError_t r = hipEventCreate(&e);
            if (r != hipSuccess) return r;
            c->ev_pool.push_back(e);
        }
        idx = ev_used++;
        return hipSuccess;
    }
    void drain() {
        if (!on) return;
        for (const Stamp &t : stamps) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev_pool[t.e0], c->ev_pool[t.e1]) != hipSuccess) continue;
            switch (t.cls) {
                case 0: K.ms_trace_closest += ms; break;
                case 1: K.ms_trace_any += ms; break;
                case 2: K.ms_shade += ms; break;
                case 3: K.ms_init += ms; break;
                case 6: K.ms_path += ms; break;
                case 4: K.ms_film_blocks += ms; K.ms_resolve += ms; break;
                case 5: K.ms_film_merge += ms; K.ms_resolve += ms; break;
                default: K.ms_resolve += ms; break;
            }
        }
        stamps.clear(); ev_used = 0;
    }
};

// ---- mi_render, phase "film plan": how ImageBlock::put is realised for this job (mi_render_cfg::film_mode), the sample log's record
// format (16-byte class records where film_classes.h covers the filter, 24-byte positions otherwise), its layout ([lane][sample], or interleaved
// over groups of 64 tiles for k_film_lanes) and its allocation ----
struct FilmPlan { int film_mode = 0; bool rec16 = false; int film_lanes = 0; uint32_t log_il = 0; size_t log_entries = 0, rec_bytes = 0; };
static mi_status plan_film_log(mi_ctx *c, const mi_render_cfg *cfg, const Options &ropt, const RenderParams &P, hipStream_t s, int plan,
                               uint32_t n_tiles, uint32_t n_lanes, uint32_t bs2, uint32_t bs2_log2, size_t film_n, FilmPlan &FP) {
    (void) plan;
    // film mode: sample log + ordered gather if the log fits, else float64 atomics
    int film_mode = cfg->film_mode;
    const size_t nl = std::max<uint32_t>(n_lanes, 1);
    if (film_mode < 0 || film_mode > 2) return fail(c, MI_ERR_INVALID, "render: film_mode must be 0, 1 or 2");
    const size_t log_lanes_entries = (size_t) nl * std::max<uint32_t>(cfg->spp, 1);
    // The log format: 16-byte records with phase classes where the host enumeration covers the filter (film_classes.h: box, tent,
    // gaussian, mitchell, catmullrom), positions + values (24 bytes) otherwise. MIW_FILM_LEGACY=1 forces the latter (A/B runs).
    bool rec16 = false;
    if (film_mode != 2 && bs2 <= 65536u && !ropt.get("MIW_FILM_LEGACY")) {
        // the cache key: the filter's fields only, in a zero-filled record (padding bytes of a stack copy would make memcmp miss)
        FilmRec key; memset(&key, 0, sizeof key);
        key.border = P.film.border; key.radius = P.film.radius; key.scale_factor = P.film.scale_factor; memcpy(key.lut, P.film.lut, sizeof key.lut);
        // (a filter the enumeration refuses is remembered as well — classes_valid — instead of being enumerated again by every render)
        if (!c->classes_valid || memcmp(&key, &c->classes_of, sizeof key) != 0 || (c->classes.ok && !c->d_fc_thr.p)) {
            c->classes = film_classes_build(P.film);             // ~30 ms, once per filter
            c->classes_of = key; c->classes_valid = true;
            if (c->classes.ok) { HIP_TRY(c, c->d_fc_thr.upload(c->classes.thr, s)); HIP_TRY(c, c->d_fc_w.upload(c->classes.w, s)); HIP_TRY(c, hipStreamSynchronize(s)); }
        }
        rec16 = c->classes.ok;
    }
    // The replay kernel for the 16-byte records (device/film_kernels.h): k_film_lanes — one 4 x 4 texel block per lane, a wavefront = one
    // block position in 64 consecutive tiles — wants those tiles' logs interleaved record by record (path.h: log_index), so the
    // choice is made before the render kernels write the log. MIW_FILM_LANES = 0, or naming another kernel's shape
    // (MIW_FILM_QUADS / _COLUMNS / _GROUP), keeps [lane][sample] and the group kernels; MIW_FILM_LANES = 2: k_film_lanes over [lane][sample].
    // k_film_lanes' wavefronts are few and long (81 per 64 tiles, each the serial replay of 64 pixel runs: ~6 ms at 512 spp however few
    // there are), so shards of fewer than 7 x 64 tiles keep the group kernel: 255 tiles (a rank's eighth of a 1080p frame) 4.1 ms by
    // k_film_quads against 6.1, 510 tiles 7.1 against 6.65, 1 020 tiles 12.9 against 10.2 (gpurun q10)
    int film_lanes = rec16 && c->classes.reach <= 2 && n_tiles >= 448u && !ropt.get("MIW_FILM_COLUMNS") && !ropt.get("MIW_FILM_GROUP") && !ropt.get("MIW_FILM_QUADS") ? 1 : 0;
    if (const char *e = ropt.get("MIW_FILM_LANES")) film_lanes = rec16 && c->classes.reach <= 2 ? atoi(e) : 0;
    const uint32_t log_il = film_lanes == 1 ? bs2_log2 + 1u : 0u;
    const size_t log_entries = log_il ? log_capacity(log_il, n_tiles, bs2, std::max<uint32_t>(cfg->spp, 1)) : log_lanes_entries;
    const size_t rec_bytes = rec16 ? sizeof(U4) : sizeof(F2) + sizeof(F4);
    if (film_mode != 2) {
        size_t need = log_entries * rec_bytes;
        size_t have = c->q_log_pos.n * sizeof(F2) + c->q_log_val.n * sizeof(F4) + c->q_log_rec.n * sizeof(U4), free_b = 0, total_b = 0;
        HIP_TRY(c, hipMemGetInfo(&free_b, &total_b));
        bool fits = need <= (size_t) ((double) (free_b + have) * 0.8);
        if (!fits) {
            if (film_mode == 1) return fail(c, MI_ERR_INVALID, "render: sample log needs %zu MiB, only %zu MiB free", need >> 20, free_b >> 20);
            film_mode = 2; rec16 = false;
        } else film_mode = 1;
    }
    if (film_mode == 1) {
        if (rec16) {
            if (c->q_log_rec.n < log_entries) { c->q_log_rec.release(); c->q_log_pos.release(); c->q_log_val.release(); }
            HIP_TRY(c, c->q_log_rec.resize(log_entries));
        } else {
            if (c->q_log_pos.n < log_entries) { c->q_log_pos.release(); c->q_log_val.release(); c->q_log_rec.release(); }
            HIP_TRY(c, c->q_log_pos.resize(log_entries)); HIP_TRY(c, c->q_log_val.resize(log_entries));
        }
    } else {
        HIP_TRY(c, c->d_accum.resize(film_n));
        HIP_TRY(c, hipMemsetAsync(c->d_accum.p, 0, film_n * sizeof(double), s));
    }
#if defined(MIW_DEBUG_POISON)
    // Debug tier (-DMIW_DEBUG_POISON=1 builds; the reference poisons its GPU interactions in debug builds the same way,
    // src/librender/scene_optix.inl:475-480): every buffer a render kernel is supposed to WRITE before anything reads it starts as
    // NaN bit patterns (0xff bytes) — the sample log, plan 1's queues, the block tiles — and mi_render fails when a NaN reaches the film.
    // A slot that is read without having been written (an indexing error, a lane that skipped its log write) then shows up as
    // MI_ERR_STATE instead of as a plausible-looking stale value from the previous frame.
    {
        auto poison = [&](void *p, size_t bytes) -> hipError_t { return p && bytes ? hipMemsetAsync(p, 0xff, bytes, s) : hipSuccess; };
        HIP_TRY(c, poison(c->q_log_rec.p, c->q_log_rec.n * sizeof(U4))); HIP_TRY(c, poison(c->q_log_pos.p, c->q_log_pos.n * sizeof(F2))); HIP_TRY(c, poison(c->q_log_val.p, c->q_log_val.n * sizeof(F4)));
        if (plan == 1) for (DevBuf<F4> *q : { &c->q_tp, &c->q_res, &c->q_ray_o, &c->q_ray_d, &c->q_hit, &c->q_sh_d, &c->q_sh_c }) HIP_TRY(c, poison(q->p, q->n * sizeof(F4)));
        HIP_TRY(c, poison(c->d_tiles.p, c->d_tiles.n * sizeof(float)));
    }
#endif
    c->counters.log_bytes = film_mode == 1 ? (uint64_t) log_entries * rec_bytes : 0u;
    c->counters.log_record_bytes = film_mode == 1 ? (uint32_t) rec_bytes : 0u;
    c->counters.film_kernel = 0u; c->counters.log_interleaved = film_mode == 1 && rec16 && log_il ? 1u : 0u;
    c->counters.film_mode = (uint32_t) film_mode;
    FP.film_mode = film_mode; FP.rec16 = rec16; FP.film_lanes = film_lanes; FP.log_il = log_il; FP.log_entries = log_entries; FP.rec_bytes = rec_bytes;
    return MI_OK;
}

// ---- mi_render, debug builds: what the instrumented kernels counted (printed when the option MIW_DEBUG is set) ----
static void print_debug_statistics(mi_ctx *c, const Options &ropt, mi_counters &K) {
    (void) c; (void) ropt; (void) K;
#if defined(MIW_VERIFY_FILTER)
        {
            unsigned int n = 0; float buf[256];
            (void) hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_verify_n), sizeof n);
            (void) hipMemcpyFromSymbol(buf, HIP_SYMBOL(g_verify), sizeof buf);
            fprintf(stderr, "[miwave] filter verification: %u mismatching queries\n", n);
            for (unsigned i = 0; i < n && i < 16; ++i) {
                const float *r = buf + i * 16; uint32_t a, b; memcpy(&a, r + 9, 4); memcpy(&b, r + 10, 4);
                fprintf(stderr, "  kind %g o=(%.9g %.9g %.9g) mint=%.9g d=(%.9g %.9g %.9g) maxt=%.9g brute=%u filter=%u t_brute=%.9g t_filter=%.9g\n",
                        r[8], r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], a, b, r[11], r[12]);
            }
            n = 0; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_verify_n), &n, sizeof n);
        }
#endif
#if defined(MIW_PHASE_STATS)
        if (ropt.get("MIW_DEBUG")) {
            unsigned long long ps[15];
            if (!K.pooled && hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_phase_stats), sizeof ps) == hipSuccess) {
                static const char *names[5] = { "node step", "triangle test", "walk end", "shade", "vote" };
                unsigned long long tot = 0;
                for (int k = 0; k < 5; ++k) tot += ps[10 + k];
                for (int k = 0; k < 5; ++k)
                    fprintf(stderr, "[miwave] phase %-13s runs/segment %7.2f  lanes/run %5.1f  cycles/run %7.0f  share of wave cycles %5.1f %%\n", names[k],
                            64.0 * (double) ps[k] / std::max<double>((double) K.segments, 1), (double) ps[5 + k] / std::max<double>((double) ps[k], 1),
                            (double) ps[10 + k] / std::max<double>((double) ps[k], 1), 100.0 * (double) ps[10 + k] / std::max<double>((double) tot, 1));
                memset(ps, 0, sizeof ps);
                (void) hipMemcpyToSymbol(HIP_SYMBOL(g_phase_stats), ps, sizeof ps);
            }
        }
#endif
#if defined(MIW_PHASE_STATS) && !MIW_SPECTRAL
        if (ropt.get("MIW_DEBUG") && K.pooled) {
            unsigned long long ps[32];
            if (hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pool_stats), sizeof ps) == hipSuccess) {
                static const char *names[5] = { "vote", "node trip", "triangle trip", "shade", "idle" };
                unsigned long long tot = 0;
                for (int k = 0; k < 5; ++k) tot += ps[10 + k];
                for (int k = 0; k < 5; ++k)
                    fprintf(stderr, "[miwave] pooled %-13s runs per 64 segments %7.2f  lanes/run %5.1f  cycles/run %7.0f  share of wave cycles %5.1f %%\n", names[k],
                            64.0 * (double) ps[k] / std::max<double>((double) K.segments, 1), (double) ps[5 + k] / std::max<double>((double) ps[k], 1),
                            (double) ps[10 + k] / std::max<double>((double) ps[k], 1), 100.0 * (double) ps[10 + k] / std::max<double>((double) tot, 1));
                fprintf(stderr, "[miwave] pooled claims: %.2f tried per segment, %.1f %% won\n", (double) ps[15] / std::max<double>((double) K.segments, 1), 100.0 * (double) ps[16] / std::max<double>((double) ps[15], 1));
                memset(ps, 0, sizeof ps);
                (void) hipMemcpyToSymbol(HIP_SYMBOL(g_pool_stats), ps, sizeof ps);
            }
        }
#endif
#if defined(MIW_WALK_STATS)
        if (ropt.get("MIW_DEBUG")) {
            unsigned long long st[8]; float stf[8];
            if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_walk_stats), sizeof st) == hipSuccess && hipMemcpyFromSymbol(stf, HIP_SYMBOL(g_walk_statsf), sizeof stf) == hipSuccess) {
                for (int k = 0; k < 2; ++k) {
                    const unsigned long long *a = st + 4 * k; const float *f = stf + 4 * k;
                    fprintf(stderr, "[miwave] walk stats %s: rays %llu, node steps/ray %.2f (SIMT eff %.3f), triangle tests/ray %.2f (SIMT eff %.3f)\n",
                            k ? "any-hit" : "closest", a[2], (double) a[0] / std::max<double>(a[2], 1), (double) a[0] / (64.0 * std::max(f[0], 1.f)),
                            (double) a[1] / std::max<double>(a[2], 1), (double) a[1] / (64.0 * std::max(f[1], 1.f)));
                }
                memset(st, 0, sizeof st); memset(stf, 0, sizeof stf);
                (void) hipMemcpyToSymbol(HIP_SYMBOL(g_walk_stats), st, sizeof st); (void) hipMemcpyToSymbol(HIP_SYMBOL(g_walk_statsf), stf, sizeof stf);
            }
        }
#endif
#if defined(MIW_SECTION_PROFILE)
        if (ropt.get("MIW_DEBUG")) {
            unsigned long long sec[16];
            if (hipMemcpyFromSymbol(sec, HIP_SYMBOL(g_sections), sizeof sec) == hipSuccess) {
                static const char *names[13] = { "fetch/begin", "leaf boxes", "E candidates", "S candidates", "path_step", "finish+begin",
                                                 "walks + votes", "shade: surface interaction", "shade: emitter hit + MIS", "shade: RR + emitter sampling + bsdf eval",
                                                 "shade: bsdf sample", "shade: finish + next sample", "shade: pixel fetch + walk start" };
                unsigned long long tot = 0;
                for (int i = 0; i < 13; ++i) tot += sec[i];
                for (int i = 0; i < 13; ++i) if (sec[i]) fprintf(stderr, "[miwave] section %-40s %6.2f %%\n", names[i], 100.0 * (double) sec[i] / (double) std::max<unsigned long long>(tot, 1));
                memset(sec, 0, sizeof sec);
                (void) hipMemcpyToSymbol(HIP_SYMBOL(g_sections), sec, sizeof sec);
            }
        }
#endif
}

// ---- mi_render, what every replay kernel is handed: the log, the pixel states, the class tables, the block -> tile map of this shard (uploaded from a
// context-owned vector: nobody waits for that copy), the tile buffer ----
struct FilmReplay { BlockReplayArgs A; uint32_t side = 0; uint32_t beside = 0; /* groups of 64 tiles replayed beside the path kernel (overlap_enqueue); the launch after it takes the rest */ };
static mi_status film_replay_setup(mi_ctx *c, const mi_render_cfg *cfg, hipStream_t s, const FilmPlan &FP, uint32_t n_tiles, uint32_t bs, uint32_t bs2_log2,
                                   uint32_t blocks_x, uint32_t blocks_y, FilmReplay &R) {
    const bool rec16 = FP.rec16;
    std::vector<int32_t> &block_tile = c->h_block_tile;
    block_tile.assign(cfg->block_count, -1);
    for (uint32_t t = 0; t < n_tiles; ++t) block_tile[cfg->tile_list ? cfg->tile_list[t] : t] = (int32_t) t;
    HIP_TRY(c, c->d_block_tile.resize(cfg->block_count));
    HIP_TRY(c, hipMemcpyAsync(c->d_block_tile.p, block_tile.data(), block_tile.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    BlockReplayArgs &A = R.A;
    A.log_pos = c->q_log_pos.p; A.log_val = c->q_log_val.p; A.st = c->q_st.p; A.spp = cfg->spp;
    A.log_rec = rec16 ? c->q_log_rec.p : nullptr; A.log_il = rec16 ? FP.log_il : 0u;
    A.cls.thr = c->d_fc_thr.p; A.cls.w = c->d_fc_w.p; A.cls.count = rec16 ? c->classes.count : 0u; A.cls.reach = rec16 ? c->classes.reach : 0;
    A.block_ids = c->d_block_ids.p; A.block_tile = c->d_block_tile.p;
    A.tile_list = cfg->tile_list ? c->d_tile_list.p : nullptr;
    A.blocks_x = blocks_x; A.blocks_y = blocks_y; A.bs2_log2 = bs2_log2;
    R.side = bs + 2u * (uint32_t) cfg->filter_border;
    A.tile_stride = R.side * R.side * MIW_FILM_CHANNELS;
    HIP_TRY(c, c->d_tiles.resize((size_t) std::max<uint32_t>(n_tiles, 1) * A.tile_stride));
    return MI_OK;
}

// The film replay of a full frame BESIDE its render (round 6). k_film_lanes waits on memory (a quarter of its issue slots used), the path kernels on
// issue slots (0.14 TB/s of HBM), and a path kernel's last ~30 ms run at falling occupancy (a lane's last pixel is a serial stream of spp samples): so
// the replay is queued on a second stream BEFORE the path kernel has run, one launch per group of 64 tiles (the unit of the tile-interleaved log), each
// behind a stream-level wait (hipStreamWaitValue32: the command processor polls, no wavefront spins) for the flag the group's LAST finished pixel raises
// (resident_kernel.h: QueueWork::store counts finished pixels per group). The groups finish in queue order, so their replays run in the wave slots the
// path kernel's tail frees, and only the last groups' replay is left when the path kernel ends. Same kernels, same log, same additions: the same film.
// A raise-all launch behind the path kernel releases every wait whatever the counts said (a stream can never be left waiting).
__global__ void k_raise_flags(uint32_t *flag, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_store(flag + i, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static mi_status overlap_prepare(mi_ctx *c, const mi_render_cfg *cfg, hipStream_t s, uint32_t n_tiles, uint32_t bs, uint32_t blocks_x, LaneQueues &Q, uint32_t bs2_log2) {
    const uint32_t n_groups = (n_tiles + 63u) / 64u;
    if (!c->ev_fork) {
        bool ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 3 && ok; ++i)
            ok = hipStreamCreateWithFlags(&c->stream2[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void) hipGetLastError(); c->overlap_state = 0; return MI_OK; }
    }
    if (c->group_flag_cap < n_groups) {
        if (c->h_group_flag) (void) hipHostFree(c->h_group_flag);
        c->h_group_flag = nullptr; c->group_flag_cap = 0;
        const uint32_t cap = std::max<uint32_t>(1024u, n_groups);
        if (hipHostMalloc((void **) &c->h_group_flag, cap * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { (void) hipGetLastError(); c->overlap_state = 0; return MI_OK; }
        c->group_flag_cap = cap;
    }
    // pixels per group: the pixels of its tiles that lie inside the film (k_init_pixels marks the others done: they are never fetched, never stored)
    std::vector<uint32_t> &expected = c->h_group_expected;
    expected.assign(n_groups, 0u);
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t b = cfg->tile_list ? cfg->tile_list[t] : t, bx = b % blocks_x, by = b / blocks_x;
        const int32_t bw = std::min<int32_t>((int32_t) bs, cfg->crop_w - (int32_t) (bx * bs)), bh = std::min<int32_t>((int32_t) bs, cfg->crop_h - (int32_t) (by * bs));
        expected[t >> 6] += (uint32_t) (std::max(bw, 0) * std::max(bh, 0));
    }
    HIP_TRY(c, c->d_group_done.resize(n_groups)); HIP_TRY(c, c->d_group_expected.resize(n_groups));
    memset(c->h_group_flag, 0, n_groups * sizeof(uint32_t));     // (the previous frame's stream work is complete: mi_render ends with a wait)
    HIP_TRY(c, hipMemsetAsync(c->d_group_done.p, 0, n_groups * sizeof(uint32_t), s));
    HIP_TRY(c, hipMemcpyAsync(c->d_group_expected.p, expected.data(), n_groups * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    Q.group_done = c->d_group_done.p; Q.group_expected = c->d_group_expected.p; Q.group_flag = c->h_group_flag; Q.group_shift = bs2_log2 + 6u;
    return MI_OK;
}
// after the path kernel has been launched on `s`: fork, the waits + launches on stream2, the release-all behind the path kernel, the join event
static mi_status overlap_enqueue(mi_ctx *c, const Options &ropt, const RenderParams &P, hipStream_t s, LaunchTimer &T, FilmReplay &R, uint32_t n_tiles, bool wait_flags) {
    const uint32_t n_all = (n_tiles + 63u) / 64u;
    // how many groups are replayed BESIDE the path kernel; the rest in one launch after it (assemble_film). What runs after it cannot take less than one
    // wavefront's chain (~8 - 10 ms) however little it is; measured at C2 (gpurun r6m): three quarters beside 263.6 ms, half 266.1, none 265.8
    uint32_t n_groups = n_all * 3u / 4u;
    if (const char *e = ropt.get("MIW_FILM_OVERLAP_BESIDE")) n_groups = (uint32_t) std::min<long>(std::max<long>(atol(e), 0), (long) n_all);
    int prio = 0;
    if (const char *e = ropt.get("MIW_FILM_OVERLAP_PRIO")) prio = std::min(3, std::max(0, atoi(e)));
    PatchArgs PC; PC.patches_x = PC.patches_y = (R.side + MIW_FL_BS - 1) / MIW_FL_BS; PC.reach = c->classes.reach;
    const uint32_t n_pos = PC.patches_x * PC.patches_y;
    const size_t lbytes = (size_t) (c->classes.count + 1u) * MIW_FQ_WSTRIDE(MIW_FL_BS) * sizeof(float);
    int fl_nt = 1;
    if (const char *e = ropt.get("MIW_FL_NT")) fl_nt = atoi(e);
    // One launch replays a SET of groups (MIW_FILM_OVERLAP_SET, default 4: 324 wavefronts) behind the flags of all of them; the sets go round the three
    // replay streams — kernels of one stream run one after the other, and a replay kernel lasts as long as ONE of its wavefronts (64 pixel runs, a
    // serial chain of ~4 ms) however few there are: one launch per group on one stream would be a chain of 32 such kernels.
    uint32_t set = 4u;
    if (const char *e = ropt.get("MIW_FILM_OVERLAP_SET")) set = (uint32_t) std::max(1, atoi(e));
    bool lean_ok = fl_nt != 0;
    if (const char *e = ropt.get("MIW_FILM_OVERLAP_LEAN")) lean_ok = lean_ok && atoi(e) != 0;
    for (int i = 0; i < 3; ++i) HIP_TRY(c, hipStreamWaitEvent(c->stream2[i], c->ev_fork, 0));
    uint32_t launches = 0;
    for (uint32_t g0 = 0; g0 < n_groups; g0 += set, ++launches) {
        hipStream_t b = c->stream2[launches % 3u];
        const uint32_t g1 = std::min(n_groups, g0 + set);
        if (wait_flags)
            for (uint32_t g = g0; g < g1; ++g) {
                const hipError_t e = hipStreamWaitValue32(b, c->h_group_flag + g, 1u, hipStreamWaitValueGte, 0xffffffffu);
                if (e != hipSuccess) {                           // the runtime refuses stream memory operations
                    (void) hipGetLastError();
                    c->overlap_state = 0;
                    if (g == 0) return MI_OK;                    // nothing has been launched: assemble_film replays as before
                    return fail(c, MI_ERR_DEVICE, "hipStreamWaitValue32 failed mid-frame: %s", hipGetErrorString(e));
                }
            }
        size_t e0 = 0, e1 = 0;
        if (T.on) { HIP_TRY(c, T.get_event(e0)); HIP_TRY(c, hipEventRecord(c->ev_pool[e0], b)); }
        // (a partial last group of tiles: the kernel's lanes past n_tiles idle, as in the one-launch replay)
        // these sets run while the path kernel does — in the wave slots mi_render keeps free of it (MIW_FILM_OVERLAP_ROOM: on that many CUs the path kernel
        // has three workgroups instead of four, which leaves 152 registers per SIMD) — as the 128-register form of the replay (two records in flight per lane)
        const bool lean = lean_ok;
        if (lean) hipLaunchKernelGGL((k_film_lanes<2, 1, 4>), dim3((g1 - g0) * n_pos), dim3(64), lbytes, b, P.film, R.A, PC, (uint32_t) n_tiles, c->d_tiles.p, g0, (uint32_t) prio);
        else if (fl_nt) hipLaunchKernelGGL((k_film_lanes<4, 1>), dim3((g1 - g0) * n_pos), dim3(64), lbytes, b, P.film, R.A, PC, (uint32_t) n_tiles, c->d_tiles.p, g0, (uint32_t) prio);
        else hipLaunchKernelGGL((k_film_lanes<4, 0>), dim3((g1 - g0) * n_pos), dim3(64), lbytes, b, P.film, R.A, PC, (uint32_t) n_tiles, c->d_tiles.p, g0, (uint32_t) prio);
        if (T.on) { HIP_TRY(c, T.get_event(e1)); HIP_TRY(c, hipEventRecord(c->ev_pool[e1], b)); T.stamps.push_back({ 4, e0, e1 }); }
    }
    for (int i = 0; i < 3; ++i) HIP_TRY(c, hipEventRecord(c->ev_join[i], c->stream2[i]));
    c->replay_launches = launches;
    R.beside = n_groups;
    hipLaunchKernelGGL(k_raise_flags, dim3((n_all + 255u) / 256u), dim3(256), 0, s, c->h_group_flag, n_all);   // behind the path kernel: no wait outlives it
    HIP_TRY(c, hipGetLastError());
    c->replay_enqueued = true;
    c->overlap_state = 1;
    return MI_OK;
}

// ---- mi_render, phase "film": the ordered replay of the sample log into block tiles (device/film_kernels.h; which kernel: the log
// format, the shard's tile count, the options) + the merge of the tiles into the film, or the resolve of the float64 sums ----
static mi_status assemble_film(mi_ctx *c, const mi_render_cfg *cfg, const Options &ropt, const RenderParams &P, hipStream_t s, LaunchTimer &T, const FilmPlan &FP,
                               uint32_t n_tiles, uint32_t bs, uint32_t bs2_log2, uint32_t blocks_x, uint32_t blocks_y, size_t film_n, void *film, mi_status result,
                               const FilmReplay *queued /* != nullptr: the replay's launches already sit in stream2 (overlap_enqueue) */) {
    const int film_mode = FP.film_mode; const bool rec16 = FP.rec16; const int film_lanes = FP.film_lanes; const uint32_t log_il = FP.log_il;
    (void) result;
#define MIW_TIMED(cls_, launch) do {                                                   \
        size_t e0_ = 0, e1_ = 0;                                                       \
        if (T.on) { HIP_TRY(c, T.get_event(e0_)); HIP_TRY(c, hipEventRecord(c->ev_pool[e0_], s)); } \
        launch;                                                                        \
        if (T.on) { HIP_TRY(c, T.get_event(e1_)); HIP_TRY(c, hipEventRecord(c->ev_pool[e1_], s)); \
                    T.stamps.push_back({ cls_, e0_, e1_ }); }                          \
    } while (0)
    {
        const size_t elem = cfg->film_f64 ? sizeof(double) : sizeof(float);
        void *dst = film;
        const bool staged = !cfg->film_on_device;
        if (staged) {
            HIP_TRY(c, c->d_out.resize(film_n * elem)); dst = c->d_out.p;
            if (cfg->accumulate) HIP_TRY(c, hipMemcpyAsync(c->d_out.p, film, film_n * elem, hipMemcpyHostToDevice, s));
        }
        float *dst32 = cfg->film_f64 ? nullptr : (float *) dst;
        double *dst64 = cfg->film_f64 ? (double *) dst : nullptr;
        if (film_mode == 1) {
            FilmReplay R;
            if (queued) R = *queued;                              // (set up before the path kernel was launched)
            else { const mi_status rs = film_replay_setup(c, cfg, s, FP, n_tiles, bs, bs2_log2, blocks_x, blocks_y, R); if (rs != MI_OK) return rs; }
            BlockReplayArgs &A = R.A;
            const uint32_t side = R.side;
            if (n_tiles) {
                PatchArgs PA;
                PA.patches_x = PA.patches_y = (side + MIW_FP_SIDE - 1) / MIW_FP_SIDE;
                PA.reach = (int32_t) floorf(cfg->filter_radius + .5f);
                const dim3 fgrid(n_tiles * PA.patches_x * PA.patches_y);
                const bool wide = cfg->filter_radius > 0.5f + MIW_RAY_EPSILON;
                if (rec16) {
                    // 16-byte class records -> per-group pixel lists (k_film_groups). Group shape: 2x2 reads 6.25 sample rows per texel,
                    // 4x4 3.06 but spends 0.77 wave-iterations per sample; 4x2 (4.4 rows, 0.55) balances HBM reads against issue slots.
                    // MIW_FILM_GROUP = 2 | 3 | 4 overrides (2x2 / 4x2 / 4x4).
                    int group = 3;
                    if (const char *e = ropt.get("MIW_FILM_GROUP")) group = atoi(e);
                    const size_t wbytes = (size_t) (c->classes.count + 1u) * MIW_FG_WSTRIDE * sizeof(float);
#define MIW_FG_LAUNCH(GW, GH) MIW_TIMED(4, hipLaunchKernelGGL((k_film_groups<GW, GH>), fgrid, dim3(64), wbytes, s, P.film, A, PA, c->d_tiles.p))
                    // round 4: a column of GH texels per lane (k_film_columns; MIW_FILM_COLUMNS = 0: the one-texel-per-lane kernel, 42 / 44: 4 x 2 / 4 x 4 groups;
                    // round 5 measured 2 x 4 and 2 x 2 groups as well: 27.6 / 27.5 ms against 25.6 at C2, gpurun r5p — not kept)
#define MIW_FC_LAUNCH(GW, GH) do { PatchArgs PC = PA; PC.patches_x = (side + (GW) - 1) / (GW); PC.patches_y = (side + (GH) - 1) / (GH); \
                                   const uint32_t per_wave = 64u / (GW), wpt = (PC.patches_x * PC.patches_y + per_wave - 1u) / per_wave; \
                                   MIW_TIMED(4, hipLaunchKernelGGL((k_film_columns<GW, GH>), dim3(n_tiles * wpt), dim3(64), wbytes, s, P.film, A, PC, c->d_tiles.p)); } while (0)
                    // round 5: groups of 4 x 2 (2 x 4: MIW_FILM_QUADS = 24) texels inside DPP quads, the records broadcast by quad_perm operands instead of
                    // staged through LDS (k_film_quads; MIW_FILM_QUADS = 24 / 28 / 44: other group shapes, = 0: the kernels below — also taken when a tile
                    // has fewer groups than a wavefront takes, i.e. tiny blocks)
                    int columns = 42;
                    if (const char *e = ropt.get("MIW_FILM_COLUMNS")) columns = atoi(e);
                    // (what runs here by default are shards of fewer than 448 tiles — k_film_lanes takes the rest —: 4 x 2 groups, twice as many and half as
                    // long wavefronts as 2 x 4: a rank's eighth of a 1080p frame 3.63 ms against 4.13, k_film_columns 4.19, gpurun q11; a whole frame 24.1 / 23.8)
                    int quads = ropt.get("MIW_FILM_COLUMNS") == nullptr && ropt.get("MIW_FILM_GROUP") == nullptr ? 42 : 0;
                    if (const char *e = ropt.get("MIW_FILM_QUADS")) quads = atoi(e) == 1 ? 42 : atoi(e);
                    if ((quads != 24 && quads != 42 && quads != 28 && quads != 44) || c->classes.reach > 2) quads = 0;   // (the kernel's LDS rows hold windows of <= 5 weights)
                    const uint32_t qw = (uint32_t) quads / 10u, qh = (uint32_t) quads % 10u;
                    if (quads && ((side + qw - 1) / qw) * ((side + qh - 1) / qh) < 64u / qw) quads = 0;
                    // round 5, the default: one 4 x 4 texel block per lane, a wavefront = one block position in 64 tiles, the sample loop specialised
                    // for the rows / column pairs a pixel's footprint covers (k_film_lanes; MIW_FILM_LANES = 0: the kernels below)
                    const bool lanes = film_lanes != 0;
                    c->counters.film_kernel = lanes ? 4u : quads ? 3u : (columns == 42 || columns == 44 || columns == 82) ? 2u : 1u;   // (mi_counters)
                    if (lanes) {
                        PatchArgs PC = PA; PC.patches_x = PC.patches_y = (side + MIW_FL_BS - 1) / MIW_FL_BS;
                        const size_t lbytes = (size_t) (c->classes.count + 1u) * MIW_FQ_WSTRIDE(MIW_FL_BS) * sizeof(float);
                        int fl_nt = 1;                                    // the log read with streaming loads (16.8 vs 17.05 ms at C2, gpurun q9); MIW_FL_NT = 0: plain loads
                        if (const char *e = ropt.get("MIW_FL_NT")) fl_nt = atoi(e);
                        // (the groups of 64 tiles below `first` were replayed beside the path kernel, on the replay streams: overlap_enqueue)
                        const uint32_t first = queued ? queued->beside : 0u, groups_all = ((uint32_t) n_tiles + 63u) / 64u;
                        const uint32_t waves_rest = (groups_all - first) * PC.patches_x * PC.patches_y;
                        if (waves_rest == 0u) { }
                        else if (fl_nt) MIW_TIMED(4, hipLaunchKernelGGL((k_film_lanes<4, 1>), dim3(waves_rest), dim3(64), lbytes, s, P.film, A, PC, (uint32_t) n_tiles, c->d_tiles.p, first, 0u));
                        else MIW_TIMED(4, hipLaunchKernelGGL((k_film_lanes<4, 0>), dim3(waves_rest), dim3(64), lbytes, s, P.film, A, PC, (uint32_t) n_tiles, c->d_tiles.p, first, 0u));
                        if (queued) { for (int i = 0; i < 3; ++i) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join[i], 0)); }
                    } else if (quads) {
                        PatchArgs PC = PA; PC.patches_x = (side + qw - 1) / qw; PC.patches_y = (side + qh - 1) / qh;
                        const uint32_t per_wave = 64u / qw, waves = (uint32_t) (((size_t) n_tiles * PC.patches_x * PC.patches_y + per_wave - 1u) / per_wave);
                        const size_t qbytes = (size_t) (c->classes.count + 1u) * (size_t) (5u + 2u * qh) * sizeof(float);
#define MIW_FQ_LAUNCH(GW, GH, U) MIW_TIMED(4, hipLaunchKernelGGL((k_film_quads<GW, GH, U>), dim3(waves), dim3(64), qbytes, s, P.film, A, PC, (uint32_t) n_tiles, c->d_tiles.p))
                        int fq_u = 4;
                        if (const char *e = ropt.get("MIW_FQ_U")) fq_u = atoi(e);
                        if (quads == 24) { if (fq_u == 8) MIW_FQ_LAUNCH(2, 4, 8); else if (fq_u == 2) MIW_FQ_LAUNCH(2, 4, 2); else MIW_FQ_LAUNCH(2, 4, 4); }
                        else if (quads == 42) { if (fq_u == 8) MIW_FQ_LAUNCH(4, 2, 8); else MIW_FQ_LAUNCH(4, 2, 4); }
                        else if (quads == 28) MIW_FQ_LAUNCH(2, 8, 4); else MIW_FQ_LAUNCH(4, 4, 4);
#undef MIW_FQ_LAUNCH
                    } else if (columns == 42) MIW_FC_LAUNCH(4, 2); else if (columns == 44) MIW_FC_LAUNCH(4, 4); else if (columns == 82) MIW_FC_LAUNCH(8, 2);
                    else if (group == 4) MIW_FG_LAUNCH(4, 4); else if (group == 2) MIW_FG_LAUNCH(2, 2); else MIW_FG_LAUNCH(4, 2);
#undef MIW_FC_LAUNCH
#undef MIW_FG_LAUNCH
                } else if (wide)
                    MIW_TIMED(4, hipLaunchKernelGGL(k_film_blocks<true>, fgrid, dim3(64), 0, s, P.film, A, PA, c->d_tiles.p));
                else
                    MIW_TIMED(4, hipLaunchKernelGGL(k_film_blocks<false>, fgrid, dim3(64), 0, s, P.film, A, PA, c->d_tiles.p));
            }
            MIW_TIMED(5, hipLaunchKernelGGL(k_film_merge, dim3(((uint32_t) cfg->crop_w + 255u) / 256u, (uint32_t) cfg->crop_h), dim3(256), 0, s,
                                            P.film, A, c->d_tiles.p, dst32, dst64, cfg->accumulate));
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(s));            // block_tile (host vector) must outlive the copy
        } else {
            dim3 block(256), grid((unsigned) ((film_n + 255) / 256));
            MIW_TIMED(4, hipLaunchKernelGGL(k_film_resolve, grid, block, 0, s, c->d_accum.p, dst32, dst64, film_n, cfg->accumulate));
        }
        if (staged) HIP_TRY(c, hipMemcpyAsync(film, c->d_out.p, film_n * elem, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(s));
        T.drain();
#if defined(MIW_DEBUG_POISON)
        if (result == MI_OK) {                                    // (a cancelled render leaves unwritten slots by design)
            std::vector<unsigned char> host(film_n * elem);
            HIP_TRY(c, hipMemcpy(host.data(), dst, film_n * elem, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < film_n; ++i) bad += cfg->film_f64 ? std::isnan(((const double *) host.data())[i]) : std::isnan(((const float *) host.data())[i]);
            if (bad) return fail(c, MI_ERR_STATE, "render (poisoned build): %zu film values are NaN - a log / queue / tile slot was read before it was written", bad);
        }
#endif
    }
#undef MIW_TIMED
    return MI_OK;
}

mi_status mi_render(mi_ctx *c, const mi_render_cfg *cfg, void *film) {
    if (!c || !cfg || !film) return MI_ERR_INVALID;
    if (!c->have_bvh) return fail(c, MI_ERR_STATE, "mi_render: call mi_scene_upload and mi_bvh_build first");
    // this render's options: the context's, under the job's debug overrides (include/miwave.h: mi_render_cfg::debug_*)
    Options ropt = c->opt;
    switch (cfg->debug_film_replay) {
        case MI_FILM_REPLAY_AUTO: break;
        case MI_FILM_REPLAY_BLOCKS: ropt.v["MIW_FILM_LEGACY"] = "1"; break;
        case MI_FILM_REPLAY_GROUPS: ropt.v.erase("MIW_FILM_LEGACY"); ropt.v["MIW_FILM_LANES"] = "0"; ropt.v["MIW_FILM_QUADS"] = "0"; ropt.v["MIW_FILM_COLUMNS"] = "0"; break;
        case MI_FILM_REPLAY_COLUMNS: ropt.v.erase("MIW_FILM_LEGACY"); ropt.v["MIW_FILM_LANES"] = "0"; ropt.v["MIW_FILM_QUADS"] = "0"; ropt.v.erase("MIW_FILM_COLUMNS"); ropt.v.erase("MIW_FILM_GROUP"); break;
        case MI_FILM_REPLAY_QUADS: ropt.v.erase("MIW_FILM_LEGACY"); ropt.v["MIW_FILM_LANES"] = "0"; ropt.v.erase("MIW_FILM_COLUMNS"); ropt.v.erase("MIW_FILM_GROUP"); if (ropt.get("MIW_FILM_QUADS") && atoi(ropt.get("MIW_FILM_QUADS")) == 0) ropt.v.erase("MIW_FILM_QUADS"); break;
        case MI_FILM_REPLAY_LANES: ropt.v.erase("MIW_FILM_LEGACY"); ropt.v["MIW_FILM_LANES"] = "1"; break;
        case MI_FILM_REPLAY_LANES_PLAIN_LOG: ropt.v.erase("MIW_FILM_LEGACY"); ropt.v["MIW_FILM_LANES"] = "2"; break;
        default: return fail(c, MI_ERR_INVALID, "render: debug_film_replay must be one of MI_FILM_REPLAY_*");
    }
    if (cfg->debug_tree_width == 4) ropt.v["MIW_BVH8"] = "0";
    else if (cfg->debug_tree_width == 8) ropt.v.erase("MIW_BVH8");
    else if (cfg->debug_tree_width != 0) return fail(c, MI_ERR_INVALID, "render: debug_tree_width must be 0, 4 or 8");
    switch (cfg->debug_path_kernel) {
        case MI_PATH_KERNEL_AUTO: break;
        case MI_PATH_KERNEL_LOCKSTEP: ropt.v["MIW_PHASED"] = "0"; break;
        case MI_PATH_KERNEL_PHASED: ropt.v.erase("MIW_PHASED"); ropt.v["MIW_POOLED"] = "0"; break;
        case MI_PATH_KERNEL_POOLED: ropt.v.erase("MIW_PHASED"); ropt.v["MIW_POOLED"] = "1"; break;
        default: return fail(c, MI_ERR_INVALID, "render: debug_path_kernel must be one of MI_PATH_KERNEL_*");
    }
    RenderParams P;
    mi_status st = fill_params(c, cfg, P);
    if (st != MI_OK) return st;
    const uint32_t bs = (uint32_t) cfg->block_size, bs2 = bs * bs;
    uint32_t bs2_log2 = 0; while ((1u << bs2_log2) < bs2) ++bs2_log2;
    const uint32_t blocks_x = (cfg->crop_w + bs - 1) / bs, blocks_y = (cfg->crop_h + bs - 1) / bs;
    if (!cfg->block_ids || cfg->block_count != blocks_x * blocks_y)
        return fail(c, MI_ERR_INVALID, "render: block_ids must hold %u entries", blocks_x * blocks_y);
    uint32_t n_tiles = cfg->tile_list ? cfg->tile_count : cfg->block_count;
    if (cfg->tile_list)
        for (uint32_t i = 0; i < n_tiles; ++i)
            if (cfg->tile_list[i] >= cfg->block_count) return fail(c, MI_ERR_INVALID, "render: tile_list entry out of range");
    uint64_t n_lanes64 = (uint64_t) n_tiles * bs2;
    if (n_lanes64 >= (1ull << 31)) return fail(c, MI_ERR_INVALID, "render: too many lanes");
    const uint32_t n_lanes = (uint32_t) n_lanes64;
    P.n_lanes = n_lanes;
    const size_t film_n = (size_t) cfg->crop_w * cfg->crop_h * MIW_FILM_CHANNELS;

    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    auto wall0 = std::chrono::steady_clock::now();
    c->cancel.store(0);

    // execution plan
    int plan = cfg->plan;
    if (plan < 0 || plan > 2) return fail(c, MI_ERR_INVALID, "render: plan must be 0, 1 or 2");
    const bool lds_resident = c->lds_cfg.brute || (c->lds_cfg.nodes_staged >= c->view.node_count && c->lds_cfg.tris_staged >= c->view.tri_count);
    // measured on MI355X: the resident plan wins whenever the scene query runs out of LDS
    // (packets / staged tree) or with the LDS-stack walk; the queue plan remains for the stackless fallback
    if (plan == 0) plan = (lds_resident || c->lds_cfg.stack) ? 2 : 1;
#if MIW_SPECTRAL
    if (cfg->plan == 1) return fail(c, MI_ERR_INVALID, "render: the scalar_spectral library runs the resident plan only (plan 0 or 2)");
    plan = 2;
#endif
    const bool direct = P.integrator == INTEG_DIRECT;
    if (direct) {                                                // direct.h runs on the paired query of the resident plan
        if (cfg->plan == 1) return fail(c, MI_ERR_INVALID, "render: the direct integrator runs the resident plan only (plan 0 or 2)");
        plan = 2;
    }
    c->counters.plan = (uint32_t) plan;

    size_t nl = std::max<uint32_t>(n_lanes, 1);
    HIP_TRY(c, c->q_st.resize(nl)); HIP_TRY(c, c->q_pixel.resize(nl));
    if (plan == 1) {
        HIP_TRY(c, c->q_tp.resize(nl)); HIP_TRY(c, c->q_res.resize(nl)); HIP_TRY(c, c->q_ray_o.resize(nl));
        HIP_TRY(c, c->q_ray_d.resize(nl)); HIP_TRY(c, c->q_hit.resize(nl)); HIP_TRY(c, c->q_sh_d.resize(nl));
        HIP_TRY(c, c->q_sh_c.resize(nl)); HIP_TRY(c, c->q_pos.resize(nl)); HIP_TRY(c, c->q_sh_vis.resize(nl));
    }
    HIP_TRY(c, c->d_cnt.resize(MIW_CNT_SHARDS));
    HIP_TRY(c, hipMemsetAsync(c->d_cnt.p, 0, sizeof(Counters) * MIW_CNT_SHARDS, s));

    // film mode: sample log + ordered gather if the log fits, else float64 atomics; the log's format and layout follow the replay kernel
    FilmPlan FP;
    FilmReplay replay;                                            // (filled before the path kernel when the replay is queued beside it)
    { const mi_status fs = plan_film_log(c, cfg, ropt, P, s, plan, n_tiles, n_lanes, bs2, bs2_log2, film_n, FP); if (fs != MI_OK) return fs; }
    const int film_mode = FP.film_mode; const bool rec16 = FP.rec16; const int film_lanes = FP.film_lanes; const uint32_t log_il = FP.log_il;
    HIP_TRY(c, c->d_block_ids.resize(cfg->block_count));
    HIP_TRY(c, hipMemcpyAsync(c->d_block_ids.p, cfg->block_ids, cfg->block_count * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (cfg->tile_list && n_tiles) {
        HIP_TRY(c, c->d_tile_list.resize(n_tiles));
        HIP_TRY(c, hipMemcpyAsync(c->d_tile_list.p, cfg->tile_list, n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }

    LaneQueues Q;
    Q.tp = c->q_tp.p; Q.res = c->q_res.p; Q.st = c->q_st.p; Q.pos = c->q_pos.p; Q.pixel = c->q_pixel.p;
    Q.ray_o = c->q_ray_o.p; Q.ray_d = c->q_ray_d.p; Q.hit = c->q_hit.p;
    Q.sh_d = c->q_sh_d.p; Q.sh_c = c->q_sh_c.p; Q.sh_vis = c->q_sh_vis.p;
    Q.log_pos = film_mode == 1 && !rec16 ? c->q_log_pos.p : nullptr; Q.log_val = film_mode == 1 && !rec16 ? c->q_log_val.p : nullptr;
    Q.lane_cost = nullptr; Q.lane_sorted = nullptr; Q.piece_list = nullptr; Q.simd_ids = nullptr; Q.piece_a = 64u; Q.piece_b = 1u;
    Q.log_rec = film_mode == 1 && rec16 ? c->q_log_rec.p : nullptr; Q.log_thr = rec16 ? c->d_fc_thr.p : nullptr; Q.log_rej = rec16 ? c->classes.count | ((film_mode == 1 ? log_il : 0u) << 8) : 0u;
    // ---- dynamic LDS of the render launches: [staged geometry | per-lane stack][256 phase thresholds (16-byte records)][the scene's
    // small tables + the environment warp's smallest levels (trace.h: stage_tables)]. The kernels that stage the tables keep four
    // workgroups per CU (160 KB / 4), so everything has to fit 40 KB less one allocation granule; a scene whose tables do not takes
    // the lock-step tree kernels, which read them from global memory (tables_fit; mi_bvh_build applies the same bound before it
    // declares a scene tiny). The 8-wide walk's stack is one 8-byte entry per LEVEL of its tree (miw/bvh8.h): a launch that will
    // run it sizes the column by the tree's depth instead of the 4-wide walk's 32 x 4 bytes, and the bytes that frees go to more
    // levels of the environment warp (round 5: the 0.9 M-triangle interior, depth 10 -> 20 KB of stack, warp levels down to 64 x 32).
    struct LdsLayout { TraceLds cfg; size_t rlds, rlds_plain, table_bytes; bool tables_fit; };
    const size_t lds_budget4 = MIW_LDS_PER_WORKGROUP - MIW_LDS_STATIC - MIW_LDS_GRANULE;
    auto lay_out = [&](size_t base, size_t env_cap, size_t lds_budget) {
        LdsLayout L; L.cfg = c->lds_cfg; L.rlds = base;
        L.cfg.thr16 = (uint32_t) ((L.rlds + 15) / 16);
        if (rec16) L.rlds = (size_t) L.cfg.thr16 * 16 + (MIW_FC_TABLE + 3) / 4 * 16;
        L.cfg.tab16 = (uint32_t) ((L.rlds + 15) / 16);
        L.table_bytes = lds_table_bytes(c, L.cfg.tab_words, c->lds_cfg.brute != 0);
        // (packet scenes always stage: mi_bvh_build bounded their tables by 24 KB, which with the packets, boxes and thresholds stays
        // inside the 64 KB a workgroup may ask for — at fewer workgroups per CU past 40 KB)
        L.tables_fit = (size_t) L.cfg.tab16 * 16 + L.table_bytes <= (c->lds_cfg.brute ? 64u * 1024u : lds_budget);
        L.cfg.env_top_count = L.cfg.env_top_base = L.cfg.env_top_words = 0;
        if (L.tables_fit && c->have_env && !(ropt.get("MIW_ENV_TOP") && atoi(ropt.get("MIW_ENV_TOP")) == 0)) {
            // ... and as many of the environment warp's smallest levels as fit what is left (at most env_cap): levels are stored from
            // the largest (0) to the smallest (n_levels - 1), so the top `count` levels are the tail of the array
            const EnvmapRec &e = c->env_host;
            const size_t used = (size_t) L.cfg.tab16 * 16 + L.table_bytes;
            const size_t room = used < lds_budget ? std::min<size_t>(env_cap, lds_budget - used) : 0;
            uint32_t count = 0;
            while (count + 1 < e.n_levels && ((size_t) c->env_levels_total - e.level_offset[e.n_levels - 1 - count]) * 4 <= room) ++count;
            if (count) {
                L.cfg.env_top_count = count; L.cfg.env_top_base = e.level_offset[e.n_levels - count];
                L.cfg.env_top_words = (uint32_t) (c->env_levels_total - L.cfg.env_top_base);
                L.table_bytes += ((size_t) L.cfg.env_top_words + 3) / 4 * 16;
            }
        }
        // (only the kernels that call stage_tables ask for the table bytes: the phase machine and the packet kernels — `rlds`; the
        // lock-step tree kernels, which read the tables from global memory, launch with `rlds_plain`: ADVICE r04)
        L.rlds_plain = L.rlds;
        if (L.tables_fit) L.rlds = (size_t) L.cfg.tab16 * 16 + L.table_bytes;
        return L;
    };
    LdsLayout lay = lay_out(c->lds_bytes, 4096, lds_budget4);
    // will this render walk the 8-wide tree? (the conditions mi_render applies below, known here already; MIW_BVH8=0 keeps the 4-wide walk)
    const bool pre8 = plan == 2 && cfg->integrator != MI_INTEGRATOR_DIRECT && !c->lds_cfg.brute && c->lds_cfg.stack && c->view.nodes4 && c->nodes8_count != 0u &&
                      !(ropt.get("MIW_BVH8") && atoi(ropt.get("MIW_BVH8")) == 0) && !(ropt.get("MIW_PHASED") && atoi(ropt.get("MIW_PHASED")) == 0) &&
                      !(ropt.get("MIW_PHASED_WAVES") && atoi(ropt.get("MIW_PHASED_WAVES")) != 4) && film_mode == 1;
    bool stack8_sized = false;
    if (pre8 && !(ropt.get("MIW_STACK8_FULL") && atoi(ropt.get("MIW_STACK8_FULL")) != 0)) {
        const size_t base8 = (size_t) c->lds_cfg.stack16 * 16 + (size_t) std::max<uint32_t>(c->nodes8_depth, 2u) * MIW_BLOCK * sizeof(U2);
        const LdsLayout lay8 = lay_out(base8, 12288, lds_budget4);
        if (lay8.tables_fit && base8 <= c->lds_bytes) { lay = lay8; stack8_sized = true; }
    }
    // The pooled phase machine (device/pooled_kernel.h, round 6): ONE workgroup of MIW_POOL_NW wavefronts per CU whose lanes share their walks
    // through LDS — per lane an 8-byte stack entry per tree level + an 80-byte job record + a status byte, then thresholds, tables and the
    // environment warp's levels once per CU: [stacks][job records][status][thresholds][tables]. All of the CU's 160 KB may be one workgroup's.
    // MIW_POOLED=0 keeps k_path_phased (A/B runs, and what shards with placed queues run).
    LdsLayout layp{}; bool pooled_fits = false;
    // shape of the workgroup: NW wavefronts x PP pixels per lane — 12 x 1 (three wavefronts per SIMD), or 8 x 2 (two per SIMD, 256 VGPRs, twice the
    // jobs per lane: MIW_POOL_SHAPE=8x2; needs 1024 job records + stacks in LDS: trees of up to 8 levels)
    int pool_nw = 12, pool_pp = 1;
    if (const char *e = ropt.get("MIW_POOL_SHAPE")) { int a = 0, b = 0; if (sscanf(e, "%dx%d", &a, &b) == 2 && ((a == 12 && b == 1) || (a == 8 && b == 2))) { pool_nw = a; pool_pp = b; } }
    // (the 8 x 2 shape is instantiated for the MATS_TRIO class only — BASELINE configs 3 / 4)
    if (!(c->trio && !(ropt.get("MIW_TRIO") && atoi(ropt.get("MIW_TRIO")) == 0) && c->rects.empty() && !c->textured)) { pool_nw = 12; pool_pp = 1; }
    if (pre8 && ropt.get("MIW_POOLED") && atoi(ropt.get("MIW_POOLED")) != 0) {
        const size_t NJ = (size_t) pool_nw * pool_pp * 64u;
        size_t base = (size_t) std::max<uint32_t>(c->nodes8_depth, 2u) * NJ * sizeof(U2);
        const uint32_t pool16 = (uint32_t) (base / 16); base += 5u * NJ * 16u;
        const uint32_t stat16 = (uint32_t) (base / 16); base += NJ;
        layp = lay_out(base, 32768, (size_t) 160u * 1024u - 1024u);
        layp.cfg.stack16 = 0u; layp.cfg.pool16 = pool16; layp.cfg.stat16 = stat16; layp.cfg.nodes_staged = layp.cfg.tris_staged = 0u;
        pooled_fits = layp.tables_fit;
        if (ropt.get("MIW_DEBUG")) fprintf(stderr, "[miwave] pooled phase machine: %d wavefronts per workgroup x %d pixels per lane, LDS %zu bytes (stacks %zu, job records %zu, tables %zu of which environment warp levels %u), fits %d\n",
                                         pool_nw, pool_pp, layp.rlds, (size_t) pool16 * 16, 5u * NJ * 16u, layp.table_bytes, layp.cfg.env_top_words * 4u, (int) pooled_fits);
    }
    TraceLds rcfg = lay.cfg; size_t rlds = lay.rlds; const size_t rlds_plain = lay.rlds_plain, table_bytes = lay.table_bytes; const bool tables_fit = lay.tables_fit;
    if (ropt.get("MIW_DEBUG")) fprintf(stderr, "[miwave] LDS per workgroup: %zu bytes dynamic with the scene tables (%zu without), tables %zu B of which environment warp levels %u B, budget %zu%s\n",
                                     rlds, rlds_plain, table_bytes, rcfg.env_top_words * 4u, lds_budget4, stack8_sized ? "; stack sized for the 8-wide tree's depth" : "");

    mi_counters &K = c->counters;
    K.samples = K.segments = K.shadow_rays = K.iterations = 0; K.lanes = n_lanes;
    K.ms_trace_closest = K.ms_trace_any = K.ms_shade = K.ms_init = K.ms_resolve = K.ms_path = K.ms_film_blocks = K.ms_film_merge = K.ms_film_pack = 0;
    K.n_trace_closest = K.n_trace_any = K.n_shade = K.n_path = 0; K.path_kernel = 0; K.placed = 0; K.tree_width = 0; K.pooled = 0; K.pool_waves = 0; K.film_overlapped = 0; K.film_groups = 0; K.job_chunk = 0; K.job_chunks = 0; c->replay_enqueued = false; K.place_cost_max = K.place_cost_unit = K.place_max_pixel = K.place_measure_spp = 0; K.place_cost_mean = 0.0;

    LaunchTimer T{ c, K, cfg->profile != 0 };                    // HIP-event time per launch class (mi_counters::ms_*)
#define MIW_TIMED(cls_, launch) do {                                                   \
        size_t e0_ = 0, e1_ = 0;                                                       \
        if (T.on) { HIP_TRY(c, T.get_event(e0_)); HIP_TRY(c, hipEventRecord(c->ev_pool[e0_], s)); } \
        launch;                                                                        \
        if (T.on) { HIP_TRY(c, T.get_event(e1_)); HIP_TRY(c, hipEventRecord(c->ev_pool[e1_], s)); \
                    T.stamps.push_back({ cls_, e0_, e1_ }); }                          \
    } while (0)

    mi_status result = MI_OK;
    auto read_counters = [&](Counters &sum) -> mi_status {
        HIP_TRY(c, hipMemcpyAsync(c->h_cnt, c->d_cnt.p, sizeof(Counters) * MIW_CNT_SHARDS, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        sum.segments = sum.samples = sum.shadow_rays = sum.active_lanes = 0;
        for (int i = 0; i < MIW_CNT_SHARDS; ++i) {
            sum.segments += c->h_cnt[i].segments; sum.samples += c->h_cnt[i].samples;
            sum.shadow_rays += c->h_cnt[i].shadow_rays; sum.active_lanes += c->h_cnt[i].active_lanes;
        }
        K.samples = sum.samples; K.segments = sum.segments; K.shadow_rays = sum.shadow_rays;
        return MI_OK;
    };
    auto out_of_time = [&]() {
        return cfg->timeout_s > 0.f &&
               std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > cfg->timeout_s;
    };
    if (n_lanes > 0 && plan == 2) {
        // ---- resident plan: every pixel advances `per_launch` samples per launch ----
        dim3 grid((n_lanes + MIW_BLOCK - 1) / MIW_BLOCK), block(MIW_BLOCK);
        InitArgs A; A.block_ids = c->d_block_ids.p; A.tile_list = cfg->tile_list ? c->d_tile_list.p : nullptr;
        A.blocks_x = blocks_x; A.blocks_y = blocks_y; A.bs = bs; A.bs2_log2 = bs2_log2; A.base_seed = cfg->base_seed;
        MIW_TIMED(3, hipLaunchKernelGGL(k_init_pixels, grid, block, 0, s, P, c->q_st.p, c->q_pixel.p, A));
        HIP_TRY(c, hipGetLastError());
        const uint32_t n_simd = (uint32_t) c->cu_count * 4u;                                  // placed queues: one per SIMD
        HIP_TRY(c, c->d_next_pixel.resize(std::max<uint32_t>(8u, n_simd)));
        const uint32_t per_launch = cfg->samples_per_launch > 0 ? (uint32_t) cfg->samples_per_launch : 128u;
        // Shards of at most one pixel per resident lane (an N-GPU frame at N >= 8): a short measuring launch (the first eighth of the
        // samples, at least 16) records what every 64-lane piece costs, the pieces are dealt to the SIMDs longest-first, and the
        // rest of the samples run with every wavefront taking its pixels from the queue of the SIMD it sits on (resident_kernel.h:
        // QueueWork). Same samples, same log slots: the film does not change. MIW_PLACE = 0 | 1 overrides.
        // (packet kernels only — QueueWork<Placed> — and only where every pixel of the shard can be resident at once: 4 wavefronts per SIMD
        // for the plain-diffuse packet kernel, 3 for the others — a shard larger than that is balanced by the queue itself)
        // Which path kernel runs (decided here because placement depends on it): tree scenes with the LDS-stack walk get the wave-level
        // phase machine (device/phased_kernel.h; MIW_PHASED=0 keeps the lock-step kernel, MIW_TRIO=0 the full BSDF table: A/B runs)
        const bool tiny = c->lds_cfg.brute != 0;
        const bool phased_on = !(ropt.get("MIW_PHASED") && atoi(ropt.get("MIW_PHASED")) == 0);
        const bool trio_on = !(ropt.get("MIW_TRIO") && atoi(ropt.get("MIW_TRIO")) == 0);
        const bool trio_kernel = c->trio && trio_on && c->rects.empty() && !c->textured;   // 52 KB of code instead of 84
        const bool phased = phased_on && !direct && !tiny && c->lds_cfg.stack && (c->view.nodes4 != nullptr || trio_kernel) && (tables_fit || !MIW_LDS_TABLES);
        // 4 waves per SIMD for every tree (measured: 0.9 M triangles +7 - 11 %, 41 k triangles +-0 before the register diet, +8 % after); MIW_PHASED_WAVES = 3 | 4 overrides
        int ph_waves = 4;
        if (const char *e = ropt.get("MIW_PHASED_WAVES")) ph_waves = atoi(e) == 4 ? 4 : 3;
        const bool phased_placeable = phased && c->view.nodes4 != nullptr && ph_waves == 4;     // (the Placed instantiations: the 4-wide tree at four waves per SIMD)
        // the 8-wide tree whenever mi_bvh_build produced one (four waves per SIMD only; MIW_BVH8=0 here keeps the 4-wide walk: A/B runs in one process)
        const bool phased8 = phased_placeable && c->nodes8_count != 0u && !(ropt.get("MIW_BVH8") && atoi(ropt.get("MIW_BVH8")) == 0);
        if (stack8_sized && !phased8) return fail(c, MI_ERR_STATE, "render: the LDS stack was sized for the 8-wide walk, which this launch does not run");
        SceneView view8 = c->view;
        if (phased8) { view8.nodes8 = c->d_nodes8.p; view8.tris = c->d_tris8.p; if (view8.tri_vn) view8.tri_vn = c->d_tri_vn8.p; }
        // wavefronts per SIMD of the path kernel this render launches (resident_kernel.h / phased_kernel.h / trace.h: what each is compiled for)
        // (the plain-diffuse packet kernel of <= 32 triangles: five — or, when the job's pixels fill no more than four wavefronts per SIMD (a rank's shard), the
        // 128-register instantiation for four: resident_kernel.h; MIW_PACKET_SHARD4=0 keeps five)
        bool packet_shard4 = !phased && tiny && c->diffuse_only && !MIW_SPECTRAL && !direct && c->view.tri_count <= 32u && MIW_PACKET_WAVES > 4 && film_mode == 1 && n_lanes <= 64u * 4u * n_simd;
        if (const char *e = ropt.get("MIW_PACKET_SHARD4")) packet_shard4 = packet_shard4 && atoi(e) != 0;
        const uint32_t res_waves = phased ? (uint32_t) ph_waves : !tiny ? (uint32_t) MIW_TREE_WAVES : (c->diffuse_only && !MIW_SPECTRAL ? (c->view.tri_count <= 32u && !packet_shard4 ? (uint32_t) MIW_PACKET_WAVES : 4u) : (uint32_t) MIW_PACKET_WAVES_ALL);
        bool place = film_mode == 1 && !direct && (tiny || phased_placeable) && n_lanes >= 64u * n_simd / 2u && n_lanes <= 64u * res_waves * n_simd &&
                     cfg->spp >= 128u && per_launch >= cfg->spp && cfg->timeout_s <= 0.f;
        if (const char *e = ropt.get("MIW_PLACE")) place = place && atoi(e) != 0;
        uint32_t measure_div = 8u;                                 // the measuring launch runs spp / 8 samples (MIW_PLACE_MEASURE = divisor)
        if (const char *e = ropt.get("MIW_PLACE_MEASURE")) measure_div = (uint32_t) std::max(2, atoi(e));
        const uint32_t measure_end = place ? std::max<uint32_t>(16u, cfg->spp / measure_div) : 0u;
        const uint32_t n_pieces = (n_lanes + 63u) / 64u;
        size_t place_tmp_bytes = 0;
        if (place) {
            HIP_TRY(c, c->d_lane_cost.resize(n_lanes)); HIP_TRY(c, c->d_cost_sorted.resize(n_lanes)); HIP_TRY(c, c->d_lane_iota.resize(n_lanes));
            HIP_TRY(c, c->d_lane_sorted.resize((size_t) n_pieces * 64u));
            HIP_TRY(c, c->d_piece_list.resize((size_t) n_simd * MIW_PLACE_PIECES)); HIP_TRY(c, c->d_simd_ids.resize(1u + (1u << 14)));
            HIP_TRY(c, hipMemsetAsync(c->d_lane_cost.p, 0, n_lanes * sizeof(uint32_t), s));      // (lanes without a pixel never report: cost 0, sorted last)
            HIP_TRY(c, hipMemsetAsync(c->d_lane_sorted.p, 0xff, (size_t) n_pieces * 64u * sizeof(uint32_t), s));
            HIP_TRY(c, hipMemsetAsync(c->d_simd_ids.p, 0, c->d_simd_ids.n * sizeof(uint32_t), s));
            hipLaunchKernelGGL(k_iota, dim3((n_lanes + 255u) / 256u), dim3(256), 0, s, c->d_lane_iota.p, n_lanes);
            HIP_TRY(c, rocprim::radix_sort_pairs_desc(nullptr, place_tmp_bytes, c->d_lane_cost.p, c->d_cost_sorted.p, c->d_lane_iota.p, c->d_lane_sorted.p, (size_t) n_lanes, 0u, 32u, s));
            HIP_TRY(c, c->d_place_tmp.resize(place_tmp_bytes + 16));
        }
        // film_mode 2: workgroup-local float64 tile in LDS (needs 16x16-pixel workgroups: block_size >= 16)
        TileArgs TA; memset(&TA, 0, sizeof TA);
        size_t tile_bytes = 0;
        if (film_mode == 2 && bs >= 16) {
            TA.tile_list = A.tile_list; TA.blocks_x = blocks_x; TA.bs = bs; TA.bs2_log2 = bs2_log2;
            TA.side = 16u + 2u * std::max<uint32_t>((uint32_t) cfg->filter_border, (uint32_t) floorf(cfg->filter_radius + .5f));
            TA.geom16 = (uint32_t) ((c->lds_bytes + 15) / 16);
            tile_bytes = (size_t) TA.geom16 * 16 - c->lds_bytes + (size_t) TA.side * TA.side * MIW_FILM_CHANNELS * sizeof(double);
            if (c->lds_bytes + tile_bytes > 64 * 1024) {
                HIP_TRY(c, hipFuncSetAttribute((const void *) k_path_resident<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (c->lds_bytes + tile_bytes)));
                HIP_TRY(c, hipFuncSetAttribute((const void *) k_path_resident<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (c->lds_bytes + tile_bytes)));
                HIP_TRY(c, hipFuncSetAttribute((const void *) k_path_resident<false, 1, MATS_ALL, false, INTEG_DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (c->lds_bytes + tile_bytes)));
                HIP_TRY(c, hipFuncSetAttribute((const void *) k_path_resident<false, 0, MATS_ALL, true, INTEG_DIRECT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) (c->lds_bytes + tile_bytes)));
            }
        }
        const uint32_t sync_every = 8;
        uint32_t launches = 0;
        bool place_stats_due = false;                              // (profile runs) the measuring launch's statistics, fetched after the last launch
        // the film replay beside the render (overlap_prepare / overlap_enqueue above): frames replayed by k_film_lanes over the tile-interleaved log whose
        // samples all run in ONE launch (a pixel is complete when that launch stores it), no placement, no time limit
        // OPT-IN (MIW_FILM_OVERLAP=1): measured at C2 (gpurun r6k - r6m, profiles/r06_experiments.txt) it hides up to 3/4 of the replay and gains 0 - 1 % of the
        // frame — the wave slots it needs cost the path kernel in proportion (32 workgroups short: +6 ms), a replay wavefront beside three packet-kernel
        // wavefronts takes 20 - 30 ms instead of 5, and what is left after the path kernel cannot take less than one wavefront's chain (~8 - 10 ms)
        int overlap_mode = 0;
        if (const char *e = ropt.get("MIW_FILM_OVERLAP")) overlap_mode = atoi(e);
        const bool overlap = overlap_mode != 0 && c->overlap_state != 0 && film_mode == 1 && FP.rec16 && FP.film_lanes != 0 && FP.log_il != 0 && !place && !phased /* (QueueWork<..., Groups = false>) */ && tiny && c->diffuse_only && !MIW_SPECTRAL && !direct /* the 120-register packet kernel: the only one a replay wavefront fits beside */ &&
                             per_launch >= cfg->spp && cfg->timeout_s <= 0.f && n_tiles > 0;
        for (uint32_t done = 0; done < cfg->spp; ) {
            uint32_t end = cfg->spp - done < per_launch ? cfg->spp : done + per_launch;
            if (film_mode == 1) {
                // persistent grid: <= 4 workgroups per CU, fed from the shared pixel queue
                HIP_TRY(c, hipMemsetAsync(c->d_next_pixel.p, 0, c->d_next_pixel.n * sizeof(uint32_t), s));
                rcfg.queues = 1u; Q.lane_cost = nullptr; Q.lane_sorted = nullptr; Q.piece_list = nullptr; Q.simd_ids = nullptr;
                if (overlap && done == 0) {                          // the replay's arguments and the per-group counters, in front of the path kernel
                    mi_status os = film_replay_setup(c, cfg, s, FP, n_tiles, bs, bs2_log2, blocks_x, blocks_y, replay);
                    if (os == MI_OK) os = overlap_prepare(c, cfg, s, n_tiles, bs, blocks_x, Q, bs2_log2);
                    if (os != MI_OK) return os;
                    if (Q.group_done) HIP_TRY(c, hipEventRecord(c->ev_fork, s));
                }
                // chunk jobs (resident_kernel.h: QueueWork::fetch): full frames of the packet kernels, all samples in one launch. Without them a frame of N pixels
                // on L resident lanes lasts ceil(N / L) rounds of one PIXEL (every lane starts at once, a pixel's samples are one serial job, the jobs are about
                // equally long): C2 = 6.33 -> 7; with chunks of spp / 8 samples ceil(8 N / L) / 8 = 6.375.
                Q.job_chunk = Q.job_pow = Q.job_total = 0u; Q.job_min = 0x80000000u;
                // Every job costs one atomic on ONE address (the queue's counter) and a hand-over (a 16-byte state word out and in, past the L2): uniform chunks of
                // 8 / 16 / 32 samples ran C2 in 702 / 496 / 305 ms, 64 / 128 / 256 in 226.8 / 226.5 / 230.6, whole pixels in 239.3 (gpurun r6p). So the chunks
                // HALVE — a chunk ends where the samples still to do are a power of two — down to job_min samples (512 spp: 256 + 128 + 64 + 64): four hand-overs
                // per pixel instead of eight, the tail as long as the smallest. The phase machine's full frames (C3, C4) take the same queue.
                if ((tiny || (phased && MIW_PHASED_JOBS != 0)) && !pooled_fits && !place && !overlap && done == 0 && per_launch >= cfg->spp && cfg->timeout_s <= 0.f) {
                    uint32_t jmin = tiny ? 64u : 32u;                                 // (gpurun r6s, halving chunks: C2 4 461 / 4 316 / 4 389 Msamples/s at 64 / 32 / 128; C3 1 298 / 1 305, C4 522 / 528 at 64 / 32)
                    if (const char *e = ropt.get("MIW_JOB_CHUNK")) { const uint32_t v = (uint32_t) std::max(0, atoi(e)); jmin = (v & (v - 1u)) ? 0u : v; }
                    // from 1.2 pixels per resident lane on (gpurun r6z, a rank's shard of C2 with / without: 4 ranks = 1.6 pixels per lane 67.8 / 79.1 ms, 5 ranks = 1.3: 58.7 / 67.5,
                    // 6 ranks = 1.06: 54.9 / 54.6; at about ONE pixel per lane every pixel is in flight all the time and no lane finds a second job: 8 ranks 47.1 against the placed launches' 41.4)
                    bool enough = (uint64_t) n_lanes * 5u >= (uint64_t) 6u * 64u * res_waves * n_simd;
                    if (const char *e = ropt.get("MIW_JOB_CHUNK_FORCE")) enough = enough || atoi(e) != 0;
                    uint32_t pow2 = 1u;
                    while (pow2 * 2u < cfg->spp) pow2 *= 2u;                          // the largest power of two below spp
                    uint64_t chunks = 1u;
                    for (uint32_t r = pow2; jmin && r >= jmin && r < cfg->spp; r >>= 1) ++chunks;
                    if (jmin && enough && chunks >= 2u && (uint64_t) n_lanes * chunks < (1ull << 31)) {
                        Q.job_chunk = jmin; Q.job_min = jmin; Q.job_pow = pow2; Q.job_total = (uint32_t) ((uint64_t) n_lanes * chunks);
                        K.job_chunk = jmin; K.job_chunks = (uint32_t) chunks;
                    }
                }
                if (place && done == 0) { end = measure_end; Q.lane_cost = c->d_lane_cost.p; Q.simd_ids = c->d_simd_ids.p; }   // the measuring launch
                else if (place) {
                    // the lanes by what their pixel cost, dearest first (device radix sort), cut into pieces of 64: consecutive sorted lanes
                    // (the pixels of a wavefront cost about the same and finish together), or every pieces-th one (one pixel of every cost
                    // stratum per wavefront). Either way a piece's cost is that of its first (dearest) lane.
                    HIP_TRY(c, rocprim::radix_sort_pairs_desc(c->d_place_tmp.p, place_tmp_bytes, c->d_lane_cost.p, c->d_cost_sorted.p, c->d_lane_iota.p, c->d_lane_sorted.p, (size_t) n_lanes, 0u, 32u, s));
                    // Which cut: measured on rank 0's 1/8 shard (gpurun r4b, path kernel ms at 256 / 128 / 512 spp; no placement -> consecutive
                    // -> spread): material balls 99.3 -> 90.7 -> 104.9, 0.9 M-triangle interior 130.8 -> 131.0 -> 120.9, Cornell packets
                    // 40.0 -> 36.7 -> 44.3. Consecutive pieces keep the dear pixels in few wavefronts, which the priorities then favour; the
                    // interior — every pixel dear, its walks bound by memory latency, tree and triangles beyond the L2s — gains from every
                    // wavefront carrying the same mix. Rule: spread when the tree does not fit the aggregate L2 (32 MB). MIW_PLACE_SPREAD = 0 | 1 overrides.
                    bool spread = phased && ((phased8 ? (size_t) c->nodes8_count * sizeof(Bvh8Node) : (size_t) c->nodes4_count * sizeof(Bvh4Node)) + (size_t) c->view.tri_count * sizeof(Tri)) > ((size_t) 32 << 20);
                    if (const char *e = ropt.get("MIW_PLACE_SPREAD")) spread = atoi(e) != 0;
                    Q.piece_a = spread ? 1u : 64u; Q.piece_b = spread ? n_pieces : 1u;
                    // one host round trip between the measuring launch and the placed one, and only for the SIMD registry: the pieces' costs stay on the
                    // device (the dealing below needs their ORDER only, which the sort fixed; round 6 — the statistics of the measuring launch,
                    // mi_counters::place_*, are read after the frame's last launch, below)
                    // the SIMDs the measuring launch ran on, numbered 0 .. nqueues - 1 (simd_ids[1 + hardware key]; word 0 = the "all dry" flag)
                    std::vector<uint32_t> &ids = c->h_simd_ids;
                    ids.resize(c->d_simd_ids.n);
                    HIP_TRY(c, hipMemcpyAsync(ids.data(), c->d_simd_ids.p, ids.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                    HIP_TRY(c, hipStreamSynchronize(s));
                    place_stats_due = cfg->profile != 0;
                    uint32_t nqueues = 0;
                    for (size_t k = 1; k < ids.size(); ++k) ids[k] = ids[k] ? nqueues++ : 0xffffffffu;
                    ids[0] = 0u;
                    if (nqueues < 2u || (uint64_t) nqueues * MIW_PLACE_PIECES < n_pieces) nqueues = 0;   // (cannot happen on a whole device; then: one queue)
                    if (nqueues) {
                    // The pieces arrive in descending cost (they are cuts of the sorted lane list), so "longest piece first onto the emptiest queue" is dealt
                    // in rounds, alternating direction (round 0: piece i to queue i; round 1: piece N + i to queue N - 1 - i; ...): every queue gets one
                    // piece of every cost stratum, dear and cheap strata paired. O(pieces) on the host between the two launches — the heap-based
                    // longest-first dealing of rounds 3 - 5 took 0.3 ms there (4 096 pieces, 1 024 queues) with the GPU idle; any dealing renders the same film.
                    std::vector<uint32_t> &list = c->h_piece_list;                                          // (context-owned: the upload below is not waited for)
                    list.assign((size_t) nqueues * MIW_PLACE_PIECES, 0xffffffffu);
                    for (uint32_t i = 0; i < n_pieces; ++i) {
                        const uint32_t round = i / nqueues, k = i % nqueues, qd = (round & 1u) ? nqueues - 1u - k : k;
                        list[(size_t) qd * MIW_PLACE_PIECES + round] = i;
                    }
                    HIP_TRY(c, c->d_piece_list.resize(list.size()));
                    HIP_TRY(c, hipMemcpyAsync(c->d_piece_list.p, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
                    HIP_TRY(c, hipMemcpyAsync(c->d_simd_ids.p, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
                    rcfg.queues = nqueues; Q.piece_list = c->d_piece_list.p; Q.lane_sorted = c->d_lane_sorted.p; Q.simd_ids = c->d_simd_ids.p;
                    K.placed = 1u;
                    }
                }
                // persistent grid: as many workgroups per CU as the kernel is compiled for wavefronts per SIMD — the plain-diffuse packet kernel FIVE since
                // round 6 (96 registers: 248.1 -> 238.3 ms at C2, resident_kernel.h), the other scalar_rgb packet kernels four. (Round 2 launched 3 per CU for shards of about one pixel per resident lane; with the
                // least-progress-first priorities below every pixel of such a shard should start at once: 42.3 vs 49.6 ms on the 1/8
                // shard of C2, profiles/r03.)
                unsigned wg_per_cu = std::max(4u, direct ? 4u : (unsigned) res_waves);   // (never fewer than four: a kernel compiled for three keeps a fourth workgroup queued behind them)
                if (const char *e = ropt.get("MIW_WG_PER_CU")) wg_per_cu = (unsigned) std::max(1, atoi(e));
                unsigned room = 0u;                                // wave slots kept free for the replay that runs beside this launch (overlap_enqueue)
                if (Q.group_done) { room = 32u; if (const char *e = ropt.get("MIW_FILM_OVERLAP_ROOM")) room = (unsigned) std::max(0, atoi(e)); }
                // (room is counted from what is RESIDENT — the kernels compiled for three wavefronts per SIMD get a fourth workgroup per CU queued behind the
                // first three, which would take every slot a shorter grid leaves)
                const unsigned resident = (unsigned) c->cu_count * (direct ? (unsigned) MIW_DIRECT_WAVES : (unsigned) res_waves);
                const unsigned full = room ? std::min((unsigned) c->cu_count * wg_per_cu, resident) : (unsigned) c->cu_count * wg_per_cu;
                const dim3 pgrid(std::min<unsigned>(grid.x, full > 2u * room ? full - room : full));
#define MIW_PATH_LAUNCH(T, M) MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<true, T, M, false>), pgrid, block, (T) != 0 ? rlds : rlds_plain, s, P, c->view, Q, (double *) nullptr, c->d_cnt.p, rcfg, end, TA, c->d_next_pixel.p))
                // kernel variants: no BSDF dispatch when every shape is plain diffuse (64-bit candidate masks: that
                // variant fits 4 waves per SIMD without spills); else 32-bit candidate masks up to 32 triangles
                // (`phased`, `trio_kernel`, `ph_waves`: decided above the loop; MIW_BVH4=0 keeps the BVH2 node body for the MATS_TRIO class: A/B runs)
                K.path_kernel = phased ? (c->view.nodes4 ? 1u : 3u) : 0u;
                const dim3 phgrid(std::min<unsigned>(grid.x, (unsigned) c->cu_count * (unsigned) ph_waves));
                // the shade vote (phased_kernel.h): shade once n_shade * num >= den * (lanes of the busier walk body). Measured on the
                // 4-wide tree (gpurun r2f / r2g, Msamples/s at 1 : 1 -> 3 : 2 -> 2 : 1): balls 844 -> 873 -> 860; interior with its
                // environment-map lookups 338 -> 348 -> 361. MIW_SHADE_VOTE=num:den overrides (A/B runs).
                // (per-XCD pixel queues — each XCD's workgroups taking their pixels from their own eighth of the image, so that the rays of
                // one L2 stay in one region — measured neutral on the material balls, 883 vs 879 Msamples/s, and -2 % on the interior,
                // 362 vs 370: the secondary rays of a closed room go everywhere. Removed again.)
                // (rcfg.queues: 1, or the SIMD count for the placed launch of a small shard, set above)
                // least-progress-first wave priorities (QueueWork::tick): the wavefronts of a SIMD finish together instead of the most
                // expensive one running on alone. Made for shards of about one pixel per resident lane (1/8 of C2: path kernel 50.1 ->
                // 42.3 ms; 1/8 of the material balls -14 %), it also shortens the drain of a full frame (C2 275.9 -> 268.7 ms), so it is
                // always on. MIW_TAIL_PRIO = 0 | 1 overrides.
                rcfg.tail_prio = 1u;
                if (const char *e = ropt.get("MIW_TAIL_PRIO")) rcfg.tail_prio = atoi(e) ? 1u : 0u;
                TraceLds ph_cfg = rcfg;
                ph_cfg.shade_num = 2; ph_cfg.shade_den = c->have_env ? 4 : 3;
                if (const char *e = ropt.get("MIW_SHADE_VOTE")) { int a = 0, b = 0; if (sscanf(e, "%d:%d", &a, &b) == 2 && a > 0 && b > 0) { ph_cfg.shade_num = (uint32_t) a; ph_cfg.shade_den = (uint32_t) b; } }
                // when the walk loops hand over (phased_kernel.h): a loop runs on until its lanes are outnumbered 2 : 1 by the other walk body's, or
                // node_exit : 1 / tri_exit : 1 by the lanes waiting for another body. Measured on the 8-wide walk (gpurun r5f - r5h, Msamples/s,
                // material balls / interior): rounds 3 - 4's rule (1 : 1 against the other body, waiting + half the lanes that left) 1 119 / 463;
                // 2 : 1 with node 1, triangle 1 -> 1 161 / 471; triangle 2 -> 1 174 / 472; node 2 -> 1 147 / 478; node 2, triangle 3 -> . / 481
                // (the interior's shade runs are long: loops that wait for them less often win there). MIW_LOOP_EXIT=node:triangle overrides.
                ph_cfg.node_exit = c->have_env ? 2u : 1u; ph_cfg.tri_exit = c->have_env ? 3u : 2u;
                if (const char *e = ropt.get("MIW_LOOP_EXIT")) { int a = 0, b = 0; if (sscanf(e, "%d:%d", &a, &b) == 2 && a >= 0 && b >= 0) { ph_cfg.node_exit = (uint32_t) a; ph_cfg.tri_exit = (uint32_t) b; } }
#define MIW_PHASED_LAUNCH_(M, A, WV, W, PL) MIW_TIMED(6, hipLaunchKernelGGL((k_path_phased<M, A, MIW_PHASE_SPEC != 0, WV, W, PL>), phgrid, block, rlds, s, P, (W) == 2 ? view8 : c->view, Q, c->d_cnt.p, ph_cfg, end, c->d_next_pixel.p))
#define MIW_PHASED_LAUNCH(M, A, W) do { if (ph_waves == 4) MIW_PHASED_LAUNCH_(M, A, 4, W, false); else MIW_PHASED_LAUNCH_(M, A, 3, W, false); } while (0)
                K.tree_width = phased ? (phased8 ? 8u : (c->view.nodes4 ? 4u : 2u)) : 0u;
#if MIW_SPECTRAL
                const bool pooled = false;
                if (pooled) { }
#else
                const bool pooled = phased8 && !place && pooled_fits;
                if (pooled) {
                    // one workgroup per CU; the vote's constants travel in the fields the old kernel's vote used (pooled_kernel.h: shade_min, walk_min, node_min, tri_min)
                    // + pool_claim_min; MIW_POOL_VOTE=shade_min:walk_min:node_min:tri_min:claim_min overrides (A/B runs)
                    TraceLds pcfg = layp.cfg; pcfg.queues = 1u; pcfg.tail_prio = ph_cfg.tail_prio;
                    pcfg.shade_num = 48; pcfg.shade_den = 16; pcfg.node_exit = 40u; pcfg.tri_exit = 20u; pcfg.pool_claim_min = 8u;
                    if (const char *e = ropt.get("MIW_POOL_VOTE")) { int a = 0, b = 0, n = 0, t = 0, m = 0; if (sscanf(e, "%d:%d:%d:%d:%d", &a, &b, &n, &t, &m) == 5 && a > 0 && b >= 0) { pcfg.shade_num = (uint32_t) a; pcfg.shade_den = (uint32_t) b; pcfg.node_exit = (uint32_t) n; pcfg.tri_exit = (uint32_t) t; pcfg.pool_claim_min = (uint32_t) std::max(m, 1); } }
                    const unsigned NL = (unsigned) pool_nw * 64u;
                    const dim3 qgrid(std::min<unsigned>((n_lanes + NL - 1u) / NL, (unsigned) c->cu_count)), qblock(NL);
                    SceneView pview = view8;
#define MIW_POOLED_LAUNCH_(M, A, NW_, PP_) do { HIP_TRY(c, hipFuncSetAttribute((const void *) k_path_pooled<M, A, NW_, PP_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) layp.rlds)); \
                                     MIW_TIMED(6, hipLaunchKernelGGL((k_path_pooled<M, A, NW_, PP_>), qgrid, qblock, layp.rlds, s, P, pview, Q, c->d_cnt.p, pcfg, end, c->d_next_pixel.p)); } while (0)
#define MIW_POOLED_LAUNCH(M, A) do { if (pool_pp == 2) MIW_POOLED_LAUNCH_(M, A, 8, 2); else MIW_POOLED_LAUNCH_(M, A, 12, 1); } while (0)
                    if (c->textured) MIW_POOLED_LAUNCH_(MATS_ALL, true, 12, 1);
                    else if (trio_kernel) MIW_POOLED_LAUNCH(MATS_TRIO, false);
                    else if (c->rects.empty()) MIW_POOLED_LAUNCH_(MATS_PLAIN, false, 12, 1);
                    else MIW_POOLED_LAUNCH_(MATS_PLAIN, true, 12, 1);
#undef MIW_POOLED_LAUNCH
#undef MIW_POOLED_LAUNCH_
                    K.pooled = 1u; K.pool_waves = (uint32_t) pool_nw;
                }
#endif
                else
                if (phased8 && place) {
                    if (c->textured) MIW_PHASED_LAUNCH_(MATS_ALL, true, 4, 2, true);
                    else if (trio_kernel) MIW_PHASED_LAUNCH_(MATS_TRIO, false, 4, 2, true);
                    else if (c->rects.empty()) MIW_PHASED_LAUNCH_(MATS_PLAIN, false, 4, 2, true);
                    else MIW_PHASED_LAUNCH_(MATS_PLAIN, true, 4, 2, true);
                } else if (phased8) {
                    if (c->textured) MIW_PHASED_LAUNCH_(MATS_ALL, true, 4, 2, false);
                    else if (trio_kernel) MIW_PHASED_LAUNCH_(MATS_TRIO, false, 4, 2, false);
                    else if (c->rects.empty()) MIW_PHASED_LAUNCH_(MATS_PLAIN, false, 4, 2, false);
                    else MIW_PHASED_LAUNCH_(MATS_PLAIN, true, 4, 2, false);
                } else if (phased && place) {                           // a shard of about one pixel per resident lane: the Placed instantiations (measuring, then placed launch)
                    if (c->textured) MIW_PHASED_LAUNCH_(MATS_ALL, true, 4, 1, true);
                    else if (trio_kernel) MIW_PHASED_LAUNCH_(MATS_TRIO, false, 4, 1, true);
                    else if (c->rects.empty()) MIW_PHASED_LAUNCH_(MATS_PLAIN, false, 4, 1, true);
                    else MIW_PHASED_LAUNCH_(MATS_PLAIN, true, 4, 1, true);
                } else if (phased) {
                    if (!c->view.nodes4) MIW_PHASED_LAUNCH(MATS_TRIO, false, 0);
                    else if (c->textured) MIW_PHASED_LAUNCH(MATS_ALL, true, 1);
                    else if (trio_kernel) MIW_PHASED_LAUNCH(MATS_TRIO, false, 1);
                    else if (c->rects.empty()) MIW_PHASED_LAUNCH(MATS_PLAIN, false, 1);
                    else MIW_PHASED_LAUNCH(MATS_PLAIN, true, 1);
                }
#undef MIW_PHASED_LAUNCH
#undef MIW_PHASED_LAUNCH_
                else if (direct) {
#define MIW_DIRECT_LAUNCH(T, M, A) MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<true, T, M, A, INTEG_DIRECT>), pgrid, block, (T) != 0 ? rlds : rlds_plain, s, P, c->view, Q, (double *) nullptr, c->d_cnt.p, rcfg, end, TA, c->d_next_pixel.p))
                    if (tiny && c->textured) MIW_DIRECT_LAUNCH(1, MATS_ALL, false);
                    else if (tiny) MIW_DIRECT_LAUNCH(1, MATS_PLAIN, false);
                    else if (c->textured) MIW_DIRECT_LAUNCH(0, MATS_ALL, true);
                    else MIW_DIRECT_LAUNCH(0, MATS_PLAIN, true);
#undef MIW_DIRECT_LAUNCH
                }
                else if (Q.group_done && c->view.tri_count <= 32u)     // (overlap: tiny && diffuse_only) the instantiation that counts finished pixels per group of 64 tiles
                    MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<true, 2, MATS_DIFFUSE, false, INTEG_PATH, true>), pgrid, block, rlds, s, P, c->view, Q, (double *) nullptr, c->d_cnt.p, rcfg, end, TA, c->d_next_pixel.p));
                else if (Q.group_done)
                    MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<true, 1, MATS_DIFFUSE, false, INTEG_PATH, true>), pgrid, block, rlds, s, P, c->view, Q, (double *) nullptr, c->d_cnt.p, rcfg, end, TA, c->d_next_pixel.p));
                else if (packet_shard4 && !Q.group_done)
                    MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<true, 2, MATS_DIFFUSE, false, INTEG_PATH, false, 4>), pgrid, block, rlds, s, P, c->view, Q, (double *) nullptr, c->d_cnt.p, rcfg, end, TA, c->d_next_pixel.p));
                else if (tiny && c->diffuse_only && c->view.tri_count <= 32u) MIW_PATH_LAUNCH(2, MATS_DIFFUSE);   // 32-bit candidate masks (BASELINE config 2: 32 triangles)
                else if (tiny && c->diffuse_only) MIW_PATH_LAUNCH(1, MATS_DIFFUSE);
                else if (tiny && c->textured) MIW_PATH_LAUNCH(1, MATS_ALL);           // texture coordinates / bitmap lookups compiled in
                else if (!tiny && c->textured) MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<true, 0, MATS_ALL, true>), pgrid, block, rlds_plain, s, P, c->view, Q, (double *) nullptr, c->d_cnt.p, rcfg, end, TA, c->d_next_pixel.p));
                else if (tiny && c->view.tri_count <= 32u) MIW_PATH_LAUNCH(2, MATS_PLAIN);
                else if (tiny) MIW_PATH_LAUNCH(1, MATS_PLAIN);
                else if (c->rects.empty()) MIW_PATH_LAUNCH(0, MATS_PLAIN);
                else MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<true, 0, MATS_PLAIN, true>), pgrid, block, rlds_plain, s, P, c->view, Q, (double *) nullptr, c->d_cnt.p, rcfg, end, TA, c->d_next_pixel.p));
#undef MIW_PATH_LAUNCH
            } else if (direct && tiny)
                MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<false, 1, MATS_ALL, false, INTEG_DIRECT>), grid, block, c->lds_bytes + tile_bytes, s, P, c->view, Q, c->d_accum.p, c->d_cnt.p, c->lds_cfg, end, TA, (uint32_t *) nullptr));
            else if (direct)
                MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<false, 0, MATS_ALL, true, INTEG_DIRECT>), grid, block, c->lds_bytes + tile_bytes, s, P, c->view, Q, c->d_accum.p, c->d_cnt.p, c->lds_cfg, end, TA, (uint32_t *) nullptr));
            else if (tiny)
                MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<false, 1>), grid, block, c->lds_bytes + tile_bytes, s, P, c->view, Q, c->d_accum.p, c->d_cnt.p, c->lds_cfg, end, TA, (uint32_t *) nullptr));
            else
                MIW_TIMED(6, hipLaunchKernelGGL((k_path_resident<false, 0>), grid, block, c->lds_bytes + tile_bytes, s, P, c->view, Q, c->d_accum.p, c->d_cnt.p, c->lds_cfg, end, TA, (uint32_t *) nullptr));
            if (Q.group_done) {                                      // the path kernel is in `s`: queue the replay beside it
                const mi_status os = overlap_enqueue(c, ropt, P, s, T, replay, n_tiles, overlap_mode != 2);
                if (os != MI_OK) return os;
                if (c->replay_enqueued) { K.film_overlapped = 1u; K.film_groups = c->replay_launches; }
                Q.group_done = nullptr; Q.group_expected = nullptr; Q.group_flag = nullptr;
            }
            K.n_path++; K.iterations++;
            done = end;
            // cancel() / timeout take effect at launch granularity (the reference checks should_stop() per block)
            if ((++launches % sync_every == 0 || cfg->timeout_s > 0.f || c->cancel.load()) && done < cfg->spp) {
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipStreamSynchronize(s));
                T.drain();
                if (c->cancel.load() || out_of_time()) { result = MI_ERR_CANCELLED; break; }
            }
        }
        HIP_TRY(c, hipGetLastError());
        Counters sum;
        mi_status rs = read_counters(sum);
        if (rs != MI_OK) return rs;
        if (place_stats_due) {             // what the measuring launch found (mi_counters::place_*): the shard's dearest pixel and the mean
            std::vector<uint32_t> all(n_lanes); uint32_t first_lane = 0, px = 0;
            HIP_TRY(c, hipMemcpyAsync(all.data(), c->d_cost_sorted.p, (size_t) n_lanes * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(&first_lane, c->d_lane_sorted.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            if (first_lane < n_lanes) HIP_TRY(c, hipMemcpy(&px, c->q_pixel.p + first_lane, sizeof(uint32_t), hipMemcpyDeviceToHost));
            double csum = 0.0; uint32_t with = 0;
            for (uint32_t v : all) if (v) { csum += (double) v; ++with; }
            K.place_cost_max = all.empty() ? 0u : all[0]; K.place_cost_mean = with ? csum / with : 0.0; K.place_max_pixel = px;
            K.place_cost_unit = phased ? 1u : 0u; K.place_measure_spp = measure_end;
        }
        if (K.placed && ropt.get("MIW_DEBUG")) {                     // how the placed launch went: SIMDs that registered, lanes handed out per queue
            std::vector<uint32_t> cur(n_simd), ids(1u + (1u << 14)), cost(n_pieces);
            (void) hipMemcpy(cur.data(), c->d_next_pixel.p, n_simd * sizeof(uint32_t), hipMemcpyDeviceToHost);
            (void) hipMemcpy(ids.data(), c->d_simd_ids.p, ids.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
            (void) hipMemcpy2D(cost.data(), sizeof(uint32_t), c->d_cost_sorted.p, Q.piece_a * sizeof(uint32_t), sizeof(uint32_t), n_pieces, hipMemcpyDeviceToHost);
            uint32_t keys = 0, key_or = 0, full = 0, over = 0; uint64_t csum = 0; uint32_t cmin = ~0u, cmax = 0;
            for (uint32_t k = 0; k < (1u << 14); ++k) if (ids[1 + k] != 0xffffffffu) { ++keys; key_or |= k; }
            ids[0] = keys;
            for (uint32_t v : cur) { full += v >= 64u * MIW_PLACE_PIECES; over += v > 64u * MIW_PLACE_PIECES; }
            for (uint32_t v : cost) { csum += v; cmin = std::min(cmin, v); cmax = std::max(cmax, v); }
            fprintf(stderr, "[miwave] placed queues: %u SIMDs numbered, %u distinct hardware keys (bits used 0x%x), %u of %u queues drained (%u asked beyond their end), "
                            "cost of the dearest pixel of a piece min / mean / max = %u / %.0f / %u (iterations, or 256-cycle units)\n", ids[0], keys, key_or, full, n_simd, over, cmin, (double) csum / n_pieces, cmax);
        }
        print_debug_statistics(c, ropt, K);                      // (debug builds only: -DMIW_PHASE_STATS / _WALK_STATS / _SECTION_PROFILE / _VERIFY_FILTER)
        T.drain();
    }
#if !MIW_SPECTRAL
    else if (n_lanes > 0) {
        dim3 grid((n_lanes + MIW_BLOCK - 1) / MIW_BLOCK), block(MIW_BLOCK);
        InitArgs A; A.block_ids = c->d_block_ids.p; A.tile_list = cfg->tile_list ? c->d_tile_list.p : nullptr;
        A.blocks_x = blocks_x; A.blocks_y = blocks_y; A.bs = bs; A.bs2_log2 = bs2_log2; A.base_seed = cfg->base_seed;
        // work lists (stream compaction + material sorting), double-buffered by iteration parity
        const size_t n_wg = grid.x, seg_lanes = n_wg * MIW_BLOCK;
        HIP_TRY(c, c->d_lists.resize((size_t) 2 * WL_LISTS * seg_lanes));
        HIP_TRY(c, c->d_list_counts.resize(2 * WL_LISTS * n_wg));
        HIP_TRY(c, hipMemsetAsync(c->d_list_counts.p, 0, 2 * WL_LISTS * n_wg * sizeof(uint32_t), s));
        WorkLists WL[2];
        for (int p = 0; p < 2; ++p) {
            for (int l = 0; l < WL_LISTS; ++l) WL[p].list[l] = c->d_lists.p + ((size_t) p * WL_LISTS + l) * seg_lanes;
            WL[p].count = c->d_list_counts.p + (size_t) p * WL_LISTS * n_wg;
        }
        MIW_TIMED(3, hipLaunchKernelGGL(k_init_lanes, grid, block, 0, s, P, Q, c->q_pixel.p, A, WL[0]));
        HIP_TRY(c, hipGetLastError());

        // trees walked with the LDS stack: one persistent stream kernel serves the E and the S rays of an iteration
        // (device/stream_trace.h); MIW_STREAM=0 keeps the slice-per-workgroup kernels (A/B runs)
        const bool stream_on = !(ropt.get("MIW_STREAM") && atoi(ropt.get("MIW_STREAM")) == 0);
        const bool stream = stream_on && c->lds_cfg.stack && !c->lds_cfg.brute && c->lds_cfg.nodes_staged == 0;
        K.path_kernel = stream ? 2u : 0u;
        // round 6: the stream kernel walks the 8-wide tree (the phase machine's two walk bodies) wherever mi_bvh_build produced it; its hit records then
        // name triangles in that tree's order, so k_sort_hits and k_shade get the view with the triangles / vertex normals of that order
        const bool stream8 = stream && c->view.nodes4 != nullptr && c->nodes8_count != 0u && !(ropt.get("MIW_BVH8") && atoi(ropt.get("MIW_BVH8")) == 0);
        SceneView pview = c->view;
        if (stream8) { pview.nodes8 = c->d_nodes8.p; pview.tris = c->d_tris8.p; if (pview.tri_vn) pview.tri_vn = c->d_tri_vn8.p; }
        K.tree_width = stream8 ? 8u : 0u;
        if (stream) {
            HIP_TRY(c, c->d_next_pixel.resize(1));
            HIP_TRY(c, hipMemsetAsync(c->d_next_pixel.p, 0, sizeof(uint32_t), s));
        }
        const dim3 sgrid(std::min<unsigned>(grid.x, (unsigned) c->cu_count * (stream8 ? 4u : (unsigned) MIW_STREAM_WAVES)));
        const int check_every = 16;
        bool first = true;
        unsigned long long active_prev = 0;
        uint32_t parity = 0;
        for (;;) {
            for (int it = 0; it < check_every; ++it, parity ^= 1u) {
                const WorkLists &cur = WL[parity], &nxt = WL[parity ^ 1u];
                if (stream) {
                    if (stream8) MIW_TIMED(0, hipLaunchKernelGGL(k_trace_stream<2>, sgrid, block, c->lds_bytes, s, pview, Q, c->lds_cfg, cur, (uint32_t) n_wg, c->d_next_pixel.p));
                    else MIW_TIMED(0, hipLaunchKernelGGL(k_trace_stream<0>, sgrid, block, c->lds_bytes, s, pview, Q, c->lds_cfg, cur, (uint32_t) n_wg, c->d_next_pixel.p));
                    MIW_TIMED(1, hipLaunchKernelGGL(k_sort_hits, grid, block, 0, s, pview, Q, cur, c->d_next_pixel.p));
                    K.n_trace_closest++; K.n_trace_any++;
                } else {
                    if (!first) {
                        MIW_TIMED(1, hipLaunchKernelGGL(k_trace<true>, grid, block, c->lds_bytes, s, c->view, Q, c->lds_cfg, cur));
                        K.n_trace_any++;
                    }
                    MIW_TIMED(0, hipLaunchKernelGGL(k_trace<false>, grid, block, c->lds_bytes, s, c->view, Q, c->lds_cfg, cur));
                    K.n_trace_closest++;
                }
                const uint32_t count_active = it == check_every - 1 ? 1u : 0u;
                if (film_mode == 1)
                    MIW_TIMED(2, hipLaunchKernelGGL(k_shade<true>, grid, block, 0, s, P, pview, Q, (double *) nullptr, c->d_cnt.p, count_active, cur, nxt));
                else
                    MIW_TIMED(2, hipLaunchKernelGGL(k_shade<false>, grid, block, 0, s, P, pview, Q, c->d_accum.p, c->d_cnt.p, count_active, cur, nxt));
                K.n_shade++; K.iterations++;
                first = false;
            }
            HIP_TRY(c, hipGetLastError());
            Counters sum;
            mi_status rs = read_counters(sum);
            if (rs != MI_OK) return rs;
            T.drain();
            // active_lanes accumulates over the counting launches: the last one's share is the delta
            unsigned long long active_now = sum.active_lanes - active_prev;
            active_prev = sum.active_lanes;
            if (active_now == 0) break;
            if (c->cancel.load()) { result = MI_ERR_CANCELLED; break; }
            if (out_of_time()) { result = MI_ERR_CANCELLED; break; }
        }
    }
#endif

    // film assembly -> caller's buffer (device pointer, or staged through d_out for a host pointer)
    { const mi_status fs = assemble_film(c, cfg, ropt, P, s, T, FP, n_tiles, bs, bs2_log2, blocks_x, blocks_y, film_n, film, result, c->replay_enqueued ? &replay : nullptr); c->replay_enqueued = false; if (fs != MI_OK) return fs; }
#undef MIW_TIMED
    K.ms_render = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (result == MI_ERR_CANCELLED) c->error = "render cancelled";
    return result;
}

// ---- mi_eval -----------------------------------------------------------------------------
mi_status mi_eval(mi_ctx *c, int32_t op, const mi_render_cfg *cfg, const float *in, int32_t is, float *out, int32_t os, uint64_t n) {
    if (!c || !in || !out || is <= 0 || os <= 0) return MI_ERR_INVALID;
    if (n == 0) return MI_OK;
    RenderParams P; memset(&P, 0, sizeof P);
    if (op == MI_EVAL_CAMERA_RAY) {
        if (!cfg) return fail(c, MI_ERR_INVALID, "mi_eval: camera ray needs a render cfg");
        mi_status st = fill_params(c, cfg, P);
        if (st != MI_OK) return st;
    }
    if ((op == MI_EVAL_BSDF || op == MI_EVAL_EMITTER_SAMPLE || op == MI_EVAL_ENVMAP || op == MI_EVAL_TEXTURE) && !c->have_bvh)
        return fail(c, MI_ERR_STATE, "mi_eval: scene required");
    HIP_TRY(c, hipSetDevice(c->device));
    TmpBuf<float> din, dout;
    HIP_TRY(c, din.resize(n * is)); HIP_TRY(c, dout.resize(n * os));
    HIP_TRY(c, hipMemcpyAsync(din.p, in, n * is * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(dout.p, 0, n * os * sizeof(float), c->stream));
    hipLaunchKernelGGL(k_eval, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, c->stream, op, P, c->view, din.p, is, dout.p, os, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout.p, n * os * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MI_OK;
}

// ---- mi_selftest -------------------------------------------------------------------------
__global__ void k_selftest_rcp(unsigned long long *mismatches) {
    unsigned long long bad = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = u2f((uint32_t) i);
        volatile float one = 1.f;                               // keeps the division a division
        const float ref = one / x, got = miw::rcp(x), got_l = miw::rcp_loop(x);
        if (f2u(ref) != f2u(got) && !(ref != ref && got != got)) ++bad;
        if (f2u(ref) != f2u(got_l) && !(ref != ref && got_l != got_l)) ++bad;
    }
    bad = wave_sum(bad);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(mismatches, bad);
}

mi_status mi_selftest(mi_ctx *c, int32_t which, uint64_t *mismatches) {
    if (!c || !mismatches) return MI_ERR_INVALID;
    if (which != MI_SELFTEST_RCP) return fail(c, MI_ERR_INVALID, "mi_selftest: unknown test %d", which);
    HIP_TRY(c, hipSetDevice(c->device));
    TmpBuf<unsigned long long> d;
    HIP_TRY(c, d.resize(1));
    HIP_TRY(c, hipMemsetAsync(d.p, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_selftest_rcp, dim3(4096), dim3(256), 0, c->stream, d.p);
    HIP_TRY(c, hipGetLastError());
    unsigned long long h = 0;
    HIP_TRY(c, hipMemcpyAsync(&h, d.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *mismatches = h;
    return MI_OK;
}

} // extern "C"
