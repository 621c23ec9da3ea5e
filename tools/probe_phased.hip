// ISA probe: only k_path_phased<MATS_PLAIN, false, MIW_PHASE_SPEC> (the C3 / C4 kernel), for quick -save-temps inspection:
//   cd /tmp/x && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -c -save-temps <repo>/tools/probe_phased.hip
#include <hip/hip_runtime.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../include/miwave.h"
#include "../mitsuba2_amd/csrc/miw/base.h"
#include "../mitsuba2_amd/csrc/miw/rng.h"
#include "../mitsuba2_amd/csrc/miw/warp.h"
#include "../mitsuba2_amd/csrc/miw/special.h"
#include "../mitsuba2_amd/csrc/miw/shape.h"
#include "../mitsuba2_amd/csrc/miw/bsdf.h"
#include "../mitsuba2_amd/csrc/miw/scene.h"
#include "../mitsuba2_amd/csrc/miw/film.h"
#include "../mitsuba2_amd/csrc/miw/bvh.h"
#include "../mitsuba2_amd/csrc/miw/bvh4.h"
#include "../mitsuba2_amd/csrc/miw/bvh8.h"
#include "../mitsuba2_amd/csrc/miw/path.h"
#include "../mitsuba2_amd/csrc/miw/direct.h"
using namespace miw;
#define MIW_BLOCK 256
#define MIW_CNT_SHARDS 1024
#include "../mitsuba2_amd/csrc/device/trace.h"
#include "../mitsuba2_amd/csrc/device/wavefront_kernels.h"
#include "../mitsuba2_amd/csrc/device/resident_kernel.h"
#include "../mitsuba2_amd/csrc/device/phased_kernel.h"
#if defined(MIW_PROBE_POOLED)   // only the pooled phase machine of BASELINE configs 3 / 4 (round 6: device/pooled_kernel.h)
#ifndef MIW_POOL_NW
#define MIW_POOL_NW 8
#endif
#include "../mitsuba2_amd/csrc/device/pooled_kernel.h"
#ifndef MIW_POOL_PP
#define MIW_POOL_PP 2
#endif
template __global__ void k_path_pooled<MATS_TRIO, false, MIW_POOL_NW, MIW_POOL_PP>(RenderParams, SceneView, LaneQueues, Counters *, TraceLds, uint32_t, uint32_t *);
#elif defined(MIW_PROBE_PLAIN_ALL)   // the path integrator's phase machine for the general scene classes (for comparison with the direct one)
template __global__ void k_path_phased<MATS_PLAIN, true, MIW_PHASE_SPEC != 0, 4, 2, false>(RenderParams, SceneView, LaneQueues, Counters *, TraceLds, uint32_t, uint32_t *);
template __global__ void k_path_phased<MATS_ALL, true, MIW_PHASE_SPEC != 0, 4, 2, false>(RenderParams, SceneView, LaneQueues, Counters *, TraceLds, uint32_t, uint32_t *);
#elif defined(MIW_PROBE_C34)   // only the kernels of BASELINE configs 3 / 4: MATS_TRIO over the 8-wide tree (round 5) and its 4-wide twin, four wavefronts per SIMD (tests/test_kernel_budget.py)
template __global__ void k_path_phased<MATS_TRIO, false, MIW_PHASE_SPEC != 0, 4, 2>(RenderParams, SceneView, LaneQueues, Counters *, TraceLds, uint32_t, uint32_t *);
template __global__ void k_path_phased<MATS_TRIO, false, MIW_PHASE_SPEC != 0, 4, 1>(RenderParams, SceneView, LaneQueues, Counters *, TraceLds, uint32_t, uint32_t *);
#else
template __global__ void k_path_phased<MATS_PLAIN, false, MIW_PHASE_SPEC != 0>(RenderParams, SceneView, LaneQueues, Counters *, TraceLds, uint32_t, uint32_t *);
#include "../mitsuba2_amd/csrc/device/stream_trace.h"
template __global__ void k_path_phased<MATS_TRIO, false, MIW_PHASE_SPEC != 0, 3, 0>(RenderParams, SceneView, LaneQueues, Counters *, TraceLds, uint32_t, uint32_t *);
#endif
