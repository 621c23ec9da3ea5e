// ISA probe: only the film replay, k_film_columns<4, 2> (see tools/probe_phased.hip):
//   cd /tmp/x && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-flush-denormals-to-zero -c -save-temps <repo>/tools/probe_film.hip
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
#include "../mitsuba2_amd/csrc/miw/path.h"
#include "../mitsuba2_amd/csrc/miw/direct.h"
using namespace miw;
#define MIW_BLOCK 256
#define MIW_CNT_SHARDS 1024
#include "../mitsuba2_amd/csrc/device/trace.h"
#include "../mitsuba2_amd/csrc/device/wavefront_kernels.h"
#include "../mitsuba2_amd/csrc/device/resident_kernel.h"
#include "../mitsuba2_amd/csrc/miw/film_gather.h"
#include "../mitsuba2_amd/csrc/film_classes.h"
#include "../mitsuba2_amd/csrc/device/film_kernels.h"
template __global__ void k_film_columns<4, 2>(FilmRec, BlockReplayArgs, PatchArgs, float *);
template __global__ void k_film_quads<2, 4, 4>(FilmRec, BlockReplayArgs, PatchArgs, uint32_t, float *);
template __global__ void k_film_quads<4, 2, 4>(FilmRec, BlockReplayArgs, PatchArgs, uint32_t, float *);
template __global__ void k_film_lanes<4, 0>(FilmRec, BlockReplayArgs, PatchArgs, uint32_t, float *, uint32_t, uint32_t);
template __global__ void k_film_lanes<4, 1>(FilmRec, BlockReplayArgs, PatchArgs, uint32_t, float *, uint32_t, uint32_t);
template __global__ void k_film_lanes<2, 1, 4>(FilmRec, BlockReplayArgs, PatchArgs, uint32_t, float *, uint32_t, uint32_t);   // the 128-register form that runs beside the path kernel (round 6)
