// Calibration of rocprofv3's FETCH_SIZE for the phase machine's access pattern: a kernel whose
// ALGORITHMIC bytes are known exactly, reading the way a node step reads — every lane fetches whole 80-byte records (five
// global_load_dwordx4) at pseudo-random, 16-byte-aligned places of a table — over tables below the aggregate L2 (32 MiB), below the
// Infinity Cache (256 MiB) and far above it; plus the guide's control case, a wide coalesced stream (16 B per lane), whose
// FETCH_SIZE is known to report half its bytes on gfx950 (MI355X_MICROARCH.md, section HBM).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
//   tools/ubench/fetch_calib <mode: scatter80 | scatter48 | stream> <table MiB> [steps]
// prints one line: mode, table MiB, requested bytes, distinct 64-byte sectors / 128-byte lines those requests touch (counted on
// the host from the same index sequence: what a cache that never hits would have to move), kernel ms.
// tools/fetch_calib.sh runs it under rocprofv3 --pmc FETCH_SIZE / TCC_HIT / TCC_MISS / TCC_EA0_RDREQ and prints the table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__host__ __device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// record r of a table of n_rec records of REC sixteen-byte words: (thread, step) -> record, the same on host and device
__host__ __device__ inline uint64_t pick(uint32_t thread, uint32_t step, uint64_t n_rec) {
    const uint64_t h = ((uint64_t) mix(thread * 0x9e3779b9u + step) << 32) | mix(step * 0x85ebca6bu ^ thread);
    return h % n_rec;
}

template <int REC>   // REC = sixteen-byte words per record: 5 = an 80-byte BVH8 node, 3 = a 48-byte triangle
__global__ __launch_bounds__(256, 4) void k_scatter(const u4 *tab, uint64_t n_rec, uint32_t steps, uint32_t *out) {
    const uint32_t thread = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        const u4 *p = tab + (size_t) REC * pick(thread, s, n_rec);
#pragma unroll
        for (int k = 0; k < REC; ++k) { const u4 q = p[k]; acc += q.x ^ q.y ^ q.z ^ q.w; }
    }
    out[thread] = acc;
}
__global__ __launch_bounds__(256, 4) void k_stream(const u4 *tab, uint64_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t) gridDim.x * 256u) { const u4 q = tab[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "scatter80";
    const uint64_t mib = argc > 2 ? (uint64_t) atoll(argv[2]) : 128u;
    const uint32_t steps = argc > 3 ? (uint32_t) atoi(argv[3]) : 256u;
    const uint64_t bytes = mib << 20;
    u4 *tab; uint32_t *out;
    const uint32_t blocks = 256 * 4 * 4 / 4;                      // 256 CUs x 4 SIMDs x 4 waves / 4 waves per block: one resident wave set
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, blocks * 256 * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(tab, 0x5a, bytes);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t threads = blocks * 256u;
    double requested = 0.0; uint64_t sectors = 0, lines = 0;
    const int rec = !strcmp(mode, "scatter48") ? 3 : 5;
    const uint64_t n_rec = bytes / (16u * rec);
    float ms = 0.f;
    if (!strcmp(mode, "stream")) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, tab, bytes / 16, out);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        requested = (double) bytes; sectors = bytes / 64; lines = bytes / 128;
    } else {
        hipEventRecord(a);
        if (rec == 5) hipLaunchKernelGGL(k_scatter<5>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, steps, out);
        else hipLaunchKernelGGL(k_scatter<3>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, steps, out);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        requested = (double) threads * steps * 16.0 * rec;
        // the 64-byte sectors / 128-byte lines every request touches, summed over requests (no reuse assumed: a cache that never hits)
        for (uint32_t t = 0; t < threads; t += 97u)               // (a sample of the threads, scaled up: the sequence is i.i.d.)
            for (uint32_t s = 0; s < steps; ++s) {
                const uint64_t off = pick(t, s, n_rec) * 16u * rec, end = off + 16u * rec - 1u;
                sectors += end / 64 - off / 64 + 1; lines += end / 128 - off / 128 + 1;
            }
        const double scale = (double) threads / (double) ((threads + 96u) / 97u);
        sectors = (uint64_t) ((double) sectors * scale); lines = (uint64_t) ((double) lines * scale);
    }
    printf("%s table_MiB %llu requested_bytes %.0f sector64_bytes %.0f line128_bytes %.0f kernel_ms %.3f GBps_requested %.1f\n", mode, (unsigned long long) mib,
           requested, (double) sectors * 64.0, (double) lines * 128.0, ms, requested / ms / 1e6);
    return 0;
}
