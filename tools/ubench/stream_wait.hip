// Feasibility probe (round 6): can a second stream start a kernel while a long kernel of the first stream is still running, released by a
// value that long kernel writes (hipStreamWaitValue32 on signal memory)? Prints when kernel B ran relative to kernel A's start / end.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/stream_wait.hip -o tools/ubench/stream_wait && tools/ubench/stream_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_long(uint32_t *flag, unsigned long long *t, int spin_ms, int grid_waves) {
    const unsigned long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = t0;
    // 100 MHz wall clock: spin_ms ms
    const unsigned long long half = (unsigned long long) spin_ms * 100000ull / 2;
    bool told = false;
    while (wall_clock64() - t0 < 2 * half) {
        if (!told && wall_clock64() - t0 > half) {
            told = true;
            if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        __builtin_amdgcn_s_sleep(10);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) t[1] = wall_clock64();
}
__global__ void k_short(unsigned long long *t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = wall_clock64(); }
static int run(uint32_t *flag, const char *what) {
    unsigned long long *t = nullptr;
    CK(hipMalloc((void **) &t, 64));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int fill = 0; fill < 3; ++fill) {
        // fill = 0: kernel A leaves most of the machine free; 1: kernel A asks for every wave slot (8 waves per SIMD); 2: kernel A on the NULL stream
        const int blocks = fill == 1 ? 256 * 8 : 64;
        hipStream_t sa = fill == 2 ? (hipStream_t) nullptr : a;
        *(volatile uint32_t *) flag = 0; CK(hipMemset(t, 0, 64)); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_long, dim3(blocks), dim3(256), 0, sa, flag, t, 40, blocks);
        hipError_t e = hipStreamWaitValue32(b, flag, (uint32_t) blocks, hipStreamWaitValueGte, 0xffffffffu);
        if (e != hipSuccess) { printf("%s: hipStreamWaitValue32: %s\n", what, hipGetErrorString(e)); return 1; }
        hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, b, t);
        CK(hipDeviceSynchronize());
        unsigned long long h[3]; CK(hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost));
        printf("%s, %4d blocks%s: kernel A ran %.2f ms; kernel B ran %.2f ms after A's start (%s A's end)\n", what, blocks, fill == 2 ? " on the null stream" : "", (h[1] - h[0]) / 1e5,
               (double) ((long long) h[2] - (long long) h[0]) / 1e5, h[2] < h[1] ? "BEFORE" : "after");
    }
    return 0;
}
int main() {
    uint32_t *flag = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **) &flag, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(8 bytes, hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) { CK(hipMemset(flag, 0, 8)); /* device memory: the host cannot write it */ }
    uint32_t *hflag = nullptr;
    CK(hipHostMalloc((void **) &hflag, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    return run(hflag, "host-coherent flag");
}
