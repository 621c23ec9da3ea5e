// Exhaustive check of candidate fast reciprocals against the IEEE-754 correctly rounded 1.f / x
// (the compiler's v_div_scale / v_div_fmas / v_div_fixup expansion, denormals on) over all 2^32 inputs.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-gpu-flush-denormals-to-zero tools/rcp_exhaustive.hip -o /tmp/rcp_ex
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float cand(int which, float x) {
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.f);
    r = __builtin_fmaf(e, r, r);
    if (which >= 2) { e = __builtin_fmaf(-x, r, 1.f); r = __builtin_fmaf(e, r, r); }
    if (which == 1 || which == 3) r = __builtin_amdgcn_div_fixupf(r, x, 1.f);
    return r;
}

__device__ __forceinline__ float cand_sqrt(int which, float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    float r = __builtin_fmaf(-s, s, x);
    float h = 0.5f * __builtin_amdgcn_rcpf(s);
    s = __builtin_fmaf(r, h, s);
    if (which == 1) { r = __builtin_fmaf(-s, s, x); s = __builtin_fmaf(r, h, s); }
    return s;
}
__global__ void k_check_sqrt(unsigned long long *hist) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {   // non-negative inputs
        const uint32_t bits = (uint32_t) i;
        float x; memcpy(&x, &bits, 4);
        const float ref = __builtin_sqrtf(x);
        uint32_t rb; memcpy(&rb, &ref, 4);
        for (int w = 0; w < 2; ++w) {
            const float c = cand_sqrt(w, x);
            uint32_t cb; memcpy(&cb, &c, 4);
            if (cb != rb && !((ref != ref) && (c != c))) atomicAdd(&hist[w * 256 + ((bits >> 23) & 255u)], 1ull);
        }
    }
}

// per exponent-field histogram of mismatches: hist[which][exp]
__global__ void k_check(unsigned long long *hist, uint32_t *first_bad) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t) i;
        float x; memcpy(&x, &bits, 4);
        volatile float one = 1.f;
        const float ref = one / x;
        uint32_t rb; memcpy(&rb, &ref, 4);
        for (int w = 0; w < 4; ++w) {
            const float c = cand(w, x);
            uint32_t cb; memcpy(&cb, &c, 4);
            const bool both_nan = (ref != ref) && (c != c);
            if (cb != rb && !both_nan) {
                atomicAdd(&hist[w * 256 + ((bits >> 23) & 255u)], 1ull);
                atomicMin(&first_bad[w], bits & 0x7fffffffu);
            }
        }
    }
}

int main() {
    unsigned long long *d_hist; uint32_t *d_first;
    hipMalloc(&d_hist, 4 * 256 * 8); hipMemset(d_hist, 0, 4 * 256 * 8);
    hipMalloc(&d_first, 16); hipMemset(d_first, 0xff, 16);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d_hist, d_first);
    hipDeviceSynchronize();
    static unsigned long long h[4 * 256]; uint32_t f[4];
    hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(f, d_first, 16, hipMemcpyDeviceToHost);
    const char *names[4] = { "rcp+1NR", "rcp+1NR+fixup", "rcp+2NR", "rcp+2NR+fixup" };
    for (int w = 0; w < 4; ++w) {
        unsigned long long tot = 0; int lo = -1, hi = -1;
        for (int e = 0; e < 256; ++e) { tot += h[w * 256 + e]; }
        printf("%-14s mismatches %llu (first |x| bits 0x%08x); by exponent field:", names[w], tot, f[w]);
        for (int e = 0; e < 256; ++e) if (h[w * 256 + e]) printf(" %d:%llu", e, h[w * 256 + e]);
        printf("\n");
    }
    hipMemset(d_hist, 0, 4 * 256 * 8);
    hipLaunchKernelGGL(k_check_sqrt, dim3(4096), dim3(256), 0, 0, d_hist);
    hipDeviceSynchronize();
    hipMemcpy(h, d_hist, sizeof h, hipMemcpyDeviceToHost);
    const char *sn[2] = { "sqrt+1step", "sqrt+2step" };
    for (int w = 0; w < 2; ++w) {
        unsigned long long tot = 0;
        for (int e = 0; e < 256; ++e) tot += h[w * 256 + e];
        printf("%-14s mismatches %llu; by exponent field:", sn[w], tot);
        for (int e = 0; e < 256; ++e) if (h[w * 256 + e]) printf(" %d:%llu", e, h[w * 256 + e]);
        printf("\n");
    }
    return 0;
}
