// Micro-benchmark behind DESIGN.md section 4.2's question "what does a wavefront pay for 64 scattered 64-byte node records?":
// every lane chases its own pseudo-random chain through a table of 64-byte records (the phase machine's node step without its
// arithmetic), with the record fetched
//   A  as the kernels do it: four global_load_dwordx4 per lane (4 x 64 scattered 16-byte requests per wave and step),
//   B  cooperatively: four lanes fetch the four quarters of ONE record (so a wave instruction covers 16 whole records, 64 contiguous
//      bytes each) and the quarters travel to their owner through ds_bpermute,
//   C  one dwordx4 per lane only (a 16-byte record: the floor for one request per lane).
// Usage: hipcc --offload-arch=gfx950 -O3 tools/ubench/node_fetch.hip -o /tmp/node_fetch && /tmp/node_fetch [records] [steps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int Mode>
__global__ __launch_bounds__(256, 4) void k_chase(const u4 *tab, uint32_t n_rec, uint32_t steps, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) % n_rec, acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        u4 q0, q1, q2, q3;
        if (Mode == 0) {
            const u4 *p = tab + 4 * (size_t) idx;
            q0 = p[0]; q1 = p[1]; q2 = p[2]; q3 = p[3];
        } else if (Mode == 1) {
            // pass k: lane l fetches quarter (l & 3) of the record wanted by lane 16 k + (l >> 2)
            u4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t want = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((16u * k + (lane >> 2)) << 2), (int) idx);
                r[k] = tab[4 * (size_t) want + (lane & 3u)];
            }
            // the owner j = 16 k + m finds quarter c of its record in lane 4 m + c of pass k = j >> 4
            const uint32_t m = lane & 15u, kk = lane >> 4;
            u4 got[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int src = (int) ((4u * m + c) << 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // every lane reads the same pass's register? no: its pass is kk — select the register first, then permute
                    const uint32_t v0 = (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) r[0][e]), v1 = (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) r[1][e]),
                                   v2 = (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) r[2][e]), v3 = (uint32_t) __builtin_amdgcn_ds_bpermute(src, (int) r[3][e]);
                    got[c][e] = kk == 0 ? v0 : kk == 1 ? v1 : kk == 2 ? v2 : v3;
                }
            }
            q0 = got[0]; q1 = got[1]; q2 = got[2]; q3 = got[3];
        } else if (Mode == 3) {
            // cooperative, transposed inside each quad of lanes with DPP (no LDS): pass k: lane q of a quad fetches quarter q of the
            // record wanted by the quad's lane k; then a two-round butterfly (lane ^ 1, lane ^ 2) turns "quarter q of records 0..3"
            // into "quarters 0..3 of my record"
            const uint32_t q = lane & 3u;
            u4 m[4];
#define PASS(k) { const uint32_t want = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) idx, (k) * 0x55, 0xf, 0xf, false); m[k] = tab[4 * (size_t) want + q]; }
            PASS(0) PASS(1) PASS(2) PASS(3)
#undef PASS
            const bool b0 = q & 1u, b1 = q & 2u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {                  // round 1: pairs (0,1), (2,3) with lane ^ 1
                    const int lo = 2 * pr, hi = 2 * pr + 1;
                    const uint32_t send = b0 ? m[lo][e] : m[hi][e];
                    const uint32_t recv = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) send, 0xB1, 0xf, 0xf, false);
                    m[hi][e] = b0 ? m[hi][e] : recv; m[lo][e] = b0 ? recv : m[lo][e];
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {                  // round 2: pairs (0,2), (1,3) with lane ^ 2
                    const int lo = pr, hi = pr + 2;
                    const uint32_t send = b1 ? m[lo][e] : m[hi][e];
                    const uint32_t recv = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) send, 0x4E, 0xf, 0xf, false);
                    m[hi][e] = b1 ? m[hi][e] : recv; m[lo][e] = b1 ? recv : m[lo][e];
                }
            }
            q0 = m[0]; q1 = m[1]; q2 = m[2]; q3 = m[3];
        } else if (Mode == 4) {
            // cooperative by PAIRS: pass (j, h): lane p of a pair fetches quarter 2 h + p of the record wanted by the pair's lane j
            // (32 contiguous bytes per pair and instruction); one butterfly round (lane ^ 1)
            const uint32_t p1 = lane & 1u;
            const uint32_t w0 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) idx, 0xA0 /* quad_perm [0,0,2,2] */, 0xf, 0xf, false),
                           w1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) idx, 0xF5 /* quad_perm [1,1,3,3] */, 0xf, 0xf, false);
            u4 a0 = tab[4 * (size_t) w0 + p1], a1 = tab[4 * (size_t) w0 + 2u + p1], c0 = tab[4 * (size_t) w1 + p1], c1 = tab[4 * (size_t) w1 + 2u + p1];
            // lane 0 holds quarters {0, 2} of both records, lane 1 quarters {1, 3}; lane j keeps its own record's and swaps the other's
            const bool b0 = p1;
            u4 mine0, mine1, other0, other1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t s0 = b0 ? a0[e] : c0[e], s1 = b0 ? a1[e] : c1[e];      // what the partner's record needs from me
                other0[e] = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) s0, 0xB1, 0xf, 0xf, false);
                other1[e] = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) s1, 0xB1, 0xf, 0xf, false);
                mine0[e] = b0 ? c0[e] : a0[e]; mine1[e] = b0 ? c1[e] : a1[e];
            }
            // lane 0: mine = quarters 0, 2 ; other = quarters 1, 3.  lane 1: mine = quarters 1, 3 ; other = quarters 0, 2
            q0 = b0 ? other0 : mine0; q1 = b0 ? mine0 : other0; q2 = b0 ? other1 : mine1; q3 = b0 ? mine1 : other1;
        } else {
            q0 = tab[4 * (size_t) idx]; q1 = q2 = q3 = q0;
        }
        const uint32_t h = q0.x ^ q1.y ^ q2.z ^ q3.w;
        acc += h;
        idx = mix(h + s) % n_rec;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_rec = argc > 1 ? (uint32_t) atoi(argv[1]) : 251152u, steps = argc > 2 ? (uint32_t) atoi(argv[2]) : 2000u;
    std::vector<uint32_t> h((size_t) n_rec * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t) (i * 2654435761u) ^ (uint32_t) (i >> 7);
    u4 *tab; uint32_t *out;
    const uint32_t blocks = 256 * 4 * 4 / 4;      // 256 CUs x 4 SIMDs x 4 waves / 4 waves per block
    hipMalloc(&tab, h.size() * 4); hipMalloc(&out, blocks * 256 * 4);
    hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    uint32_t ref = 0;
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, steps, out);
            if (mode == 1) hipLaunchKernelGGL(k_chase<1>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, steps, out);
            if (mode == 2) hipLaunchKernelGGL(k_chase<2>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, steps, out);
            if (mode == 3) hipLaunchKernelGGL(k_chase<3>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, steps, out);
            if (mode == 4) hipLaunchKernelGGL(k_chase<4>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, steps, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            uint32_t v; hipMemcpy(&v, out + 12345, 4, hipMemcpyDeviceToHost);
            if (mode == 0) ref = v;
            if (rep) printf("mode %d (%s): %8.3f ms, %7.1f ns per step, %6.2f G record fetches/s, %7.1f GB/s of records%s\n", mode,
                            mode == 0 ? "4 x dwordx4 per lane" : mode == 1 ? "cooperative 4 lanes per record + bpermute" : mode == 2 ? "1 x dwordx4 per lane (16 B records)" : mode == 3 ? "cooperative quads + DPP butterfly" : "cooperative pairs + DPP swap",
                            ms, ms * 1e6 / steps, (double) blocks * 256 * steps / ms / 1e6, (double) blocks * 256 * steps * (mode == 2 ? 16 : 64) / ms / 1e6,
                            mode != 0 && mode != 2 ? (v == ref ? "  [same result as mode 0]" : "  [RESULT DIFFERS]") : "");
        }
    }
    return 0;
}
