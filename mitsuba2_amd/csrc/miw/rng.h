// PCG32 sampler stream, TEA hash and Morton decode — integer work, bit-exact by
// construction.
//
// Reference call sites: enoki::PCG32 (not vendored; include/mitsuba/core/random.h:36,52-54),
// src/librender/sampler.cpp:83-96 (seeding), src/samplers/independent.cpp:73-82
// (next_1d / next_2d), include/mitsuba/core/random.h:75-116 (TEA),
// src/librender/integrator.cpp:200 (enoki::morton_decode).
// The PCG32 algorithm is O'Neill's pcg32 (XSH-RR 64/32) exactly as enoki/random.h
// restates it; pinned in tests by the public pcg32-demo vector seed(42,54).
#pragma once
#include "base.h"

namespace miw {

#define MIW_PCG32_DEFAULT_STATE  0x853c49e6748fea9bULL
#define MIW_PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define MIW_PCG32_MULT           0x5851f42d4c957f2dULL

struct PCG32 { uint64_t state, inc; };

MIW_HD uint32_t pcg32_next_u32(PCG32 &r) {
    uint64_t old = r.state;
    r.state = old * MIW_PCG32_MULT + r.inc;
    uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t) (old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
}

MIW_HD void pcg32_seed(PCG32 &r, uint64_t initstate, uint64_t initseq) {
    r.state = 0u;
    r.inc = (initseq << 1u) | 1u;
    pcg32_next_u32(r);
    r.state += initstate;
    pcg32_next_u32(r);
}

// next_float32: uniform on [0,1) with 23 random mantissa bits
MIW_HD float pcg32_next_f32(PCG32 &r) {
    return u2f((pcg32_next_u32(r) >> 9) | 0x3f800000u) - 1.f;
}

// The scalar_* sampler uses inc = (PCG32_DEFAULT_STREAM << 1) | 1 for every
// pixel (sampler.cpp:94), so a lane only has to carry `state`.
#define MIW_PCG32_SCALAR_INC ((MIW_PCG32_DEFAULT_STREAM << 1u) | 1u)

// include/mitsuba/core/random.h:75-86
MIW_HD uint32_t sample_tea_32(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v1;
}
// random.h:105-116
MIW_HD uint64_t sample_tea_64(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return (uint64_t) v0 + ((uint64_t) v1 << 32);
}
// random.h:136-140 / :163-167
MIW_HD float sample_tea_float32(uint32_t v0, uint32_t v1, int rounds) {
    return u2f((sample_tea_32(v0, v1, rounds) >> 9) | 0x3f800000u) - 1.f;
}
MIW_HD double sample_tea_float64(uint32_t v0, uint32_t v1, int rounds) {
    union { uint64_t u; double d; } c;
    c.u = (sample_tea_64(v0, v1, rounds) >> 12) | 0x3ff0000000000000ull;
    return c.d - 1.0;
}

// 2-D Morton decode: x = even bits, y = odd bits
MIW_HD uint32_t morton_compact1(uint32_t x) {
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    x = (x ^ (x >> 8)) & 0x0000ffffu;
    return x;
}
MIW_HD void morton_decode2(uint32_t i, uint32_t &x, uint32_t &y) {
    x = morton_compact1(i);
    y = morton_compact1(i >> 1);
}

} // namespace miw
