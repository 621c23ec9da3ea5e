// miwave leaf arithmetic: scalar/vector primitives shared by the gfx950 kernels
// and by every host-side consumer (host classes, CPU checker).
//
// Bit-parity contract: every function below is built
// from IEEE-754 correctly rounded float32 {+,-,*,/,sqrt,fma} only, written in
// one fixed operation order, and compiled with -ffp-contract=off on every
// target, so that hipcc (gfx950) and g++ (x86-64) produce identical bits.
// `fmaf` appears exactly where the reference spells fmadd/fmsub/fnmadd (or
// where Enoki's own dot/cross/matrix kernels are fma chains).
//
// Frozen Enoki scalar semantics (the library is not vendored in the reference
// checkout, so these choices are *definitions* for this code base):
//   rcp(x) = 1/x           rsqrt(x) = 1/sqrt(x)      sqr(x) = x*x
//   max(a,b) = a<b ? b : a  min(a,b) = b<a ? b : a    (std::max/min semantics)
//   safe_sqrt(x) = sqrt(max(x,0))
//   dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   cross(a,b) = fmsub(a.y,b.z, a.z*b.y), ...          (enoki cross kernel)
//   v / s      = v * rcp(s)                            (array / scalar)
//   M * v      = fma(col2,v.z, fma(col1,v.y, col0*v.x))
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#  include <hip/hip_runtime.h>
#  define MIW_HD __host__ __device__ __forceinline__
#else
#  define MIW_HD inline
#endif

namespace miw {

// include/mitsuba/core/math.h:25-38
#define MIW_PI          3.14159265358979323846f
#define MIW_INV_PI      0.31830988618379067154f
#define MIW_INFINITY    __builtin_inff()
#define MIW_EPSILON     5.9604644775390625e-08f          /* 2^-24 = float eps / 2 */
#define MIW_RAY_EPSILON (5.9604644775390625e-08f * 1500.f)
#define MIW_SHADOW_EPSILON (5.9604644775390625e-08f * 1500.f * 10.f)

MIW_HD float fmadd(float a, float b, float c)  { return __builtin_fmaf(a, b, c); }
MIW_HD float fmsub(float a, float b, float c)  { return __builtin_fmaf(a, b, -c); }
MIW_HD float fnmadd(float a, float b, float c) { return __builtin_fmaf(-a, b, c); }
// rcp(x) = the correctly rounded 1/x on every target. On gfx950 v_rcp_f32 (1 ulp) followed by one Newton step
// IS the correctly rounded reciprocal for every |x| in [2^-126, 2^126) — checked against the IEEE division over
// all 2^32 inputs (mi_selftest(MI_SELFTEST_RCP), tools/rcp_exhaustive.hip); zeros, denormals, |x| >= 2^126, inf
// and NaN take the compiler's v_div_scale / v_div_fmas / v_div_fixup expansion. 5 issue slots instead of 10.
MIW_HD float rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float ax = __builtin_fabsf(x);
    if (__builtin_expect(ax >= 0x1p-126f && ax < 0x1p126f, 1)) {
        const float r = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(__builtin_fmaf(-x, r, 1.f), r, r);
    }
#endif
    return 1.f / x;
}
// The same reciprocal for inner loops (round 4): the fast form unconditionally, the division only as a patch over it for the
// inputs outside its range — ONE predicated region instead of an if / else pair, i.e. fewer exec-mask instructions around the
// reciprocal (the packet kernel's candidate loops: path kernel of C2 760.6 -> 753.4 ms, gpurun r4c). Not the general form:
// used everywhere it costs the 128-register phase machine ten more spilled registers (C4 -1.4 %).
MIW_HD float rcp_loop(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float ax = __builtin_fabsf(x);
    const float r0 = __builtin_amdgcn_rcpf(x);
    float r = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.f), r0, r0);
    if (__builtin_expect(!(ax >= 0x1p-126f && ax < 0x1p126f), 0)) r = 1.f / x;
    return r;
#else
    return 1.f / x;
#endif
}
MIW_HD float sqr(float x)   { return x * x; }
MIW_HD float rsqrt(float x) { return 1.f / __builtin_sqrtf(x); }
MIW_HD float max_(float a, float b) { return a < b ? b : a; }
MIW_HD float min_(float a, float b) { return b < a ? b : a; }
MIW_HD float abs_(float a) { return __builtin_fabsf(a); }
MIW_HD float safe_sqrt(float x) { return __builtin_sqrtf(max_(x, 0.f)); }
MIW_HD float clamp_(float v, float lo, float hi) { return min_(max_(v, lo), hi); }

MIW_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
MIW_HD float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

// enoki sign/mulsign: sign-bit transfers, never a multiply
MIW_HD float sign_(float x)              { return u2f(0x3f800000u | (f2u(x) & 0x80000000u)); }
MIW_HD float mulsign(float a, float b)     { return u2f(f2u(a) ^ (f2u(b) & 0x80000000u)); }
MIW_HD float mulsign_neg(float a, float b) { return u2f(f2u(a) ^ (~f2u(b) & 0x80000000u)); }
MIW_HD bool  isfinite_(float x) { return (f2u(x) & 0x7f800000u) != 0x7f800000u; }

// enoki lerp(a,b,t) = fmadd(b, t, fnmadd(a, t, a))
MIW_HD float lerp_(float a, float b, float t) { return fmadd(b, t, fnmadd(a, t, a)); }

struct V2 { float x, y; };
struct V3 { float x, y, z; };
// 16- / 8-byte queue words (one dwordx4 / dwordx2 per lane)
struct alignas(16) F4 { float x, y, z, w; };
struct alignas(16) U4 { uint32_t x, y, z, w; };
struct alignas(8)  F2 { float x, y; };
struct alignas(8)  U2 { uint32_t x, y; };

MIW_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MIW_HD V3 v3(float s) { return v3(s, s, s); }
MIW_HD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }

MIW_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
MIW_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
MIW_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
MIW_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
MIW_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
MIW_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
// array / scalar  ==  array * rcp(scalar)   (frozen, see header)
MIW_HD V3 operator/(V3 a, float s) { float r = rcp(s); return v3(a.x * r, a.y * r, a.z * r); }
// array / array: component-wise IEEE division
MIW_HD V3 div3(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }

MIW_HD float dot(V3 a, V3 b) { return fmadd(a.z, b.z, fmadd(a.y, b.y, a.x * b.x)); }
MIW_HD float abs_dot(V3 a, V3 b) { return abs_(dot(a, b)); }
MIW_HD V3 cross(V3 a, V3 b) {
    return v3(fmsub(a.y, b.z, a.z * b.y),
              fmsub(a.z, b.x, a.x * b.z),
              fmsub(a.x, b.y, a.y * b.x));
}
MIW_HD float squared_norm(V3 a) { return dot(a, a); }
MIW_HD float norm(V3 a) { return __builtin_sqrtf(squared_norm(a)); }
MIW_HD V3 normalize(V3 a) { return a * rsqrt(squared_norm(a)); }
MIW_HD float hmax(V3 a) { return max_(max_(a.x, a.y), a.z); }
MIW_HD V3 abs3(V3 a) { return v3(abs_(a.x), abs_(a.y), abs_(a.z)); }
MIW_HD bool all_zero(V3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
// vector fnmadd with scalar b:  -a*b + c  per component (interaction.h:154)
MIW_HD V3 fnmadd3(V3 a, float b, V3 c) {
    return v3(fnmadd(a.x, b, c.x), fnmadd(a.y, b, c.y), fnmadd(a.z, b, c.z));
}
// vector fmsub with scalar b:  a*b - c  (fresnel.h:283)
MIW_HD V3 fmsub3(V3 a, float b, V3 c) {
    return v3(fmsub(a.x, b, c.x), fmsub(a.y, b, c.y), fmsub(a.z, b, c.z));
}

MIW_HD float squared_norm2(V2 p) { return fmadd(p.y, p.y, p.x * p.x); }

// ---- Spectrum type of the compiled variant (include/mitsuba/core/spectrum.h) ---------
// Like the reference, one build = one variant: scalar_rgb carries Color3f (Spec == V3),
// scalar_spectral carries Spectrum<Float, 4> (MIW_SPECTRAL, 4 wavelengths per sample).
#if !defined(MIW_SPECTRAL)
#  define MIW_SPECTRAL 0
#endif
#if MIW_SPECTRAL
#  define MIW_SPEC_N 4
struct Spec { float c[4]; };
MIW_HD Spec spec(float s) { Spec r; r.c[0] = r.c[1] = r.c[2] = r.c[3] = s; return r; }
MIW_HD Spec operator+(Spec a, Spec b) { Spec r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
MIW_HD Spec operator*(Spec a, Spec b) { Spec r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] * b.c[i]; return r; }
MIW_HD Spec operator*(Spec a, float s) { Spec r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] * s; return r; }
MIW_HD Spec operator*(float s, Spec a) { Spec r; for (int i = 0; i < 4; ++i) r.c[i] = s * a.c[i]; return r; }
MIW_HD Spec operator/(Spec a, float s) { float q = rcp(s); Spec r; for (int i = 0; i < 4; ++i) r.c[i] = a.c[i] * q; return r; }
MIW_HD float hmax(Spec a) { return max_(max_(max_(a.c[0], a.c[1]), a.c[2]), a.c[3]); }
MIW_HD bool all_zero(Spec a) { return a.c[0] == 0.f && a.c[1] == 0.f && a.c[2] == 0.f && a.c[3] == 0.f; }
struct Wavelengths { float l[4]; };
#else
#  define MIW_SPEC_N 3
typedef V3 Spec;
MIW_HD Spec spec(float s) { return v3(s); }
struct Wavelengths { };
#endif

// ---- Frame (include/mitsuba/core/frame.h:25-37) -------------------------------
struct Frame { V3 s, t, n; };

MIW_HD V3 to_local(const Frame &f, V3 v) { return v3(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
// s*v.x + t*v.y + n*v.z, plain (non-fused) ops, left to right (frame.h:35-37)
MIW_HD V3 to_world(const Frame &f, V3 v) { return f.s * v.x + f.t * v.y + f.n * v.z; }

// include/mitsuba/core/vector.h:116-136  (Duff et al. orthonormal basis)
MIW_HD void coordinate_system(V3 n, V3 &s, V3 &t) {
    float sign = sign_(n.z),
          a    = -rcp(sign + n.z),
          b    = n.x * n.y * a;
    s = v3(mulsign(sqr(n.x) * a, n.z) + 1.f,
           mulsign(b, n.z),
           mulsign_neg(n.x, n.z));
    t = v3(b, sign + sqr(n.y) * a, -n.y);
}

// frame.h:60 / :107-118
MIW_HD float sin_theta_2(V3 v) { return fmadd(v.x, v.x, sqr(v.y)); }
MIW_HD void sincos_phi(V3 v, float &sin_phi, float &cos_phi) {
    float st2 = sin_theta_2(v), inv_st = rsqrt(st2);
    float rx = v.x * inv_st, ry = v.y * inv_st;
    if (abs_(st2) <= 4.f * MIW_EPSILON) { rx = 1.f; ry = 0.f; }
    else { rx = clamp_(rx, -1.f, 1.f); ry = clamp_(ry, -1.f, 1.f); }
    sin_phi = ry; cos_phi = rx;
}

// ---- 4x4 column-major transforms (include/mitsuba/core/transform.h:90-127) ----
// m[c*4 + r] = column c, row r  (enoki matrix.coeff(c) = column c)
MIW_HD V3 xf_point_persp(const float *m, V3 p) {          // operator*(Point): divides by w
    float r0 = m[12], r1 = m[13], r2 = m[14], r3 = m[15];
    r0 = fmadd(m[0], p.x, r0); r1 = fmadd(m[1], p.x, r1); r2 = fmadd(m[2],  p.x, r2); r3 = fmadd(m[3],  p.x, r3);
    r0 = fmadd(m[4], p.y, r0); r1 = fmadd(m[5], p.y, r1); r2 = fmadd(m[6],  p.y, r2); r3 = fmadd(m[7],  p.y, r3);
    r0 = fmadd(m[8], p.z, r0); r1 = fmadd(m[9], p.z, r1); r2 = fmadd(m[10], p.z, r2); r3 = fmadd(m[11], p.z, r3);
    return v3(r0, r1, r2) / r3;
}
MIW_HD V3 xf_point_affine(const float *m, V3 p) {         // transform_affine(Point)
    float r0 = m[12], r1 = m[13], r2 = m[14];
    r0 = fmadd(m[0], p.x, r0); r1 = fmadd(m[1], p.x, r1); r2 = fmadd(m[2],  p.x, r2);
    r0 = fmadd(m[4], p.y, r0); r1 = fmadd(m[5], p.y, r1); r2 = fmadd(m[6],  p.y, r2);
    r0 = fmadd(m[8], p.z, r0); r1 = fmadd(m[9], p.z, r1); r2 = fmadd(m[10], p.z, r2);
    return v3(r0, r1, r2);
}
MIW_HD V3 xf_vector(const float *m, V3 v) {               // operator*(Vector)
    float r0 = m[0] * v.x, r1 = m[1] * v.x, r2 = m[2] * v.x;
    r0 = fmadd(m[4], v.y, r0); r1 = fmadd(m[5], v.y, r1); r2 = fmadd(m[6],  v.y, r2);
    r0 = fmadd(m[8], v.z, r0); r1 = fmadd(m[9], v.z, r1); r2 = fmadd(m[10], v.z, r2);
    return v3(r0, r1, r2);
}

// include/mitsuba/core/spectrum.h:221-228  (M * rgb, enoki column fma chain)
MIW_HD V3 srgb_to_xyz(V3 c) {
    return v3(fmadd(0.180423f, c.z, fmadd(0.357580f, c.y, 0.412453f * c.x)),
              fmadd(0.072169f, c.z, fmadd(0.715160f, c.y, 0.212671f * c.x)),
              fmadd(0.950227f, c.z, fmadd(0.119193f, c.y, 0.019334f * c.x)));
}
// spectrum.h:231-238
MIW_HD V3 xyz_to_srgb(V3 c) {
    return v3(fmadd(-0.498535f, c.z, fmadd(-1.537150f, c.y,  3.240479f * c.x)),
              fmadd( 0.041556f, c.z, fmadd( 1.875991f, c.y, -0.969256f * c.x)),
              fmadd( 1.057311f, c.z, fmadd(-0.204043f, c.y,  0.055648f * c.x)));
}

} // namespace miw
