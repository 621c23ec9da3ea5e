// Host-side construction of the analytic rectangle's device record — Rectangle::update(), src/shapes/rectangle.cpp:86-96
// — and of the two bounding triangles the BVH builders see in its place (corners of bbox(), :98-105).
// Shared by the device library's uploader and the CPU checkers (like bvh_build.h / envmap_build.h).
#pragma once
#include <cstring>
#include "miw/shape.h"

namespace miw {

// to_world / to_object: 4x4 column-major (Transform4f::matrix and its inverse)
inline AnalyticRec rect_record(const float *to_world, const float *to_object, uint32_t shape, uint32_t prim) {
    AnalyticRec r;
    std::memcpy(r.to_world, to_world, 64); std::memcpy(r.to_object, to_object, 64);
    const V3 dp_du = xf_vector(to_world, v3(2.f, 0.f, 0.f)), dp_dv = xf_vector(to_world, v3(0.f, 2.f, 0.f));   // :89-90
    // m_to_world * Normal3f(0, 0, 1): columns of the inverse transpose = rows of the inverse, transform.h:134-142
    const V3 r0 = v3(to_object[0], to_object[4], to_object[8]), r1 = v3(to_object[1], to_object[5], to_object[9]),
             r2 = v3(to_object[2], to_object[6], to_object[10]);
    V3 n = r0 * 0.f;
    n = v3(fmadd(r1.x, 0.f, n.x), fmadd(r1.y, 0.f, n.y), fmadd(r1.z, 0.f, n.z));
    n = v3(fmadd(r2.x, 1.f, n.x), fmadd(r2.y, 1.f, n.y), fmadd(r2.z, 1.f, n.z));
    n = normalize(n);                                                            // :91
    r.n[0] = n.x; r.n[1] = n.y; r.n[2] = n.z;
    r.dp_du[0] = dp_du.x; r.dp_du[1] = dp_du.y; r.dp_du[2] = dp_du.z;
    r.dp_dv[0] = dp_dv.x; r.dp_dv[1] = dp_dv.y; r.dp_dv[2] = dp_dv.z;
    r.inv_area = rcp(norm(cross(dp_du, dp_dv)));                                 // :94, surface_area() :103-105
    r.shape = shape; r.prim = prim;
    r.kind = ANALYTIC_RECTANGLE; r.radius = 0.f; r.flip = 0; r.pad_ = 0;
    return r;
}

// the two triangles (A, B, C), (A, C, D) over the corners (-1,-1), (1,-1), (1,1), (-1,1): same bounds as bbox()
inline void rect_bounding_tris(const AnalyticRec &r, uint32_t rect_index, Tri out[2]) {
    const V3 a = xf_point_affine(r.to_world, v3(-1.f, -1.f, 0.f)), b = xf_point_affine(r.to_world, v3(1.f, -1.f, 0.f)),
             c = xf_point_affine(r.to_world, v3(1.f, 1.f, 0.f)), d = xf_point_affine(r.to_world, v3(-1.f, 1.f, 0.f));
    auto put = [](float *dst, V3 p) { dst[0] = p.x; dst[1] = p.y; dst[2] = p.z; };
    for (int k = 0; k < 2; ++k) { out[k].shape = r.shape; out[k].prim = r.prim; out[k].pad = rect_index + 1u; }
    put(out[0].p0, a); put(out[0].p1, b); put(out[0].p2, c);
    put(out[1].p0, a); put(out[1].p1, c); put(out[1].p2, d);
}

// Sphere::update(), src/shapes/sphere.cpp:108-131: center, radius, flip, the rebuilt to_world (uniform scale,
// rotation, translation) and its inverse come from the host (transform_decompose / transform_compose).
inline AnalyticRec sphere_record(const float *center, float radius, bool flip, const float *to_world, const float *to_object,
                             uint32_t shape, uint32_t prim) {
    AnalyticRec r;
    std::memset(&r, 0, sizeof r);
    std::memcpy(r.to_world, to_world, 64); std::memcpy(r.to_object, to_object, 64);
    r.n[0] = center[0]; r.n[1] = center[1]; r.n[2] = center[2];
    r.inv_area = rcp(4.f * MIW_PI * radius * radius);                           // :131, surface_area() :141-143
    r.shape = shape; r.prim = prim; r.kind = ANALYTIC_SPHERE; r.radius = radius; r.flip = flip ? 1u : 0u;
    return r;
}
// two triangles that both span bbox() = center -+ radius (:133-139) corner to corner: every leaf box holding one of
// them contains the whole sphere
inline void sphere_bounding_tris(const AnalyticRec &r, uint32_t index, Tri out[2]) {
    const float lo[3] = { r.n[0] - r.radius, r.n[1] - r.radius, r.n[2] - r.radius },
                hi[3] = { r.n[0] + r.radius, r.n[1] + r.radius, r.n[2] + r.radius };
    for (int k = 0; k < 2; ++k) {
        out[k].shape = r.shape; out[k].prim = r.prim; out[k].pad = index + 1u;
        std::memcpy(out[k].p0, lo, 12); std::memcpy(out[k].p1, hi, 12);
    }
    out[0].p2[0] = lo[0]; out[0].p2[1] = hi[1]; out[0].p2[2] = lo[2];
    out[1].p2[0] = hi[0]; out[1].p2[1] = lo[1]; out[1].p2[2] = hi[2];
}
inline void analytic_bounding_tris(const AnalyticRec &r, uint32_t index, Tri out[2]) {
    if (r.kind == ANALYTIC_SPHERE) sphere_bounding_tris(r, index, out); else rect_bounding_tris(r, index, out);
}

} // namespace miw
