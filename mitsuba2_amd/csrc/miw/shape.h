// Triangle intersection, surface-interaction reconstruction and mesh area
// sampling — the Mesh/Shape part of the hot path.
//
// Follows: include/mitsuba/render/mesh.h:194-226 (Moeller-Trumbore, no culling),
// src/librender/mesh.cpp:449-545 (compute_surface_interaction),
// include/mitsuba/render/interaction.h:58-61,153-156,571-596,
// src/librender/mesh.cpp:352-397 (sample_position),
// include/mitsuba/core/distr_1d.h:144-203 (DiscreteDistribution).
#pragma once
#include "base.h"
#include "warp.h"
#include "special.h"

namespace miw {

struct Ray { V3 o, d; float mint, maxt; };

// One triangle as the device stores it (48 B, BVH leaf order). p1/p2 are kept
// (not edges) because compute_surface_interaction interpolates the original
// vertices (mesh.cpp:484); the edges are one exactly-rounded subtraction away.
struct Tri {
    float p0[3], p1[3], p2[3];
    uint32_t shape;      // index into the shape table
    uint32_t prim;       // global primitive id (scene order) — closest-hit tie break
    uint32_t pad;        // 0: a mesh triangle. k > 0: one of the two bounding triangles of analytic rectangle k - 1
                         // (AnalyticRec table): the BVH builders see ordinary triangles, the leaf test runs the
                         // rectangle's own intersection routine (both halves give the same answer)
};

// Analytic shapes. kind 0 = rectangle (src/shapes/rectangle.cpp): [-1, 1]^2 in z = 0 of object space;
// kind 1 = sphere (src/shapes/sphere.cpp). 4x4 matrices column-major.
enum : uint32_t { ANALYTIC_RECTANGLE = 0, ANALYTIC_SPHERE = 1 };
struct AnalyticRec {
    float to_world[16], to_object[16];
    float n[3], inv_area;         // rectangle: m_frame.n; sphere: m_center. m_inv_surface_area
    float dp_du[3]; uint32_t shape;
    float dp_dv[3]; uint32_t prim;
    uint32_t kind; float radius;  // sphere: m_radius
    uint32_t flip;                // sphere: m_flip_normals
    uint32_t pad_;
};

MIW_HD V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

// mesh.h:194-226 from the edge vectors on (e1 = p1 - p0, e2 = p2 - p0: storing them
// is bit-identical to recomputing them — one correctly rounded subtraction each).
MIW_HD bool ray_intersect_triangle_edges(V3 p0, V3 e1, V3 e2, V3 o, V3 d, float mint, float maxt,
                                         float &t_out, float &u_out, float &v_out) {
    V3 pvec = cross(d, e2);
    float inv_det = rcp_loop(dot(e1, pvec));
    V3 tvec = o - p0;
    float u = dot(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    V3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    float t = dot(e2, qvec) * inv_det;
    active = active && t >= mint && t <= maxt;
    t_out = t; u_out = u; v_out = v;
    return active;
}

// mesh.h:194-226. Returns true on a hit inside [mint, maxt].
MIW_HD bool ray_intersect_triangle(V3 p0, V3 p1, V3 p2, V3 o, V3 d, float mint, float maxt,
                                   float &t_out, float &u_out, float &v_out) {
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 pvec = cross(d, e2);
    float inv_det = rcp(dot(e1, pvec));
    V3 tvec = o - p0;
    float u = dot(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    V3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    float t = dot(e2, qvec) * inv_det;
    active = active && t >= mint && t <= maxt;
    t_out = t; u_out = u; v_out = v;
    return active;
}

// Rectangle::ray_intersect_preliminary / ray_test, rectangle.cpp:139-173 (u, v = prim_uv = local x, y)
MIW_HD bool ray_intersect_rectangle(const AnalyticRec &r, V3 o, V3 d, float mint, float maxt,
                                    float &t_out, float &u_out, float &v_out) {
    V3 ol = xf_point_affine(r.to_object, o), dl = xf_vector(r.to_object, d);   // m_to_object.transform_affine(ray_)
    float t = -ol.z * rcp(dl.z);                                                 // -ray.o.z() * ray.d_rcp.z()
    V3 local = v3(fmadd(dl.x, t, ol.x), fmadd(dl.y, t, ol.y), fmadd(dl.z, t, ol.z));   // ray(t)
    t_out = t; u_out = local.x; v_out = local.y;
    return t >= mint && t <= maxt && abs_(local.x) <= 1.f && abs_(local.y) <= 1.f;
}
// ---- the accept rule every scene query shares -------------------------------------------------------------
// Moeller-Trumbore is ill-conditioned for a ray (nearly) inside the triangle's plane: det -> 0, and the t it
// reports can lie anywhere — typically a shadow or bounce ray leaving a surface at grazing angle "re-hits" the face
// it starts on at some t > mint although it has long left the face. Whether such a phantom is found then depends
// on which triangles a query happens to test: brute force tests all of them, a spatial structure (the reference's
// kd-tree, kdtree.h:2079-2171, like any BVH) only those near the ray. At 10^9 samples that is no longer a
// thought experiment (1 sample in 1.3*10^8 on the Cornell box), so the rule is made structure-independent: a
// triangle hit counts iff the point it stands for, o + t*d, lies inside the triangle's bounding box grown by
// `accept_pad` (1e-5 x the largest |coordinate| of the scene; the BVH boxes are grown by twice that, so a hit
// that counts is never culled). Brute force, packet sweep, leaf filter and both tree walks then agree by
// construction; for well-conditioned hits the rule never fires.
struct PrimCtx { const AnalyticRec *rects; float accept_pad; };
struct TriBounds { float lo[3], hi[3]; };                  // bounding box of a triangle's vertices, grown by accept_pad
MIW_HD TriBounds tri_bounds(V3 p0, V3 p1, V3 p2, float pad) {
    TriBounds b;
    b.lo[0] = min_(p0.x, min_(p1.x, p2.x)) - pad; b.hi[0] = max_(p0.x, max_(p1.x, p2.x)) + pad;
    b.lo[1] = min_(p0.y, min_(p1.y, p2.y)) - pad; b.hi[1] = max_(p0.y, max_(p1.y, p2.y)) + pad;
    b.lo[2] = min_(p0.z, min_(p1.z, p2.z)) - pad; b.hi[2] = max_(p0.z, max_(p1.z, p2.z)) + pad;
    return b;
}
// Build switch: -DMIW_ACCEPT_RULE=0 (all libraries and the checker alike) drops the rule — the reference's bare test,
// mesh.h:194-226; results then depend on which triangles a structure happens to test for ill-conditioned grazing rays.
#ifndef MIW_ACCEPT_RULE
#define MIW_ACCEPT_RULE 1
#endif
MIW_HD bool hit_in_bounds(const TriBounds &b, V3 o, V3 d, float t) {
#if !MIW_ACCEPT_RULE
    (void) b; (void) o; (void) d; (void) t;
    return true;
#endif
    const float px = fmadd(d.x, t, o.x), py = fmadd(d.y, t, o.y), pz = fmadd(d.z, t, o.z);
    return px >= b.lo[0] && px <= b.hi[0] && py >= b.lo[1] && py <= b.hi[1] && pz >= b.lo[2] && pz <= b.hi[2];
}
// ---- Sphere (src/shapes/sphere.cpp) --------------------------------------------------------------------------
// math::solve_quadratic in double precision (include/mitsuba/core/math.h:371-413), as the scalar variants run it
MIW_HD bool solve_quadratic_d(double a, double b, double c, double &x0, double &x1) {
    const bool linear_case = a == 0.0, valid_linear = linear_case && b != 0.0;
    x0 = x1 = -c / b;
    const double discrim = __builtin_fma(b, b, -(4.0 * a * c));
    const bool valid_quadratic = !linear_case && discrim >= 0.0;
    if (valid_quadratic) {
        const double sqrt_discrim = __builtin_sqrt(discrim);
        const double temp = -0.5 * (b + __builtin_copysign(sqrt_discrim, b));
        const double x0p = temp / a, x1p = c / temp;
        x0 = x1p < x0p ? x1p : x0p;                       // min / max, std semantics
        x1 = x0p < x1p ? x1p : x0p;
    }
    return valid_linear || valid_quadratic;
}
// Sphere::ray_intersect_preliminary / ray_test, sphere.cpp:281-336 (u = v = 0: the sphere reports no prim_uv)
MIW_HD bool ray_intersect_sphere(const AnalyticRec &r, V3 o_, V3 d_, float mint_, float maxt_,
                                 float &t_out, float &u_out, float &v_out) {
    const double mint = (double) mint_, maxt = (double) maxt_;
    const double ox = (double) o_.x - (double) r.n[0], oy = (double) o_.y - (double) r.n[1], oz = (double) o_.z - (double) r.n[2];
    const double dx = d_.x, dy = d_.y, dz = d_.z;
    // squared_norm / dot: enoki fma chains, fma(z, z, fma(y, y, x * x))
    const double A = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
    const double B = 2.0 * __builtin_fma(oz, dz, __builtin_fma(oy, dy, ox * dx));
    const double rad = (double) r.radius;
    const double C = __builtin_fma(oz, oz, __builtin_fma(oy, oy, ox * ox)) - rad * rad;
    double near_t, far_t;
    const bool found = solve_quadratic_d(A, B, C, near_t, far_t);
    const bool out_bounds = !(near_t <= maxt && far_t >= mint);     // NaN-aware
    const bool in_bounds = near_t < mint && far_t > maxt;
    t_out = near_t < mint ? (float) far_t : (float) near_t;
    u_out = v_out = 0.f;
    return found && !out_bounds && !in_bounds;
}

// the leaf test: Mesh::ray_intersect_triangle (+ the rule above) or the analytic shape's own routine
// (kdtree.h:2362-2391 intersect_prim). Analytic = false: the caller knows the scene holds triangles only.
template <bool Analytic = true>
MIW_HD bool prim_intersect(const Tri &tr, PrimCtx ctx, V3 o, V3 d, float mint, float maxt,
                           float &t, float &u, float &v) {
    if (Analytic && tr.pad) {
        const AnalyticRec &a = ctx.rects[tr.pad - 1u];
        return a.kind == ANALYTIC_SPHERE ? ray_intersect_sphere(a, o, d, mint, maxt, t, u, v)
                                         : ray_intersect_rectangle(a, o, d, mint, maxt, t, u, v);
    }
    const V3 p0 = ld3(tr.p0), p1 = ld3(tr.p1), p2 = ld3(tr.p2);
    return ray_intersect_triangle(p0, p1, p2, o, d, mint, maxt, t, u, v) &&
           hit_in_bounds(tri_bounds(p0, p1, p2, ctx.accept_pad), o, d, t);
}

// What the shading stage needs of a SurfaceInteraction3f (interaction.h).
struct SurfaceInteraction {
    float t;
    V3 p, n;          // position, geometric normal
    Frame sh;         // shading frame
    V3 wi;            // incident direction, local frame
    V2 uv;
    uint32_t shape, prim;
};

// mesh.cpp:449-545 + interaction.h:571-596.
// `vn` = per-vertex normals of this face or nullptr (mesh.cpp:514-519);
// `tc` = per-vertex texture coordinates of this face (u0 v0 u1 v1 u2 v2) or nullptr (:492-511): they replace the
// barycentric uv and, where the uv parameterisation is not degenerate, the tangents the shading frame is built on.
MIW_HD void compute_surface_interaction(V3 p0, V3 p1, V3 p2, const float *vn, const float *tc,
                                        float t, float b1, float b2, V3 ray_d,
                                        SurfaceInteraction &si) {
    float b0 = 1.f - b1 - b2;
    V3 dp0 = p1 - p0, dp1 = p2 - p0;
    si.t = t;
    si.p = p0 * b0 + p1 * b1 + p2 * b2;                // mesh.cpp:484
    si.n = normalize(cross(dp0, dp1));                 // :487
    si.uv = v2(b1, b2);                                // :490
    V3 dp_du, dp_dv;
    coordinate_system(si.n, dp_du, dp_dv);             // :491
    if (tc) {                                          // :492-511
        const V2 uv0 = v2(tc[0], tc[1]), uv1 = v2(tc[2], tc[3]), uv2 = v2(tc[4], tc[5]);
        si.uv = v2(uv0.x * b0 + uv1.x * b1 + uv2.x * b2, uv0.y * b0 + uv1.y * b1 + uv2.y * b2);   // :497
        const V2 duv0 = v2(uv1.x - uv0.x, uv1.y - uv0.y), duv1 = v2(uv2.x - uv0.x, uv2.y - uv0.y);
        const float det = fmsub(duv0.x, duv1.y, duv0.y * duv1.x), inv_det = rcp(det);             // :503-504
        if (det != 0.f) {                                                                          // :506-509
            dp_du = fmsub3(dp0, duv1.y, dp1 * duv0.y) * inv_det;
            dp_dv = fnmadd3(dp0, duv1.x, dp1 * duv0.x) * inv_det;
        }
    }
    if (vn) {                                          // :514-519
        V3 n0 = ld3(vn), n1 = ld3(vn + 3), n2 = ld3(vn + 6);
        si.sh.n = normalize(n0 * b0 + n1 * b1 + n2 * b2);
    } else {
        si.sh.n = si.n;                                // :541
    }
    // initialize_sh_frame, interaction.h:153-156
    si.sh.s = normalize(fnmadd3(si.sh.n, dot(si.sh.n, dp_du), dp_du));
    si.sh.t = cross(si.sh.n, si.sh.s);
    si.wi = to_local(si.sh, -ray_d);                   // interaction.h:591
}
MIW_HD void compute_surface_interaction(V3 p0, V3 p1, V3 p2, const float *vn, float t, float b1, float b2, V3 ray_d,
                                        SurfaceInteraction &si) {
    compute_surface_interaction(p0, p1, p2, vn, nullptr, t, b1, b2, ray_d, si);
}

// Rectangle::compute_surface_interaction, rectangle.cpp:175-208 + interaction.h:571-596
MIW_HD void compute_surface_interaction_rect(const AnalyticRec &r, float t, float u, float v, V3 ray_o, V3 ray_d,
                                             SurfaceInteraction &si) {
    si.t = t;
    si.p = v3(fmadd(ray_d.x, t, ray_o.x), fmadd(ray_d.y, t, ray_o.y), fmadd(ray_d.z, t, ray_o.z));   // ray(pi.t), :196
    si.n = ld3(r.n); si.sh.n = si.n;                   // :198-199
    V3 dp_du = ld3(r.dp_du);                           // :200
    si.uv = v2(fmadd(u, .5f, .5f), fmadd(v, .5f, .5f));   // :202-203
    si.sh.s = normalize(fnmadd3(si.sh.n, dot(si.sh.n, dp_du), dp_du));   // initialize_sh_frame, interaction.h:153-156
    si.sh.t = cross(si.sh.n, si.sh.s);
    si.wi = to_local(si.sh, -ray_d);                   // interaction.h:591
}

// enoki's unit_angle_z (not vendored in the reference checkout; frozen here): the angle between v and +z,
// 2 asin(|v - sign(z) e_z| / 2), mirrored for z < 0
MIW_HD float unit_angle_z(V3 v) {
    float temp = 2.f * asin_(.5f * __builtin_sqrtf(sqr(v.x) + sqr(v.y) + sqr(v.z - mulsign(1.f, v.z))));
    return v.z >= 0.f ? temp : MIW_PI - temp;
}
// Sphere::compute_surface_interaction, sphere.cpp:338-402 + interaction.h:571-596
MIW_HD void compute_surface_interaction_sphere(const AnalyticRec &r, float t, V3 ray_o, V3 ray_d, SurfaceInteraction &si) {
    const V3 center = ld3(r.n);
    si.t = t;
    V3 n = normalize(v3(fmadd(ray_d.x, t, ray_o.x), fmadd(ray_d.y, t, ray_o.y), fmadd(ray_d.z, t, ray_o.z)) - center);   // :359
    si.p = v3(fmadd(n.x, r.radius, center.x), fmadd(n.y, r.radius, center.y), fmadd(n.z, r.radius, center.z));            // :362
    V3 local = xf_point_affine(r.to_object, si.p);         // :365
    float phi = atan2_(local.y, local.x);
    if (phi < 0.f) phi += 2.f * MIW_PI;
    si.uv = v2(phi * (.5f * MIW_INV_PI), unit_angle_z(local) * MIW_INV_PI);                                               // :371-373
    V3 dp_du = xf_vector(r.to_world, v3(-local.y, local.x, 0.f)) * (2.f * MIW_PI);                                        // :375, :391
    if (r.flip) n = -n;                                    // :396-397
    si.sh.n = n; si.n = n;
    si.sh.s = normalize(fnmadd3(si.sh.n, dot(si.sh.n, dp_du), dp_du));   // initialize_sh_frame, interaction.h:153-156
    si.sh.t = cross(si.sh.n, si.sh.s);
    si.wi = to_local(si.sh, -ray_d);
}

// interaction.h:58-61 — (1 + hmax(abs(p))) * RayEpsilon
MIW_HD float spawn_mint(V3 p) { return (1.f + hmax(abs3(p))) * MIW_RAY_EPSILON; }

// ---- area sampling of an emitter mesh -----------------------------------------
// Per-emitter tables built on the host exactly as DiscreteDistribution does
// (distr_1d.h:55-87: CDF accumulated in double, stored float).
struct MeshSampler {
    const float *tri;     // 9 floats per face: p0,p1,p2 (face order of the mesh)
    const float *vnorm;   // 9 floats per face or nullptr
    const float *pmf;     // face areas (unnormalized)
    const float *cdf;     // running sum (unnormalized)
    uint32_t count;       // number of faces
    uint32_t valid_lo, valid_hi; // first / last face with non-zero area
    float sum, normalization;
};

// distr_1d.h:144-154: first index in [valid_lo, valid_hi] with cdf[idx] >= value
MIW_HD uint32_t distr_sample(const MeshSampler &m, float value) {
    value *= m.sum;
    uint32_t lo = m.valid_lo, hi = m.valid_hi;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (m.cdf[mid] < value) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct PositionSample { V3 p, n; V2 uv; float pdf; };

// Rectangle::sample_position, rectangle.cpp:111-125
MIW_HD PositionSample rect_sample_position(const AnalyticRec &r, V2 sample) {
    PositionSample ps;
    ps.p = xf_point_affine(r.to_world, v3(sample.x * 2.f - 1.f, sample.y * 2.f - 1.f, 0.f));
    ps.n = ld3(r.n); ps.pdf = r.inv_area; ps.uv = sample;
    return ps;
}

// mesh.cpp:352-397
MIW_HD PositionSample mesh_sample_position(const MeshSampler &m, V2 sample) {
    // sample_reuse, distr_1d.h:193-203
    uint32_t idx = distr_sample(m, sample.y);
    float pmf = m.pmf[idx] * m.normalization,
          cdf = idx > 0 ? m.cdf[idx - 1] * m.normalization : 0.f;
    sample.y = (sample.y - cdf) / pmf;

    const float *f = m.tri + 9 * (size_t) idx;
    V3 p0 = ld3(f), p1 = ld3(f + 3), p2 = ld3(f + 6);
    V3 e0 = p1 - p0, e1 = p2 - p0;
    V2 b = square_to_uniform_triangle(sample);

    PositionSample ps;
    ps.p = p0 + e0 * b.x + e1 * b.y;
    ps.pdf = m.normalization;
    ps.uv = b;
    if (m.vnorm) {
        const float *vn = m.vnorm + 9 * (size_t) idx;
        V3 n0 = ld3(vn), n1 = ld3(vn + 3), n2 = ld3(vn + 6);
        ps.n = normalize(n0 * (1.f - b.x - b.y) + n1 * b.x + n2 * b.y);
    } else {
        ps.n = normalize(cross(e0, e1));
    }
    return ps;
}

// mesh.h:107-117
MIW_HD float face_area(V3 p0, V3 p1, V3 p2) { return 0.5f * norm(cross(p1 - p0, p2 - p0)); }

} // namespace miw
