"""Independent restatements of the float leaves the oracle SHARES with the product.

oracle/miw_oracle.cpp has its own control flow (path loop, scene queries, splat, spiral) but includes the product's leaf
headers (csrc/miw/{bsdf,scene,shape,warp,...}.h): a transcription slip inside one of them would pass every device-vs-oracle
test by construction. This file closes that gap for the leaves that carry the arithmetic of BASELINE configs 2 - 4: each
function below is written a SECOND time, in float64 numpy, straight from the reference source it cites — sharing no code with
csrc/ — and compared with what the oracle's leaf entry points return for random inputs. Float64 against the float32 of the
headers: agreement to ~1e-5 relative wherever the function is well-conditioned (the tolerance of each check is next to it;
samples that sit on a branch boundary of the float32 evaluation — lobe choice, quadrant choice, CDF bin — are excluded by an
explicit margin, never by the comparison's outcome).
"""
import math

import numpy as np
import pytest

PI = math.pi


# ---------------------------------------------------------------- warp.h
def disk_concentric(u):
    """core/warp.h:54-90"""
    x, y = 2.0 * u[0] - 1.0, 2.0 * u[1] - 1.0
    if x == 0 and y == 0:
        return 0.0, 0.0
    if abs(x) < abs(y):
        r, phi = y, 0.5 * PI - 0.25 * PI * x / y
    else:
        r, phi = x, 0.25 * PI * y / x
    return r * math.cos(phi), r * math.sin(phi)


def cosine_hemisphere(u):
    """core/warp.h:325-334"""
    px, py = disk_concentric(u)
    return np.array([px, py, math.sqrt(max(0.0, 1.0 - px * px - py * py))])


# ---------------------------------------------------------------- fresnel.h
def fresnel(cos_i, eta):
    """render/fresnel.h:34-70 -> r, cos_theta_t, eta_it, eta_ti"""
    outside = cos_i >= 0
    eta_it, eta_ti = (eta, 1 / eta) if outside else (1 / eta, eta)
    cos_t_sqr = 1 - (1 - cos_i * cos_i) * eta_ti * eta_ti
    ci, ct = abs(cos_i), math.sqrt(max(0.0, cos_t_sqr))
    if eta == 1:
        r = 0.0
    elif ci == 0:
        r = 1.0
    else:
        a_s = (ci - eta_it * ct) / (ci + eta_it * ct)
        a_p = (ct - eta_it * ci) / (ct + eta_it * ci)
        r = 0.5 * (a_s * a_s + a_p * a_p)
    return r, (-ct if cos_i >= 0 else ct), eta_it, eta_ti          # mulsign_neg(cos_theta_t_abs, cos_theta_i)


def fresnel_conductor(cos_i, eta, k):
    """render/fresnel.h:92-116 (per channel)"""
    c2 = cos_i * cos_i
    s2 = 1 - c2
    s4 = s2 * s2
    t1 = eta * eta - k * k - s2
    a2pb2 = np.sqrt(np.maximum(0, t1 * t1 + 4 * k * k * eta * eta))
    a = np.sqrt(np.maximum(0, 0.5 * (a2pb2 + t1)))
    term1, term2 = a2pb2 + c2, 2 * cos_i * a
    rs = (term1 - term2) / (term1 + term2)
    term3, term4 = a2pb2 * c2 + s4, term2 * s2
    rp = rs * (term3 - term4) / (term3 + term4)
    return 0.5 * (rs + rp)


# ---------------------------------------------------------------- microfacet.h
class Microfacet:
    def __init__(self, kind, au, av, visible):
        self.kind, self.visible = kind, visible
        self.au, self.av = max(au, 1e-4), max(av, 1e-4)          # configure(), microfacet.h:415-418

    def eval(self, m):                                           # :184-202
        c = m[2]
        c2 = c * c
        if self.kind == "beckmann":
            res = math.exp(-((m[0] / self.au) ** 2 + (m[1] / self.av) ** 2) / c2) / (PI * self.au * self.av * c2 * c2)
        else:
            res = 1 / (PI * self.au * self.av * ((m[0] / self.au) ** 2 + (m[1] / self.av) ** 2 + m[2] ** 2) ** 2)
        return res if res * c > 1e-20 else 0.0

    def g1(self, v, m):                                          # :331-355
        xy = (self.au * v[0]) ** 2 + (self.av * v[1]) ** 2
        if xy == 0:
            res = 1.0
        else:
            t2 = xy / (v[2] * v[2])
            if self.kind == "beckmann":
                a = 1 / math.sqrt(t2)
                res = 1.0 if a >= 1.6 else (3.535 * a + 2.181 * a * a) / (1 + 2.276 * a + 2.577 * a * a)
            else:
                res = 2 / (1 + math.sqrt(1 + t2))
        return 0.0 if np.dot(v, m) * v[2] <= 0 else res

    def G(self, wi, wo, m):
        return self.g1(wi, m) * self.g1(wo, m)

    def pdf(self, wi, m):                                        # :214-223
        if self.visible:
            return self.eval(m) * self.g1(wi, m) * abs(np.dot(wi, m)) / wi[2]
        return self.eval(m) * m[2]

    def sample_visible_11_ggx(self, cos_i, u):                   # :395-411
        px, py = disk_concentric(u)
        s = 0.5 * (1 + cos_i)
        py = (1 - s) * math.sqrt(max(0.0, 1 - px * px)) + s * py
        z = math.sqrt(max(0.0, 1 - px * px - py * py))
        sin_i = math.sqrt(max(0.0, 1 - cos_i * cos_i))
        norm = 1 / (sin_i * py + cos_i * z)
        return (cos_i * py - sin_i * z) * norm, px * norm

    def sample(self, wi, u):                                     # :234-316 (GGX; Beckmann's visible branch is table-tested)
        if not self.visible:
            if self.au == self.av:
                sin_phi, cos_phi = math.sin(2 * PI * u[1]), math.cos(2 * PI * u[1])
                a2 = self.au * self.au
            else:
                tmp = self.av / self.au * math.tan(2 * PI * u[1])
                cos_phi = 1 / math.sqrt(tmp * tmp + 1)
                cos_phi = math.copysign(cos_phi, abs(u[1] - 0.5) - 0.25)
                sin_phi = cos_phi * tmp
                a2 = 1 / ((cos_phi / self.au) ** 2 + (sin_phi / self.av) ** 2)
            if self.kind == "beckmann":
                cos_t = 1 / math.sqrt(1 - a2 * math.log(1 - u[0]))
                pdf = (1 - u[0]) / (PI * self.au * self.av * max(cos_t ** 3, 1e-20))
            else:
                tan2 = a2 * u[0] / (1 - u[0])
                cos_t = 1 / math.sqrt(1 + tan2)
                pdf = 1 / (PI * self.au * self.av * max(cos_t ** 3, 1e-20) * (1 + tan2 / a2) ** 2)
            sin_t = math.sqrt(1 - cos_t * cos_t)
            return np.array([cos_phi * sin_t, sin_phi * sin_t, cos_t]), pdf
        wp = np.array([self.au * wi[0], self.av * wi[1], wi[2]])
        wp /= np.linalg.norm(wp)
        sin_t = math.sqrt(max(0.0, 1 - wp[2] * wp[2]))                          # Frame::sincos_phi, frame.h
        sin_phi, cos_phi = (wp[1] / sin_t, wp[0] / sin_t) if sin_t > 1e-12 else (0.0, 1.0)
        sx, sy = self.sample_visible_11_ggx(wp[2], u)
        sx, sy = (cos_phi * sx - sin_phi * sy) * self.au, (sin_phi * sx + cos_phi * sy) * self.av
        m = np.array([-sx, -sy, 1.0])
        m /= np.linalg.norm(m)
        return m, self.eval(m) * self.g1(wi, m) * abs(np.dot(wi, m)) / wi[2]


# ---------------------------------------------------------------- bsdfs
def roughconductor(d, eta, k, wi, u2, wo_eval):
    """roughconductor.cpp:196-275 (sample), :277-345 (eval), :347-382 (pdf) -> (wo, pdf, weight[3]), eval[3], pdf"""
    zero = np.zeros(3)
    smp = (zero, 0.0, zero)
    if wi[2] > 0:
        m, pdf = d.sample(wi, u2)
        wo = 2 * np.dot(wi, m) * m - wi
        if pdf != 0 and wo[2] > 0:
            w = d.g1(wo, m) if d.visible else d.G(wi, wo, m) * np.dot(wi, m) / (wi[2] * m[2])
            smp = (wo, pdf / (4 * np.dot(wo, m)), fresnel_conductor(np.dot(wi, m), eta, k) * w)
        else:
            smp = (wo, pdf / (4 * np.dot(wo, m)) if np.dot(wo, m) != 0 else 0.0, zero)
    ev, pd = zero, 0.0
    if wi[2] > 0 and wo_eval[2] > 0:
        h = wo_eval + wi
        h /= np.linalg.norm(h)
        D = d.eval(h)
        if D != 0:
            ev = fresnel_conductor(np.dot(wi, h), eta, k) * D * d.G(wi, wo_eval, h) / (4 * wi[2])
        if np.dot(wi, h) > 0 and np.dot(wo_eval, h) > 0:
            pd = D * d.g1(wi, h) / (4 * wi[2]) if d.visible else d.pdf(wi, h) / (4 * np.dot(wo_eval, h))
    return smp, ev, pd


def dielectric_sample(eta, wi, s1):
    """dielectric.cpp:201-310, unpolarised, both lobes enabled, TransportMode::Radiance -> wo, pdf, eta, weight"""
    r, cos_t, eta_it, eta_ti = fresnel(wi[2], eta)
    if s1 <= r:
        return np.array([-wi[0], -wi[1], wi[2]]), r, 1.0, 1.0, r
    return np.array([-eta_ti * wi[0], -eta_ti * wi[1], cos_t]), 1 - r, eta_it, eta_ti * eta_ti, r


def _bsdf_inputs(rng, n, upper=True):
    wi = rng.normal(size=(n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    if upper:
        wi[:, 2] = np.abs(wi[:, 2]); wo[:, 2] = np.abs(wo[:, 2])
    x = np.zeros((n, 10), np.float32)
    x[:, 1:4] = wi; x[:, 4:7] = rng.random((n, 3)); x[:, 7:10] = wo
    return x


def _one_bsdf_scene(native, bsdf):
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    return native.Scene([native.Mesh("t", v, np.array([[0, 1, 2]], np.uint32), bsdf=bsdf)]).build(-1)


def _close(a, b, rtol, atol=1e-7):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= atol + rtol * np.maximum(np.abs(a), np.abs(b)))


@pytest.mark.parametrize("kw", [
    dict(distribution="ggx", alpha=0.1),                                   # BASELINE config 3's metal
    dict(distribution="ggx", alpha=0.35, sample_visible=False),
    dict(distribution="ggx", alpha_u=0.4, alpha_v=0.12),
    dict(distribution="ggx", alpha_u=0.15, alpha_v=0.3, sample_visible=False),
    dict(distribution="beckmann", alpha=0.3, sample_visible=False),
    dict(distribution="beckmann", alpha_u=0.2, alpha_v=0.45, sample_visible=False),
])
def test_roughconductor_against_float64_restatement(native, oracle, kw):
    """sample / eval / pdf of roughconductor.cpp + microfacet.h + fresnel_conductor, every distribution branch the configs
    can reach except Beckmann's visible-normal inversion (Newton iterations on erf: pinned by the reference's own sample table,
    test_oracle_kat.py, and by chi-square)."""
    eta, k = np.array([0.2, 0.92, 1.1]), np.array([3.9, 2.45, 2.14])
    scene = _one_bsdf_scene(native, native.BSDF("roughconductor", eta=tuple(eta), k=tuple(k), **kw))
    x = _bsdf_inputs(np.random.default_rng(11), 600)
    up = np.array([[0.3, 0.2, 0.93]]) / np.linalg.norm([0.3, 0.2, 0.93])
    x[:, 1:4] = np.where(x[:, 3:4] < 0.05, up.astype(np.float32), x[:, 1:4])                       # keep wi off the horizon
    out = oracle.eval(3, x, scene.desc())
    au, av = kw.get("alpha_u", kw.get("alpha")), kw.get("alpha_v", kw.get("alpha"))
    d = Microfacet(kw["distribution"], au, av, kw.get("sample_visible", True))
    checked = 0
    for xi, o in zip(x.astype(np.float64), out):
        wi, u2, wo = xi[1:4], xi[5:7], xi[7:10]
        if min(u2[0], 1 - u2[0], abs(abs(2 * u2[0] - 1) - abs(2 * u2[1] - 1))) < 1e-3 or abs(abs(u2[1] - 0.5) - 0.25) < 1e-3:
            continue                                             # the concentric map's quadrant seam / tan()'s poles
        (s_wo, s_pdf, s_w), ev, pd = roughconductor(d, eta, k, wi, u2, wo)
        assert _close(o[9:12], ev, 2e-4, 1e-6) and _close(o[12], pd, 2e-4, 1e-6), (kw, wi, wo, o[9:13], ev, pd)
        if s_pdf > 1e-3 and s_wo[2] > 1e-3:                      # well-conditioned samples: direction, density, weight
            assert _close(o[0:3], s_wo, 0, 2e-4) and _close(o[3], s_pdf, 1e-3) and _close(o[6:9], s_w, 5e-4, 1e-6), (kw, wi, u2, o[:9], s_wo, s_pdf, s_w)
            assert o[4] == 1.0
            checked += 1
    assert checked > 300


def test_dielectric_and_diffuse_against_float64_restatement(native, oracle):
    """dielectric.cpp:201-310 with fresnel.h:34-70 / :275-294 (config 3's bk7 glass, from both sides) and diffuse.cpp:78-135."""
    eta = 1.5046 / 1.000277
    scene = _one_bsdf_scene(native, native.BSDF("dielectric", int_ior=1.5046, ext_ior=1.000277))
    x = _bsdf_inputs(np.random.default_rng(3), 800, upper=False)
    out = oracle.eval(3, x, scene.desc())
    n_t = 0
    for xi, o in zip(x.astype(np.float64), out):
        wi = xi[1:4]
        wo, pdf, bs_eta, w, r = dielectric_sample(eta, wi, xi[4])
        if abs(xi[4] - r) < 1e-4:
            continue                                             # lobe choice on the float32 knife edge
        assert _close(o[0:3], wo, 0, 3e-6) and _close(o[3], pdf, 2e-5) and _close(o[4], bs_eta, 2e-6) and _close(o[6:9], [w] * 3, 2e-6), (wi, xi[4], o[:9])
        assert not o[9:13].any()                                 # eval = pdf = 0 (delta lobes), dielectric.cpp:312-320
        n_t += bs_eta != 1.0
    assert n_t > 200
    refl = np.array([0.63, 0.065, 0.05])
    scene = _one_bsdf_scene(native, native.BSDF("diffuse", reflectance=tuple(refl)))
    x = _bsdf_inputs(np.random.default_rng(4), 600, upper=False)
    out = oracle.eval(3, x, scene.desc())
    for xi, o in zip(x.astype(np.float64), out):
        wi, wo = xi[1:4], xi[7:10]
        both = wi[2] > 0 and wo[2] > 0
        assert _close(o[9:12], refl / PI * wo[2] if both else np.zeros(3), 2e-6)          # eval, :104-119
        assert _close(o[12], wo[2] / PI if both else 0.0, 2e-6)                           # pdf, :121-135
        u2 = xi[5:7]
        if wi[2] > 0 and abs(abs(2 * u2[0] - 1) - abs(2 * u2[1] - 1)) > 1e-3:
            s = cosine_hemisphere(u2)
            assert _close(o[0:3], s, 0, 3e-6) and _close(o[3], s[2] / PI, 1e-5, 2e-6) and _close(o[6:9], refl, 1e-6)   # sample, :78-102 (z = sqrt(1 - r^2) cancels near the horizon)


def test_area_light_sampling_against_float64_restatement(native, oracle):
    """Scene::sample_emitter_direction (scene.cpp:164-200, one emitter) -> AreaLight::sample_direction (area.cpp:121-166)
    -> Shape::sample_direction (shape.cpp:292-309) -> Mesh::sample_position (mesh.cpp:352-397: face by area through
    DiscreteDistribution::sample_reuse, distr_1d.h:193-203, then square_to_uniform_triangle, warp.h:153-156), and
    pdf_direction (area.cpp:168-187, shape.cpp:311-323) on a three-face emitter with unequal areas."""
    v = np.array([[0, 2, 0], [1, 2, 0], [1, 2, 1], [0, 2, 1], [3, 2, 0.5], [0.5, 2, 3]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3], [2, 4, 5]], np.uint32)          # normals point down (-y)
    radiance = (17.0, 12.0, 4.0)
    light = native.Mesh("light", v, f, emitter=native.AreaLight(radiance))
    floor = native.Mesh("floor", np.array([[-5, 0, -5], [5, 0, -5], [5, 0, 5], [-5, 0, 5]], np.float32),
                        np.array([[0, 2, 1], [0, 3, 2]], np.uint32), bsdf=native.BSDF("diffuse", reflectance=(0.5, 0.5, 0.5)))
    scene = native.Scene([floor, light]).build(-1)
    rng = np.random.default_rng(8)
    n = 500
    ref = np.concatenate([rng.uniform(-3, 3, (n, 1)), rng.uniform(-1, 3.5, (n, 1)), rng.uniform(-3, 3, (n, 1)), rng.random((n, 2))], 1).astype(np.float32)
    out = oracle.eval(6, ref, scene.desc())
    P = v.astype(np.float64)
    areas = np.array([0.5 * np.linalg.norm(np.cross(P[b] - P[a], P[c] - P[a])) for a, b, c in f])
    cdf = np.cumsum(areas) / areas.sum()
    checked = 0
    for r, o in zip(ref.astype(np.float64), out):
        p_ref, u = r[0:3], r[3:5]
        face = int(np.searchsorted(cdf, u[1], side="left")) if u[1] > 0 else 0          # first index with cdf >= value
        face = min(face, 2)
        lo = cdf[face - 1] if face else 0.0
        if min(abs(u[1] - cdf[face]), abs(u[1] - lo)) < 1e-5:
            continue                                              # CDF bin edge
        u1 = (u[1] - lo) / (areas[face] / areas.sum())            # sample_reuse
        t = math.sqrt(max(0.0, 1 - u[0]))
        b = (1 - t, t * u1)                                       # square_to_uniform_triangle
        a_, b_, c_ = f[face]
        p = P[a_] + (P[b_] - P[a_]) * b[0] + (P[c_] - P[a_]) * b[1]
        nrm = np.cross(P[b_] - P[a_], P[c_] - P[a_]); nrm /= np.linalg.norm(nrm)
        d = p - p_ref
        dist2 = d @ d
        dist = math.sqrt(dist2)
        d /= dist
        dp = abs(d @ nrm)
        pdf = (1 / areas.sum()) * (dist2 / dp if dp != 0 else 0.0)
        active = d @ nrm < 0 and pdf != 0
        if dp < 1e-3:
            continue                                              # grazing: pdf ill-conditioned
        assert _close(o[0:3], d, 0, 3e-6) and _close(o[3], dist, 3e-6) and _close(o[4], pdf, 1e-4) and _close(o[5:8], p, 0, 3e-6) and _close(o[8:11], nrm, 0, 1e-6), (r, o)
        assert _close(o[11:14], np.array(radiance) / pdf if active else np.zeros(3), 1e-4, 1e-9)
        checked += 1
    assert checked > 400


def test_surface_interaction_against_float64_restatement(native, oracle):
    """Mesh::compute_surface_interaction (mesh.cpp:449-545) with vertex normals and texture coordinates (position from the
    barycentrics, dp_du from the uv Jacobian), the Gram-Schmidt shading frame of interaction.h:153-156 and wi = to_local(-d)
    (interaction.h:591), through the Scene::ray_intersect surface (mi_ray_intersect's checker). A second mesh without
    texture coordinates takes the other branch: dp_du from coordinate_system(n), vector.h:116-136."""
    rng = np.random.default_rng(21)
    v = np.array([[0, 0, 0], [2, 0, 0.3], [0.2, 1.5, 0.1], [2.1, 1.7, -0.2]], np.float32)
    f = np.array([[0, 1, 2], [1, 3, 2]], np.uint32)
    nrm = np.array([[0.1, -0.1, 1], [0.3, 0.0, 1], [-0.2, 0.2, 1], [0.0, 0.3, 1]], np.float64)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    uv = np.array([[0, 0], [1, 0.1], [0.1, 0.9], [1.2, 1.1]], np.float32)
    mesh = native.Mesh("m", v, f, normals=nrm.astype(np.float32), texcoords=uv, bsdf=native.BSDF("diffuse", reflectance=(0.5, 0.5, 0.5)))
    scene = native.Scene([mesh]).build(-1)
    n = 300
    o = np.concatenate([rng.uniform(0.2, 1.8, (n, 1)), rng.uniform(0.2, 1.3, (n, 1)), rng.uniform(1.0, 3.0, (n, 1))], 1).astype(np.float32)
    d = np.concatenate([rng.uniform(-0.3, 0.3, (n, 2)), -np.ones((n, 1))], 1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    si = oracle.ray_intersect(scene.desc(), o, d, 1e-4, np.inf)
    P, N, UV = v.astype(np.float64), nrm.astype(np.float32).astype(np.float64), uv.astype(np.float64)
    checked = 0
    for i in range(n):
        if not np.isfinite(si["t"][i]):
            continue
        oo, dd = o[i].astype(np.float64), d[i].astype(np.float64)
        best = None
        for face, (a, b, c) in enumerate(f):                     # mesh.h:194-226
            e1, e2 = P[b] - P[a], P[c] - P[a]
            pv = np.cross(dd, e2)
            det = e1 @ pv
            tv = oo - P[a]
            uu = (tv @ pv) / det
            qv = np.cross(tv, e1)
            vv = (dd @ qv) / det
            t = (e2 @ qv) / det
            if uu >= 0 and vv >= 0 and uu + vv <= 1 and t >= 1e-4 and (best is None or t < best[0]):
                best = (t, uu, vv, face)
        assert best is not None
        t, b1, b2, face = best
        if min(b1, b2, 1 - b1 - b2) < 1e-4:
            continue                                              # on the shared edge: either face is a valid answer
        a, b, c = f[face]
        b0 = 1 - b1 - b2
        p = P[a] * b0 + P[b] * b1 + P[c] * b2
        dp0, dp1 = P[b] - P[a], P[c] - P[a]
        ng = np.cross(dp0, dp1); ng /= np.linalg.norm(ng)
        tex = UV[a] * b0 + UV[b] * b1 + UV[c] * b2
        ns = N[a] * b0 + N[b] * b1 + N[c] * b2; ns /= np.linalg.norm(ns)
        duv0, duv1 = UV[b] - UV[a], UV[c] - UV[a]                # dp_du from the texture coordinates, mesh.cpp:498-510
        det = duv0[0] * duv1[1] - duv0[1] * duv1[0]
        assert det != 0
        dp_du = (duv1[1] * dp0 - duv0[1] * dp1) / det
        s_ = dp_du - ns * (ns @ dp_du); s_ /= np.linalg.norm(s_)  # initialize_sh_frame, interaction.h:153-156
        t_ = np.cross(ns, s_)
        wi = np.array([-dd @ s_, -dd @ t_, -dd @ ns])
        assert face == si["prim_index"][i] and si["shape_index"][i] == 0
        assert _close(si["t"][i], t, 2e-6) and _close(si["p"][i], p, 0, 2e-6) and _close(si["n"][i], ng, 0, 1e-6)
        assert _close(si["uv"][i], tex, 0, 1e-6) and _close(si["sh_n"][i], ns, 0, 1e-6)
        assert _close(si["sh_s"][i], s_, 0, 2e-6) and _close(si["sh_t"][i], t_, 0, 2e-6) and _close(si["wi"][i], wi, 0, 3e-6)
        checked += 1
    assert checked > 200

    # no texture coordinates, no vertex normals: uv = barycentrics, frame from coordinate_system(face normal)
    scene = native.Scene([native.Mesh("m", v, f, bsdf=native.BSDF("diffuse", reflectance=(0.5, 0.5, 0.5)))]).build(-1)
    si = oracle.ray_intersect(scene.desc(), o, d, 1e-4, np.inf)
    checked = 0
    for i in range(n):
        if not np.isfinite(si["t"][i]):
            continue
        a, b, c = f[si["prim_index"][i]]
        ng = np.cross(P[b] - P[a], P[c] - P[a]); ng /= np.linalg.norm(ng)
        sgn = math.copysign(1.0, ng[2])                           # vector.h:116-136
        aa = -1 / (sgn + ng[2]); bb = ng[0] * ng[1] * aa
        ms = (lambda x, y: x if y >= 0 else -x)                  # enoki::mulsign: x times the sign of y
        dp_du = np.array([ms(ng[0] * ng[0] * aa, ng[2]) + 1, ms(bb, ng[2]), -ms(ng[0], ng[2])])
        s_ = dp_du - ng * (ng @ dp_du); s_ /= np.linalg.norm(s_)
        dd = d[i].astype(np.float64)
        assert _close(si["sh_n"][i], ng, 0, 1e-6) and _close(si["sh_s"][i], s_, 0, 2e-6) and _close(si["sh_t"][i], np.cross(ng, s_), 0, 2e-6)
        assert _close(si["wi"][i], [-dd @ s_, -dd @ np.cross(ng, s_), -dd @ ng], 0, 3e-6)
        bary = si["uv"][i].astype(np.float64)
        assert _close(si["p"][i], P[a] * (1 - bary.sum()) + P[b] * bary[0] + P[c] * bary[1], 0, 2e-6)
        checked += 1
    assert checked > 200


def test_environment_map_against_float64_restatement(native, oracle):
    """EnvironmentMapEmitter (envmap.cpp): eval (:134-147, lat-long coordinates + the bilinear lookup of eval_spectrum
    :269-320), pdf_direction (:192-208) with the density Hierarchical2D normalises in its constructor (distr_2d.h:372-440:
    patch averages of luminance * sin(theta), summed in double) and sample_direction (:157-190) — whose warp is checked here
    through what it must satisfy: the direction it returns carries exactly the density pdf_direction assigns to it, the value
    is eval / pdf, the point lies two bounding-sphere radii away (set_scene, :127-131). The warp's own mapping is pinned by
    the reference's spot values and a chi-square test (tests/test_oracle_kat.py)."""
    from mitsuba2_amd import scenes
    Wd, Hd = 48, 24
    img = scenes.sky_envmap(Wd, Hd).astype(np.float64)
    scale = 1.7
    org, tgt, up = np.array([0.0, 0, 0]), np.array([0.3, 0.1, 1.0]), np.array([0.0, 1, 0])
    fwd = (tgt - org) / np.linalg.norm(tgt - org)                                  # Transform::look_at, transform.h:241-258
    left = np.cross(up, fwd); left /= np.linalg.norm(left)
    R = np.stack([left, np.cross(fwd, left), fwd], 1)                              # columns: to_world's rotation
    env = native.EnvMap(img.astype(np.float32), scale=scale, to_world=dict(origin=tuple(org), target=tuple(tgt), up=tuple(up)))
    v = np.array([[-2, 0, -2], [2, 0, -2], [2, 0, 2], [-2, 0, 2], [0, 3, 0]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [0, 1, 4]], np.uint32)
    scene = native.Scene([native.Mesh("m", v, f, bsdf=native.BSDF("diffuse", reflectance=(0.5, 0.5, 0.5)))], envmap=env).build(-1)
    lo, hi = v.astype(np.float64).min(0), v.astype(np.float64).max(0)
    eps = 2.0 ** -24 * 1500
    radius = max(eps, np.linalg.norm(0.5 * (lo + hi) - hi) * (1 + eps))             # bbox.h:329-332, envmap.cpp:127-131

    lum = img[..., 0] * 0.212671 + img[..., 1] * 0.715160 + img[..., 2] * 0.072169    # mitsuba::luminance, spectrum.h
    lum = lum.astype(np.float32).astype(np.float64)
    sin_t = np.sin(np.arange(Hd) / (Hd - 1) * PI)
    dens = lum * sin_t[:, None]
    avg = 0.25 * (dens[:-1, :-1] + dens[:-1, 1:] + dens[1:, :-1] + dens[1:, 1:])
    dens = dens * ((Wd - 1) * (Hd - 1) / avg.sum())

    def bilinear(table, uv):
        x, y = uv[0] * (Wd - 1), uv[1] * (Hd - 1)
        px, py = min(int(x), Wd - 2), min(int(y), Hd - 2)
        w1x, w1y = x - px, y - py
        return ((1 - w1y) * ((1 - w1x) * table[py, px] + w1x * table[py, px + 1]) +
                w1y * ((1 - w1x) * table[py + 1, px] + w1x * table[py + 1, px + 1]))

    def to_uv(d_world):
        dl = R.T @ d_world
        uv = np.array([math.atan2(dl[0], -dl[2]) / (2 * PI), math.acos(max(-1.0, min(1.0, dl[1]))) / PI])
        return uv - np.floor(uv), dl

    rng = np.random.default_rng(31)
    n = 400
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    x = np.concatenate([d, rng.uniform(-1, 1, (n, 3)), rng.random((n, 2))], 1).astype(np.float32)
    out = oracle.eval(9, x, scene.desc())
    checked = 0
    for xi, o in zip(x.astype(np.float64), out.astype(np.float64)):
        uv, dl = to_uv(xi[0:3])
        if min(uv[0], 1 - uv[0]) < 1e-3 or abs(dl[1]) > 0.999:
            continue                                              # the atan2 seam / the poles (inv_sin_theta ill-conditioned)
        inv_sin = 1 / math.sqrt(max(dl[0] ** 2 + dl[2] ** 2, (2.0 ** -24) ** 2))
        assert _close(o[0:3], bilinear(img, uv) * scale, 3e-4, 1e-5), (xi[:3], uv, o[:3])            # eval
        assert _close(o[3], bilinear(dens, uv) * inv_sin / (2 * PI * PI), 3e-4, 1e-7)                # pdf_direction
        sd, dist, pdf, spec = o[4:7], o[7], o[8], o[9:12]                                             # sample_direction
        assert abs(np.linalg.norm(sd) - 1) < 1e-5 and _close(dist, 2 * radius, 1e-6)
        suv, sdl = to_uv(sd / np.linalg.norm(sd))
        if min(suv[0], 1 - suv[0]) < 1e-3 or abs(sdl[1]) > 0.999 or pdf <= 0:
            continue
        s_inv_sin = 1 / math.sqrt(max(sdl[0] ** 2 + sdl[2] ** 2, (2.0 ** -24) ** 2))
        assert _close(pdf, bilinear(dens, suv) * s_inv_sin / (2 * PI * PI), 2e-3, 1e-7), (xi[6:8], suv, pdf)
        assert _close(spec * pdf, bilinear(img, suv) * scale, 2e-3, 1e-5)
        checked += 1
    assert checked > 300


def fresnel_diffuse_reflectance(eta):
    """render/fresnel.h:327-362"""
    if eta < 1:
        return -1.4399 * eta * eta + 0.7099 * eta + 0.6681 + 0.0636 / eta
    i = 1 / eta
    return 0.919317 - 3.4793 * i + 6.75335 * i ** 2 - 7.80989 * i ** 3 + 4.98554 * i ** 4 - 1.36881 * i ** 5


@pytest.mark.parametrize("nonlinear", [False, True])
def test_plastic_and_conductor_against_float64_restatement(native, oracle, nonlinear):
    """plastic.cpp:161-290 (lobe choice by the Fresnel-weighted sampling weights of parameters_changed, :161-176; the diffuse
    lobe's 1 / (1 - fdr_int * rho) with and without `nonlinear`) and the smooth conductor (conductor.cpp:216-262)."""
    rho = np.array([0.55, 0.3, 0.12])
    int_ior, ext_ior = 1.49, 1.000277
    eta = int_ior / ext_ior
    scene = _one_bsdf_scene(native, native.BSDF("plastic", diffuse_reflectance=tuple(rho), int_ior=int_ior, ext_ior=ext_ior, nonlinear=nonlinear))
    x = _bsdf_inputs(np.random.default_rng(13), 700, upper=False)
    out = oracle.eval(3, x, scene.desc())
    fdr_int = fresnel_diffuse_reflectance(1 / eta)
    ssw = 1.0 / (rho.mean() + 1.0)                               # specular sampling weight, s_mean = 1
    inv_eta_2 = 1 / (eta * eta)
    n_spec = n_diff = 0
    for xi, o in zip(x.astype(np.float64), out.astype(np.float64)):
        wi, s1, u2, wo = xi[1:4], xi[4], xi[5:7], xi[7:10]
        f_i = fresnel(wi[2], eta)[0]
        ps, pd_ = f_i * ssw, (1 - f_i) * (1 - ssw)
        ps = ps / (ps + pd_)
        diff = rho / (1 - (rho * fdr_int if nonlinear else fdr_int))
        if wi[2] > 0 and wo[2] > 0:
            f_o = fresnel(wo[2], eta)[0]
            assert _close(o[9:12], diff * wo[2] / PI * inv_eta_2 * (1 - f_i) * (1 - f_o), 3e-5, 1e-8)       # eval
            assert _close(o[12], wo[2] / PI * (1 - ps), 3e-5, 1e-8)                                            # pdf
        else:
            assert not o[9:13].any()
        if wi[2] <= 0:
            assert not o[6:9].any()
            continue
        if abs(s1 - ps) < 1e-4:
            continue
        if s1 < ps:
            assert _close(o[0:3], [-wi[0], -wi[1], wi[2]], 0, 1e-7) and _close(o[3], ps, 3e-5) and _close(o[6:9], [f_i / ps] * 3, 3e-5)
            n_spec += 1
        elif abs(abs(2 * u2[0] - 1) - abs(2 * u2[1] - 1)) > 1e-3:
            s = cosine_hemisphere(u2)
            f_o = fresnel(s[2], eta)[0]
            assert _close(o[0:3], s, 0, 3e-6) and _close(o[3], (1 - ps) * s[2] / PI, 3e-5, 2e-6)
            assert _close(o[6:9], diff * inv_eta_2 * (1 - f_i) * (1 - f_o) / (1 - ps), 1e-4, 1e-7)
            n_diff += 1
    assert n_spec > 10 and n_diff > 150
    if nonlinear:
        return
    eta_c, k_c = np.array([0.2, 0.92, 1.1]), np.array([3.9, 2.45, 2.14])
    scene = _one_bsdf_scene(native, native.BSDF("conductor", eta=tuple(eta_c), k=tuple(k_c)))
    x = _bsdf_inputs(np.random.default_rng(14), 300, upper=False)
    out = oracle.eval(3, x, scene.desc())
    for xi, o in zip(x.astype(np.float64), out.astype(np.float64)):
        wi = xi[1:4]
        assert not o[9:13].any()                                 # a delta lobe: eval = pdf = 0
        if wi[2] > 0:
            assert _close(o[0:3], [-wi[0], -wi[1], wi[2]], 0, 1e-7) and o[3] == 1.0 and o[4] == 1.0
            assert _close(o[6:9], fresnel_conductor(wi[2], eta_c, k_c), 3e-5)
        else:
            assert not o[6:9].any()


def roughdielectric(kind, au, av, visible, eta, wi, s1, u2, wo_eval):
    """roughdielectric.cpp:203-310 (sample), :312-390 (eval), :392-447 (pdf); TransportMode::Radiance, both lobes enabled"""
    ms = lambda v, s: v if s >= 0 else -v                         # enoki::mulsign on vectors / scalars
    d = Microfacet(kind, au, av, visible)
    ci = wi[2]
    out_s = None
    if ci != 0:
        sd = Microfacet(kind, au, av, visible)
        if not visible:
            k = 1.2 - 0.2 * math.sqrt(abs(ci))
            sd.au, sd.av = sd.au * k, sd.av * k                  # scale_alpha, microfacet.h:173-176
        m, pdf = sd.sample(ms(wi, ci), u2)
        if pdf != 0:
            F, cos_t, eta_it, eta_ti = fresnel(float(np.dot(wi, m)), eta)
            if s1 <= F:
                wo = 2 * np.dot(wi, m) * m - wi
                pdf *= F; bs_eta = 1.0; w = 1.0
                dwh = 1 / (4 * np.dot(wo, m))
            else:
                wo = m * (np.dot(wi, m) * eta_ti + cos_t) - wi * eta_ti
                pdf *= 1 - F; bs_eta = eta_it; w = eta_ti * eta_ti
                dwh = (bs_eta ** 2 * np.dot(wo, m)) / (np.dot(wi, m) + bs_eta * np.dot(wo, m)) ** 2
            w *= d.g1(wo, m) if visible else d.G(wi, wo, m) * np.dot(wi, m) / (ci * m[2])
            out_s = (wo, pdf * abs(dwh), bs_eta, w, F)
    ev = pd = 0.0
    co = wo_eval[2]
    if ci != 0:
        refl = ci * co > 0
        e, inv_e = (eta, 1 / eta) if ci > 0 else (1 / eta, eta)
        m = wi + wo_eval * (1.0 if refl else e)
        m /= np.linalg.norm(m)
        m = ms(m, m[2])
        D = d.eval(m)
        F = fresnel(float(np.dot(wi, m)), eta)[0]
        G = d.G(wi, wo_eval, m)
        if refl:
            ev = F * D * G / (4 * abs(ci))
        else:
            ev = abs((inv_e ** 2 * (1 - F) * D * G * e * e * np.dot(wi, m) * np.dot(wo_eval, m)) /
                     (ci * (np.dot(wi, m) + e * np.dot(wo_eval, m)) ** 2))
        if np.dot(wi, m) * ci > 0 and np.dot(wo_eval, m) * co > 0:
            dwh = 1 / (4 * np.dot(wo_eval, m)) if refl else (e * e * np.dot(wo_eval, m)) / (np.dot(wi, m) + e * np.dot(wo_eval, m)) ** 2
            sd = Microfacet(kind, au, av, visible)
            if not visible:
                k = 1.2 - 0.2 * math.sqrt(abs(ci))
                sd.au, sd.av = sd.au * k, sd.av * k
            pd = sd.pdf(ms(wi, ci), m) * (F if refl else 1 - F) * abs(dwh)
    return out_s, ev, pd


@pytest.mark.parametrize("kw", [
    dict(distribution="ggx", alpha=0.2),
    dict(distribution="ggx", alpha_u=0.1, alpha_v=0.35, sample_visible=False),
    dict(distribution="beckmann", alpha=0.3, sample_visible=False),
])
def test_roughdielectric_against_float64_restatement(native, oracle, kw):
    """roughdielectric.cpp: reflection and transmission lobes from both sides, the half-vector Jacobians, Walter's roughness
    scaling for plain sampling, the solid-angle compression factor (Beckmann's visible-normal inversion excepted, as above)."""
    int_ior, ext_ior = 1.5046, 1.000277
    eta = int_ior / ext_ior
    scene = _one_bsdf_scene(native, native.BSDF("roughdielectric", int_ior=int_ior, ext_ior=ext_ior, **kw))
    x = _bsdf_inputs(np.random.default_rng(23), 900, upper=False)
    x[np.abs(x[:, 3]) < 0.05, 3] = 0.3                           # keep wi off the horizon (pdfs blow up there)
    x[:, 1:4] /= np.linalg.norm(x[:, 1:4], axis=1, keepdims=True)
    out = oracle.eval(3, x, scene.desc())
    au, av = kw.get("alpha_u", kw.get("alpha")), kw.get("alpha_v", kw.get("alpha"))
    vis = kw.get("sample_visible", True)
    n_r = n_t = n_ev = 0
    for xi, o in zip(x.astype(np.float64), out.astype(np.float64)):
        wi, s1, u2, wo = xi[1:4], xi[4], xi[5:7], xi[7:10]
        if min(u2[0], 1 - u2[0], abs(abs(2 * u2[0] - 1) - abs(2 * u2[1] - 1))) < 1e-3 or abs(abs(u2[1] - 0.5) - 0.25) < 1e-3:
            continue
        smp, ev, pd = roughdielectric(kw["distribution"], au, av, vis, eta, wi, s1, u2, wo)
        if abs(wo[2]) > 0.02:
            assert _close(o[9], ev, 5e-4, 1e-6) and _close(o[12], pd, 5e-4, 1e-6), (kw, wi, wo, o[9:13], ev, pd)
            n_ev += ev > 0
        if smp is None:
            continue
        s_wo, s_pdf, s_eta, s_w, F = smp
        if abs(s1 - F) < 1e-3 or s_pdf < 1e-3 or abs(s_wo[2]) < 1e-2 or s_w == 0:
            continue
        assert _close(o[0:3], s_wo, 0, 3e-4) and _close(o[3], s_pdf, 2e-3) and _close(o[4], s_eta, 1e-6) and _close(o[6:9], [s_w] * 3, 1e-3, 1e-6), (kw, wi, s1, u2, o[:9], smp)
        n_r += s_eta == 1.0; n_t += s_eta != 1.0
    assert n_r > 30 and n_t > 200 and n_ev > 300


def _reference_tables():
    """CIE 1931 (src/libcore/spectrum.cpp:110-186, 95 samples x 3, 360 - 830 nm) and D65 (src/spectra/d65.cpp:12-25), parsed from
    the reference's sources where they lie — this test runs only where /root/reference exists (the CPU tier's box)."""
    import os
    import re
    ref = "/root/reference"
    if not os.path.exists(ref + "/src/libcore/spectrum.cpp"):
        pytest.skip("the reference tree is not on this box")
    num = r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?"
    txt = open(ref + "/src/libcore/spectrum.cpp").read()
    body = txt[txt.index("cie1931_tbl[MTS_CIE_SAMPLES * 3] = {"):]
    body = body[body.index("{") + 1:body.index("};")]
    cie = np.array([float(x) for x in re.findall(num, re.sub(r"(?<=\d)f", "", body))]).reshape(3, 95)
    txt = open(ref + "/src/spectra/d65.cpp").read()
    body = txt[txt.index("const float data[95] = {"):]
    body = body[body.index("{") + 1:body.index("};")]
    d65 = np.array([float(x) for x in re.findall(num, re.sub(r"(?<=\d)f", "", body))])
    assert cie.shape == (3, 95) and d65.shape == (95,)
    return cie, d65


def test_spectral_leaves_against_float64_restatement(native, oracle_spectral):
    """scalar_spectral (config 5): sample_shifted (math.h:419-442) + sample_rgb_spectrum (spectrum.h:271-285), srgb_model_eval
    (srgb.h:9-23), the srgb_d65 emitter spectrum = the upsampled colour times the D65 table (srgb_d65.cpp:56-62, d65.cpp:50-66,
    `regular`'s linear interpolation distr_1d.h:378-392) and spectrum_to_xyz over the CIE 1931 table (spectrum.h:147-217) —
    with both tables read from the reference's sources, not from the product's generated header."""
    cie, d65 = _reference_tables()

    def lerp_table(tab, lam):
        t = (lam - 360.0) * (94 / 470.0)
        i = int(min(max(int(t), 0), 93))
        return (1 - (t - i)) * tab[i] + (t - i) * tab[i + 1] if 360.0 <= lam <= 830.0 else 0.0

    rng = np.random.default_rng(41)
    n = 300
    x = np.zeros((n, 5), np.float32)
    x[:, 0] = rng.random(n)
    x[:, 1] = rng.normal(0, 2e-5, n); x[:, 2] = rng.normal(0, 2e-2, n); x[:, 3] = rng.normal(0, 3.0, n)   # sigmoid-polynomial coefficients
    x[:, 4] = rng.uniform(0.2, 30.0, n) / 10568.0                                                          # D65Spectrum::m_scale (d65.cpp:54-55: scale / 10568; the host folds it into the record)
    out = oracle_spectral.eval(11, x).astype(np.float64)
    for xi, o in zip(x.astype(np.float64), out):
        u, c0, c1, c2, scale = xi
        for k in range(4):
            s = u + k / 4.0
            if s > 1:
                s -= 1
            if min(s, 1 - s) < 1e-3:
                continue                                          # atanh's argument near its poles
            lam = 538.0 - math.atanh(0.8569106254698279 - 1.8275019724092267 * s) * 138.88888888888889
            wgt = 253.82 * math.cosh(0.0072 * (lam - 538.0)) ** 2
            assert _close(o[k], lam, 3e-6) and _close(o[4 + k], wgt, 3e-5), (u, k, o[k], lam)
            lam32 = o[k]                                          # evaluate the spectra at the wavelength the float32 code drew
            v = (c0 * lam32 + c1) * lam32 + c2
            srgb = max(0.0, 0.5 * v / math.sqrt(v * v + 1) + 0.5)
            assert _close(o[8 + k], srgb, 2e-4, 2e-6), (c0, c1, c2, lam32, o[8 + k], srgb)
            # (the sigmoid cancels badly for very negative v: the D65 factor is checked on the float32 colour the line above accepted)
            assert _close(o[12 + k], o[8 + k] * lerp_table(d65, lam32) * scale, 2e-5, 1e-12)
        lam4, w4, sd4 = o[0:4], o[4:8], o[12:16]
        xyz = [np.mean([lerp_table(cie[c], lam4[k]) * w4[k] * sd4[k] for k in range(4)]) for c in range(3)]
        assert _close(o[16:19], xyz, 2e-5, 1e-12)


def hier2d_build(data):
    """Hierarchical2D<Float, 0> constructor, distr_2d.h:372-462 -> levels[0] = normalised data, levels[1..] = MIP hierarchy"""
    h, w = data.shape
    ny, nx = h - 1, w - 1
    avg = 0.25 * (data[:-1, :-1] + data[:-1, 1:] + data[1:, :-1] + data[1:, 1:])
    scale = nx * ny / avg.sum()
    levels = [data * scale]

    def pad(a):
        return np.pad(a, ((0, a.shape[0] & 1), (0, a.shape[1] & 1)))
    cur = pad(avg * scale)
    levels.append(cur)
    max_level = int(math.ceil(math.log2(max(nx, ny)))) if max(nx, ny) > 1 else 0
    for _ in range(2, max_level + 2):
        nxt = cur[0::2, 0::2] + cur[0::2, 1::2] + cur[1::2, 0::2] + cur[1::2, 1::2]
        cur = pad(nxt) if max(nxt.shape) > 1 else nxt
        levels.append(cur)
    return levels, (nx, ny)


def hier2d_sample(levels, npatch, u):
    """Hierarchical2D::sample, distr_2d.h:473-556 + warp::square_to_bilinear / interval_to_linear, warp.h:359-407"""
    sx, sy = min(max(u[0], 0.0), 1.0), min(max(u[1], 0.0), 1.0)
    ox = oy = 0
    for l in range(len(levels) - 2, 0, -1):
        lv = levels[l]
        ox, oy = ox * 2, oy * 2
        # the four entries the reference fetches are consecutive in ITS storage (2 x 2 blocks, Level::index); here by coordinates
        v00, v10, v01, v11 = lv[oy, ox], lv[oy, ox + 1], lv[oy + 1, ox], lv[oy + 1, ox + 1]
        sx, sy = min(max(sx, 0.0), 1.0), min(max(sy, 0.0), 1.0)
        r0, r1 = v00 + v10, v01 + v11
        sy *= r0 + r1
        m = sy > r0
        if m:
            oy += 1; sy -= r0
        sy /= r1 if m else r0
        c0, c1 = (v01, v11) if m else (v00, v10)
        sx *= c0 + c1
        m = sx > c0
        if m:
            sx -= c0; ox += 1
        sx /= c1 if m else c0
    d = levels[0]
    v00, v10, v01, v11 = d[oy, ox], d[oy, ox + 1], d[oy + 1, ox], d[oy + 1, ox + 1]

    def i2l(v0, v1, s):
        if abs(v0 - v1) > 1e-4 * (v0 + v1):
            return (v0 - math.sqrt(max(0.0, v0 * v0 + (v1 * v1 - v0 * v0) * s))) / (v0 - v1)
        return s
    r0, r1 = v00 + v10, v01 + v11
    sy = i2l(r0, r1, sy)
    c0, c1 = v00 + (v01 - v00) * sy, v10 + (v11 - v10) * sy
    sx = i2l(c0, c1, sx)
    return (ox + sx) / npatch[0], (oy + sy) / npatch[1], c0 + (c1 - c0) * sx


@pytest.mark.parametrize("shape", [(5, 7), (2, 2), (9, 4), (17, 33)])
def test_hierarchical2d_sample_against_float64_restatement(oracle, shape):
    """The warp the environment map is sampled with: the MIP hierarchy of patch averages (zero-padded to even sizes) and the
    top-down row / column selection of Hierarchical2D::sample, then the bilinear patch — odd and even sizes, the 1-patch case."""
    import ctypes as C
    rng = np.random.default_rng(shape[0] * 100 + shape[1])
    data = (rng.random(shape) * 10 + 0.05).astype(np.float32)
    levels, npatch = hier2d_build(data.astype(np.float64))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    checked = 0
    for u in rng.random((300, 2)):
        xy = np.array(u, np.float32); out = np.zeros(3, np.float32)
        assert oracle.L.orc_hier2d(fp(data), shape[1], shape[0], 0, fp(xy), fp(out)) == 0
        x, y, pdf = hier2d_sample(levels, npatch, xy.astype(np.float64))
        # a sample within float32 rounding of a row / column boundary may legitimately fall on the other side
        if abs(x * npatch[0] - round(x * npatch[0])) < 2e-3 or abs(y * npatch[1] - round(y * npatch[1])) < 2e-3:
            continue
        assert _close(out[0:2], [x, y], 0, 3e-5) and _close(out[2], pdf, 3e-4), (shape, u, out, (x, y, pdf))
        checked += 1
    assert checked > 200
