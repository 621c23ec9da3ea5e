"""Chi-square goodness-of-fit tests of the sampling routines against their densities — the statistical tier of the
reference's test pyramid (src/python/python/chi2.py, used by src/bsdfs/tests/test_diffuse.py:41-53,
test_rough_conductor.py:6-97, test_rough_dielectric.py, src/libcore/tests/test_warp.py). Restated in numpy over the
oracle's batched leaf evaluation (orc_eval: the same leaf code the device runs; the device's bit-equality with it is
the GPU tests' job): histogram the sampled directions over a (phi, cos theta) grid, integrate the density over every
cell by a midpoint sub-grid, pool cells with small expectation, Pearson statistic, p-value from scipy."""
import numpy as np
import pytest
from scipy import stats

SIGNIFICANCE = 0.01
N_TESTS = 19                                   # Sidak correction over the tests of this module


def _dir(phi, z):
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    return np.stack([s * np.cos(phi), s * np.sin(phi), z], -1)


def chi2_sphere(sample_dirs, pdf_of_dirs, n_samples, res=(64, 48), ires=16, z_range=(-1.0, 1.0)):
    """sample_dirs: (n, 3) sampled directions (rows of zeros = no sample). pdf_of_dirs(dirs (m, 3)) -> solid-angle density."""
    nphi, nz = res
    valid = np.linalg.norm(sample_dirs, axis=1) > 0.5
    d = sample_dirs[valid]
    phi = np.mod(np.arctan2(d[:, 1], d[:, 0]), 2 * np.pi)
    z = np.clip(d[:, 2], -1, 1)
    z0, z1 = z_range
    ix = np.minimum((phi / (2 * np.pi) * nphi).astype(int), nphi - 1)
    iz = np.minimum(((z - z0) / (z1 - z0) * nz).astype(int), nz - 1)
    inside = (z >= z0) & (z <= z1)
    obs = np.zeros((nz, nphi)); np.add.at(obs, (iz[inside], ix[inside]), 1.0)
    # expected counts: midpoint rule on an ires x ires sub-grid per cell (d omega = dphi dz)
    sp = (np.arange(nphi * ires) + 0.5) / (nphi * ires) * 2 * np.pi
    sz = z0 + (np.arange(nz * ires) + 0.5) / (nz * ires) * (z1 - z0)
    P, Z = np.meshgrid(sp, sz)
    pts = _dir(P.ravel(), Z.ravel()).astype(np.float32)
    dens = np.concatenate([pdf_of_dirs(pts[i:i + (1 << 20)]) for i in range(0, len(pts), 1 << 20)])
    dens = dens.reshape(nz * ires, nphi * ires).astype(np.float64)
    cell = dens.reshape(nz, ires, nphi, ires).sum((1, 3)) * (2 * np.pi / (nphi * ires)) * ((z1 - z0) / (nz * ires))
    exp = cell * n_samples
    # pool low-expectation cells (chi2.py: pooling threshold 5)
    order = np.argsort(exp.ravel())
    e, o = exp.ravel()[order], obs.ravel()[order]
    k = int(np.searchsorted(np.cumsum(e), 5.0)) + 1
    e = np.concatenate([[e[:k].sum()], e[k:]]); o = np.concatenate([[o[:k].sum()], o[k:]])
    keep = e > 0
    stat = float(((o[keep] - e[keep]) ** 2 / e[keep]).sum())
    dof = int(keep.sum()) - 1
    p = float(stats.chi2.sf(stat, dof))
    return p, stat, dof, float(exp.sum() / n_samples), float(inside.sum() / n_samples)


def _threshold():
    return 1.0 - (1.0 - SIGNIFICANCE) ** (1.0 / N_TESTS)


def _bsdf_chi2(native, oracle, plugin, wi, n=400000, seed=0, res=(64, 48), ires=16, **kw):
    bsdf = native.BSDF(plugin, **kw)
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    scene = native.Scene([native.Mesh("t", v, np.array([[0, 1, 2]], np.uint32), bsdf=bsdf)]).build(-1)
    wi = np.asarray(wi, np.float32); wi = wi / np.linalg.norm(wi)
    rng = np.random.default_rng(seed)
    inp = np.zeros((n, 10), np.float32)
    inp[:, 0] = np.zeros(n, np.uint32).view(np.float32); inp[:, 1:4] = wi; inp[:, 4:7] = rng.random((n, 3)); inp[:, 7:10] = (0, 0, 1)
    out = oracle.eval(3, inp, scene.desc())
    wo, weight = out[:, 0:3].copy(), out[:, 6:9]
    wo[~(weight != 0).any(1)] = 0                                  # chi2.py BSDFAdapter: zero-weight samples do not count

    def pdf(dirs):
        q = np.zeros((len(dirs), 10), np.float32)
        q[:, 0] = np.zeros(len(dirs), np.uint32).view(np.float32); q[:, 1:4] = wi; q[:, 7:10] = dirs
        return oracle.eval(3, q, scene.desc())[:, 12]
    return chi2_sphere(wo, pdf, n, res=res, ires=ires)


def test_chi2_detects_a_wrong_density(native, oracle):
    """the harness itself: cosine-hemisphere samples against the uniform-hemisphere density must fail"""
    rng = np.random.default_rng(0)
    u = rng.random((200000, 2)).astype(np.float32)
    d = oracle.eval(2, u)[:, 0:3]
    p, *_ = chi2_sphere(d, lambda x: np.where(x[:, 2] > 0, 1 / (2 * np.pi), 0.0), len(d))
    assert p < 1e-6
    p, stat, dof, mass, frac = chi2_sphere(d, lambda x: np.maximum(x[:, 2], 0) / np.pi, len(d))
    assert p > _threshold() and abs(mass - 1) < 1e-3                 # src/libcore/tests/test_warp.py: cosine hemisphere


def test_chi2_diffuse(native, oracle):
    """test_diffuse.py:41-53"""
    p, stat, dof, mass, frac = _bsdf_chi2(native, oracle, "diffuse", [0.3, 0.2, 0.93])
    assert p > _threshold() and abs(mass - 1) < 2e-3


@pytest.mark.parametrize("kw,wi", [
    (dict(alpha=0.05), [0.8, 0.3, 0.05]),                                            # test_rough_conductor.py test01
    (dict(alpha=0.5), [0.8, 0.3, 0.05]),                                             # test02 rough, grazing
    (dict(alpha=0.5, distribution="beckmann", sample_visible=False), [0.5, 0.0, 0.5]),   # test03
    (dict(alpha=0.5, distribution="beckmann", sample_visible=True), [0.5, 0.0, 0.5]),    # test04
    (dict(alpha=0.5, distribution="ggx", sample_visible=False), [0.5, 0.0, 0.5]),        # test05
    (dict(alpha=0.5, distribution="ggx", sample_visible=True), [0.5, 0.0, 0.5]),         # test06
    (dict(alpha_u=0.4, alpha_v=0.1, distribution="ggx"), [0.5, 0.5, 0.5]),               # anisotropic
])
def test_chi2_roughconductor(native, oracle, kw, wi):
    res = (256, 192) if min(kw.get("alpha", 1), kw.get("alpha_v", 1)) < 0.1 else (64, 48)
    p, stat, dof, mass, frac = _bsdf_chi2_res(native, oracle, "roughconductor", wi, res, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), **kw)
    assert p > _threshold(), (p, stat, dof, mass, frac)
    assert abs(mass - frac) < 0.02                                  # density mass == share of valid samples


@pytest.mark.parametrize("kw,wi", [
    (dict(alpha=0.05), [0.8, 0.3, 0.05]),                                            # test_rough_dielectric.py test01
    (dict(alpha=0.5), [0.8, 0.3, 0.05]),                                             # test02
    (dict(alpha=0.5, distribution="beckmann", sample_visible=False), [0.5, 0.0, 0.5]),   # test03
    (dict(alpha=0.5, distribution="beckmann", sample_visible=True), [0.5, 0.0, 0.5]),    # test04
    (dict(alpha=0.5, distribution="ggx", sample_visible=False), [0.5, 0.0, 0.5]),        # test05
    (dict(alpha=0.5, distribution="ggx", sample_visible=True), [0.5, 0.0, 0.5]),         # test06
    (dict(alpha=0.3, distribution="ggx"), [0.4, -0.2, -0.6]),                            # from inside the medium
])
def test_chi2_roughdielectric(native, oracle, kw, wi):
    res = (256, 192) if kw.get("alpha", 1) < 0.1 else (64, 48)
    p, stat, dof, mass, frac = _bsdf_chi2_res(native, oracle, "roughdielectric", wi, res, int_ior=1.5046, ext_ior=1.000277, **kw)
    assert p > _threshold(), (p, stat, dof, mass, frac)
    assert abs(mass - frac) < 0.02


def _bsdf_chi2_res(native, oracle, plugin, wi, res, n=600000, **kw):
    fine = res[0] > 64
    return _bsdf_chi2(native, oracle, plugin, wi, n=n, res=res, ires=16, **kw)     # quadrature error << sampling noise


# ---- emitter sampling (scene.cpp:164-231, area.cpp:121-187, shape.cpp:292-323, rectangle.cpp:111-130, envmap.cpp) ----

_TO_Y = np.array([[1, 0, 0], [0, 0, 1], [0, 1, 0]], np.float64)         # harness frame (polar axis z) <-> scene (+y up)


@pytest.mark.parametrize("analytic", [False, True])
def test_chi2_area_light_sampling(native, oracle, analytic):
    """Scene::sample_emitter_direction on a quad area light — a two-triangle mesh (area CDF + triangle warp) or the
    analytic rectangle (uniform in object space) — against the closed-form solid-angle density (1 / area) r^2 / |cos|
    of a uniformly sampled quad, from a reference point below it."""
    from mitsuba2_amd import scenes
    corners = np.array(scenes._CBOX["light"], np.float64)              # y = 548, x in [213, 343], z in [227, 332]
    light = native.AreaLight((17.0, 12.0, 4.0))
    if analytic:
        shape = scenes._rect("light", corners, (278, 0, 280), emitter=light)
    else:
        v, f = scenes._quad(corners, inward_point=(278, 0, 280))
        shape = native.Mesh("light", v, f, emitter=light)
    scene = native.Scene([shape]).build(-1)
    ref = np.array([250.0, 300.0, 260.0])
    n = 400000
    rng = np.random.default_rng(4)
    inp = np.zeros((n, 5), np.float32); inp[:, 0:3] = ref; inp[:, 3:5] = rng.random((n, 2))
    out = oracle.eval(6, inp, scene.desc())
    d_s, pdf_s, val = out[:, 0:3].astype(np.float64), out[:, 4], out[:, 11:14]
    assert (pdf_s > 0).all() and np.allclose(val * pdf_s[:, None], (17.0, 12.0, 4.0), rtol=1e-4)
    x0, x1, zz0, zz1, yl = 213.0, 343.0, 227.0, 332.0, 548.0
    area = (x1 - x0) * (zz1 - zz0)

    def pdf(dirs_h):                                                   # harness frame -> scene frame
        d = dirs_h.astype(np.float64) @ _TO_Y.T
        t = (yl - ref[1]) / np.where(d[:, 1] > 1e-9, d[:, 1], np.inf)
        p = ref + d * t[:, None]
        inside = (d[:, 1] > 1e-9) & (p[:, 0] >= x0) & (p[:, 0] <= x1) & (p[:, 2] >= zz0) & (p[:, 2] <= zz1)
        return np.where(inside, (1 / area) * t * t / np.maximum(d[:, 1], 1e-9), 0.0)
    sel = rng.integers(0, n, 2000)
    assert np.allclose(pdf(d_s[sel] @ _TO_Y), pdf_s[sel], rtol=2e-4)   # the sampled pdf is that density
    p, stat, dof, mass, frac = chi2_sphere(d_s @ _TO_Y, pdf, n, res=(96, 96), ires=12, z_range=(0.85, 1.0))
    assert p > _threshold() and abs(mass - 1) < 3e-3 and frac == 1.0, (p, stat, dof, mass, frac)


def test_chi2_envmap_sampling(native, oracle):
    """EnvironmentMapEmitter::sample_direction through the hierarchical warp (envmap.cpp:157-190, distr_2d.h) against
    its own pdf_direction (:192-208) over the whole sphere"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.open_box(16, 16, 1, device=-1, with_area_light=False, env_size=(64, 32))
    n = 600000
    rng = np.random.default_rng(9)
    inp = np.zeros((n, 8), np.float32); inp[:, 0:3] = (0, 0, 1); inp[:, 3:6] = (278, 200, 280); inp[:, 6:8] = rng.random((n, 2))
    out = oracle.eval(9, inp, scene.desc())
    d_s, pdf_s = out[:, 4:7].astype(np.float64), out[:, 8]

    def pdf(dirs):
        q = np.zeros((len(dirs), 8), np.float32); q[:, 0:3] = dirs; q[:, 3:6] = (278, 200, 280)
        return oracle.eval(9, q, scene.desc())[:, 3]
    sel = rng.integers(0, n, 2000)
    assert np.allclose(pdf(d_s[sel].astype(np.float32)), pdf_s[sel], rtol=2e-3)
    p, stat, dof, mass, frac = chi2_sphere(d_s, pdf, n, res=(128, 96), ires=8)
    assert p > _threshold() and abs(mass - 1) < 5e-3, (p, stat, dof, mass, frac)
