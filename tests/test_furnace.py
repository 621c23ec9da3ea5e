"""Closed-form end-to-end checks of the estimator (no shared code between expectation and renderer).

Inside a closed diffuse enclosure whose every wall emits Le and reflects rho, the radiance is the same in every
direction: L = Le + rho L, i.e. L = Le / (1 - rho). The path integrator (emitter sampling + BSDF sampling under MIS,
Russian roulette from depth 5, unbounded depth) must converge to that value in every pixel; a lossless smooth
dielectric placed inside the enclosure must stay invisible (radiance is conserved along rays up to the eta^2 factor
that the Radiance-mode weights of dielectric.cpp:302-307 undo on the way out); and the direct integrator sees
exactly one bounce: Le (1 + rho)."""
import numpy as np
import pytest


def _cube(native, le, rho, extra=()):
    v = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float32) * 2
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5], [4, 7, 6], [0, 4, 5], [0, 5, 1], [3, 2, 6], [3, 6, 7],
                  [0, 3, 7], [0, 7, 4], [1, 5, 6], [1, 6, 2]], np.uint32)      # normals point inwards
    walls = native.Mesh("walls", v, f, bsdf=native.BSDF("diffuse", reflectance=(rho, rho, rho)),
                        emitter=native.AreaLight(radiance=(le, le, le)))
    return native.Scene([walls] + list(extra)).build(-1)


def _sensor(native, spp, w=24, h=16):
    film = native.Film(rfilter="box", width=w, height=h)
    return native.Sensor(film, native.Sampler(sample_count=spp, seed=1), fov=70.0,
                         to_world=dict(origin=(0.3, 0.2, -1.2), target=(0, 0, 0.5), up=(0, 1, 0)))


def _radiance(film):
    # box filter: W = samples per pixel; Y of a grey radiance equals the radiance (srgb_to_xyz row 2 sums to 1)
    return film[..., 1] / film[..., 4]


def test_normals_point_inwards(native, oracle):
    scene = _cube(native, 1.0, 0.0)
    sensor = _sensor(native, 4)
    f, _, st = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=4)
    assert np.allclose(_radiance(f), 1.0, atol=1e-6) and (f[..., 3] == f[..., 4]).all()      # black walls: the emission alone


@pytest.mark.parametrize("rho", [0.3, 0.7])
def test_diffuse_furnace_converges_to_closed_form(native, oracle, rho):
    le = 0.5
    scene = _cube(native, le, rho)
    sensor = _sensor(native, 192)
    f, _, st = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=16)
    L = _radiance(f)
    want = le / (1 - rho)
    assert abs(L.mean() / want - 1) < 4e-3                           # image mean: 73k paths
    assert np.abs(L / want - 1).max() < 0.15                         # every pixel on its own (192 paths each; roulette noise)
    d, _, _ = oracle.render(scene.desc(), native.DirectIntegrator().render_job(sensor), threads=16)
    assert abs(_radiance(d).mean() / (le * (1 + rho)) - 1) < 4e-3    # one bounce


def test_lossless_dielectric_is_invisible_in_the_furnace(native, oracle):
    from mitsuba2_amd import scenes
    le, rho = 0.5, 0.5
    p, f, n = scenes.icosphere((0.1, 0.0, 0.6), 0.5, 2)
    ball = native.Mesh("ball", p, f, normals=n, bsdf=native.BSDF("dielectric", int_ior=1.5, ext_ior=1.0))
    scene = _cube(native, le, rho, extra=[ball])
    sensor = _sensor(native, 192)
    img, _, st = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=16)
    L = _radiance(img)
    want = le / (1 - rho)
    assert abs(L.mean() / want - 1) < 6e-3
    assert np.abs(L / want - 1).max() < 0.12
    assert st.segments / st.samples > 2.5


def test_rough_coatings_never_gain_energy_in_the_furnace(native, oracle):
    """rough conductor (white specular reflectance, k = 0 would be lossless; the copper-like one absorbs) and rough
    plastic objects can only lose energy: every pixel stays at or below the empty furnace's radiance (+ noise)"""
    from mitsuba2_amd import scenes
    le, rho = 0.5, 0.5
    p, f, n = scenes.icosphere((0.1, 0.0, 0.6), 0.5, 2)
    for bsdf in (native.BSDF("roughplastic", alpha=0.2, diffuse_reflectance=(0.9, 0.9, 0.9)),
                 native.BSDF("roughconductor", alpha=0.3, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14))):
        scene = _cube(native, le, rho, extra=[native.Mesh("ball", p, f, normals=n, bsdf=bsdf)])
        img, _, _ = oracle.render(scene.desc(), native.PathIntegrator().render_job(_sensor(native, 128)), threads=16)
        L = _radiance(img)
        assert L.mean() < 1.0 + 5e-3 and L.max() < 1.12 and L.min() > 0.3
