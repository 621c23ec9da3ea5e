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


def test_convex_object_under_a_uniform_environment(native, oracle):
    """A convex diffuse object lit by a uniform environment sees Le over its whole hemisphere: L = rho Le on the
    object (no inter-reflection), Le on the background — environment-map sampling (hierarchical warp), its MIS against
    BSDF sampling and the analytic sphere in one closed form."""
    le, rho = 0.8, 0.6
    env = native.EnvMap(np.full((16, 32, 3), le, np.float32))
    ball = native.Mesh.sphere(center=(0, 0, 0), radius=1.0, bsdf=native.BSDF("diffuse", reflectance=(rho, rho, rho)))
    scene = native.Scene([ball], envmap=env).build(-1)
    film = native.Film(rfilter="box", width=32, height=32)
    sensor = native.Sensor(film, native.Sampler(sample_count=128, seed=2), fov=40.0,
                           to_world=dict(origin=(0, 0, -4), target=(0, 0, 0), up=(0, 1, 0)))
    img, _, st = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=16)
    L = _radiance(img); alpha = img[..., 3] / img[..., 4]
    on, off = alpha == 1, alpha == 0
    assert on.sum() > 150 and off.sum() > 400
    assert np.allclose(L[off], le, rtol=1e-5)                        # misses see the map itself
    assert abs(L[on].mean() / (rho * le) - 1) < 3e-3 and np.abs(L[on] / (rho * le) - 1).max() < 0.1


def test_spectral_furnace_scales_like_the_closed_form(spectral, oracle_spectral):
    """scalar_spectral: uniform reflectance rho under a D65 emitter: L(lambda) = Le(lambda) / (1 - rho) wavelength by
    wavelength, so the XYZ film of the rho = 0.5 enclosure is twice that of the black one"""
    sensor = _sensor(spectral, 128)
    films = []
    for rho in (0.0, 0.5):
        v = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float32) * 2
        f = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5], [4, 7, 6], [0, 4, 5], [0, 5, 1], [3, 2, 6], [3, 6, 7],
                      [0, 3, 7], [0, 7, 4], [1, 5, 6], [1, 6, 2]], np.uint32)
        walls = spectral.Mesh("walls", v, f, bsdf=spectral.BSDF("diffuse", reflectance=float(rho)),
                              emitter=spectral.AreaLight(radiance=(1.0, 1.0, 1.0)))
        scene = spectral.Scene([walls]).build(-1)
        img, _, _ = oracle_spectral.render(scene.desc(), spectral.PathIntegrator().render_job(sensor), threads=16)
        films.append(img)
    for k in range(3):
        ratio = films[1][..., k].sum() / films[0][..., k].sum()
        assert abs(ratio / 2.0 - 1) < 6e-3, (k, ratio)


@pytest.mark.parametrize("light", ["rectangle", "mesh"])
@pytest.mark.parametrize("integ", ["direct", "path"])
def test_irradiance_under_a_square_light_matches_the_form_factor(native, oracle, light, integ):
    """A diffuse floor point straight below the centre of a parallel square emitter (half side a, height h) receives
    E = Le pi F with the classic differential-element-to-rectangle form factor F = 4 F_corner(a / h, a / h),
    F_corner(X, Y) = [X / sqrt(1 + X^2) atan(Y / sqrt(1 + X^2)) + Y / sqrt(1 + Y^2) atan(X / sqrt(1 + Y^2))] / (2 pi);
    the camera sees rho Le F there. Analytic rectangle light and two-triangle mesh light, one-bounce integrators."""
    a, h, le, rho = 0.5, 1.0, 3.0, 0.8
    X = a / h
    f_corner = (X / np.sqrt(1 + X * X) * np.arctan(X / np.sqrt(1 + X * X))) * 2 / (2 * np.pi)
    want = rho * le * 4 * f_corner
    floor = native.Mesh("floor", [[-20, 0, -20], [20, 0, -20], [20, 0, 20], [-20, 0, 20]], [[0, 2, 1], [0, 3, 2]],
                        bsdf=native.BSDF("diffuse", reflectance=(rho, rho, rho)))
    black = native.BSDF("diffuse", reflectance=(0, 0, 0))
    if light == "mesh":
        lamp = native.Mesh("lamp", [[-a, h, -a], [a, h, -a], [a, h, a], [-a, h, a]], [[0, 1, 2], [0, 2, 3]], bsdf=black,
                           emitter=native.AreaLight(radiance=(le, le, le)))
    else:
        m = np.array([[a, 0, 0, 0], [0, 0, -a, h], [0, a, 0, 0], [0, 0, 0, 1]], np.float32)     # [-1,1]^2 -> the same square, normal -y
        lamp = native.Mesh.rectangle(to_world=m, bsdf=black, emitter=native.AreaLight(radiance=(le, le, le)))
    scene = native.Scene([floor, lamp]).build(-1)
    film = native.Film(rfilter="box", width=8, height=8)
    # looking at the origin from the side, under the lamp's plane, with a narrow field of view
    sensor = native.Sensor(film, native.Sampler(sample_count=2048, seed=5), fov=0.5,
                           to_world=dict(origin=(3.0, 0.6, 0.0), target=(0, 0, 0), up=(0, 1, 0)))
    integrator = native.DirectIntegrator() if integ == "direct" else native.PathIntegrator(max_depth=2)
    img, _, st = oracle.render(scene.desc(), integrator.render_job(sensor), threads=16)
    L = _radiance(img)
    assert (img[..., 3] == img[..., 4]).all()
    assert abs(L.mean() / want - 1) < 4e-3, (L.mean(), want)


def test_irradiance_from_a_sphere_light(native, oracle):
    """A surface element facing a spherical emitter (radius r, centre at distance d along its normal) receives
    E = pi Le (r / d)^2: the analytic sphere's solid-angle (cone) sampling and its pdf under MIS"""
    r, d, le, rho = 0.4, 1.5, 5.0, 0.7
    want = rho * le * (r / d) ** 2
    floor = native.Mesh("floor", [[-20, 0, -20], [20, 0, -20], [20, 0, 20], [-20, 0, 20]], [[0, 2, 1], [0, 3, 2]],
                        bsdf=native.BSDF("diffuse", reflectance=(rho, rho, rho)))
    lamp = native.Mesh.sphere(center=(0, d, 0), radius=r, bsdf=native.BSDF("diffuse", reflectance=(0, 0, 0)),
                              emitter=native.AreaLight(radiance=(le, le, le)))
    scene = native.Scene([floor, lamp]).build(-1)
    film = native.Film(rfilter="box", width=8, height=8)
    sensor = native.Sensor(film, native.Sampler(sample_count=2048, seed=6), fov=0.5,
                           to_world=dict(origin=(3.0, 0.6, 0.0), target=(0, 0, 0), up=(0, 1, 0)))
    for integrator in (native.DirectIntegrator(), native.DirectIntegrator(emitter_samples=0, bsdf_samples=4)):
        img, _, _ = oracle.render(scene.desc(), integrator.render_job(sensor), threads=16)
        tol = 4e-3 if integrator.render_job(sensor).cfg.emitter_samples else 2e-2       # BSDF sampling alone is noisier
        assert abs(_radiance(img).mean() / want - 1) < tol, (_radiance(img).mean(), want)


def test_floor_under_a_zenith_weighted_sky(native, oracle):
    """Lat-long environment map whose radiance depends on the polar angle only, L(theta) = C max(0, cos theta)^2 with
    theta measured from +Y (row 0 = +Y pole, envmap.cpp:140-143): a horizontal diffuse floor receives
    E = C int cos^3 theta d omega = pi C / 2 and shows rho C / 2 — the map's orientation, its importance sampling
    through the hierarchical warp and the MIS with BSDF sampling, against a closed form"""
    C, rho, H, W = 2.0, 0.6, 256, 512
    theta = np.pi * np.arange(H) / (H - 1)
    row = C * np.maximum(0.0, np.cos(theta)) ** 2
    sky = np.repeat(np.repeat(row[:, None, None], W, axis=1), 3, axis=2).astype(np.float32)
    floor = native.Mesh("floor", [[-50, 0, -50], [50, 0, -50], [50, 0, 50], [-50, 0, 50]], [[0, 2, 1], [0, 3, 2]],
                        bsdf=native.BSDF("diffuse", reflectance=(rho, rho, rho)))
    scene = native.Scene([floor], envmap=native.EnvMap(sky)).build(-1)
    film = native.Film(rfilter="box", width=8, height=8)
    sensor = native.Sensor(film, native.Sampler(sample_count=1024, seed=7), fov=1.0,
                           to_world=dict(origin=(2.0, 3.0, 1.0), target=(0, 0, 0), up=(0, 1, 0)))
    want = rho * C / 2
    for integrator in (native.PathIntegrator(), native.DirectIntegrator(shading_samples=2)):
        img, _, _ = oracle.render(scene.desc(), integrator.render_job(sensor), threads=16)
        assert abs(_radiance(img).mean() / want - 1) < 4e-3, (_radiance(img).mean(), want)
