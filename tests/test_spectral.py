"""scalar_spectral variant (BASELINE config 5): known answers of the reference's own spectral
tests, the host layer's sRGB upsampling fetch against the reference's ext/rgb2spec code, and the resident sample
loop against the scalar oracle — all on the CPU."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _spec(oracle, op, inp, n_out):
    a = np.ascontiguousarray(inp, np.float32); out = np.zeros(n_out, np.float32)
    oracle.L.orc_spectral(op, fp(a), fp(out))
    return out


# ---- src/librender/tests/test_spectra.py ---------------------------------------------------------------------------
def test_cie1931_spot_check(oracle_spectral):
    """test01_cie1931"""
    xyz = _spec(oracle_spectral, 0, [600.0], 3)
    assert np.allclose(xyz, [1.0622, 0.631, 0.0008], rtol=1e-5, atol=1e-6)
    assert np.array_equal(_spec(oracle_spectral, 0, [359.9], 3), [0, 0, 0])
    assert np.array_equal(_spec(oracle_spectral, 0, [830.1], 3), [0, 0, 0])


def test_d65_spot_check(oracle_spectral):
    """test02_d65: [350, 456, 700, 840] nm -> [0, 117.49, 71.6091, 0] / 10568"""
    got = [_spec(oracle_spectral, 1, [np.float32(1.0) * np.float32(1.0 / 10568.0), w], 1)[0] for w in (350, 456, 700, 840)]
    assert np.allclose(got, np.array([0, 117.49, 71.6091, 0]) / 10568.0, rtol=1e-5, atol=1e-9)


def test_sample_rgb_spectrum_spot_checks(oracle_spectral):
    """test05_sample_rgb_spectrum + math::sample_shifted"""
    for sample, wav, weight in ((0.1, 424.343, 465.291), (0.5, 545.903, 254.643), (0.8, 635.381, 400.432)):
        r = _spec(oracle_spectral, 2, [sample], 8)
        assert np.isclose(r[0], wav, rtol=1e-5) and np.isclose(r[4], weight, rtol=1e-5)
        pdf = 0.003939804229326285 / np.cosh(0.0072 * (r[0].astype(np.float64) - 538.0)) ** 2      # pdf_rgb_spectrum
        assert np.isclose(pdf, 1.0 / r[4], rtol=1e-4)
    r = _spec(oracle_spectral, 2, [0.6], 8)                      # shifted copies: 0.6, 0.85, 0.1 (wrapped), 0.35
    for k, sft in enumerate((0.6, 0.85, 0.1, 0.35)):
        assert np.isclose(r[k], _spec(oracle_spectral, 2, [sft], 8)[0], rtol=1e-6)
    assert np.isclose(r[2], 424.343, rtol=1e-5)
    assert (r[:4] >= 360).all() and (r[:4] <= 830).all()


def test_srgb_d65_is_d65_times_srgb(oracle_spectral):
    """test04_srgb_d65: the emitter texture = D65 x intensity x srgb(normalised colour)"""
    wl = np.array([412.0, 503.0, 611.0, 702.0], np.float32)
    coeff = np.array([1.2e-4, -0.13, 33.0], np.float32)         # any sigmoid-polynomial coefficients
    scale = np.float32(2.8) * np.float32(1.0 / 10568.0)
    u = lambda t: np.array([t], np.uint32).view(np.float32)[0]
    both = _spec(oracle_spectral, 3, [u(4), *coeff, scale, *wl], 4)
    srgb = _spec(oracle_spectral, 3, [u(2), *coeff, 0, *wl], 4)
    d65 = _spec(oracle_spectral, 3, [u(3), scale, 0, 0, 0, *wl], 4)
    assert np.array_equal(both, d65 * srgb) and (srgb > 0).all() and (srgb < 1).all()
    uni = _spec(oracle_spectral, 3, [u(1), 0.37, 0, 0, 0, 359.0, 360.0, 830.0, 831.0], 4)
    assert np.array_equal(uni, np.array([0, 0.37, 0.37, 0], np.float32))
    inf = _spec(oracle_spectral, 3, [u(2), 0, 0, np.inf, 0, *wl], 4)          # srgb_model_fetch(1,1,1)
    ninf = _spec(oracle_spectral, 3, [u(2), 0, 0, -np.inf, 0, *wl], 4)        # srgb_model_fetch(0,0,0)
    assert np.array_equal(inf, np.ones(4, np.float32)) and np.array_equal(ninf, np.zeros(4, np.float32))


# ---- host layer: srgb_model_fetch vs the reference's ext/rgb2spec --------------------------------------------------
def test_srgb_model_fetch_matches_reference_rgb2spec(spectral):
    """The host layer's table fetch (srgb.cpp:14-42 + rgb2spec_fetch) against the reference's own rgb2spec.c, compiled
    from where it lies into oracle/_ref/librgb2spec_ref.so; both read mitsuba2_amd/data/srgb.coeff = `rgb2spec_opt 64`."""
    from conftest import SRGB_COEFF
    ref_path = os.path.join(ROOT, "oracle", "_ref", "librgb2spec_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref/librgb2spec_ref.so missing")
    R = C.CDLL(ref_path)
    R.rgb2spec_load.restype = C.c_void_p; R.rgb2spec_load.argtypes = [C.c_char_p]
    R.rgb2spec_fetch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    model = R.rgb2spec_load(SRGB_COEFF.encode())
    assert model
    rng = np.random.default_rng(3)
    colors = np.concatenate([rng.random((200, 3)), [[0.725, 0.71, 0.68], [0.63, 0.065, 0.05], [0.14, 0.45, 0.091],
                                                      [0.5, 0.5, 0.5], [1, 0, 0], [0.2, 1.0, 1.0]]]).astype(np.float32)
    for c in colors:
        ref = np.zeros(3, np.float32)
        R.rgb2spec_fetch(model, fp(np.ascontiguousarray(c)), fp(ref))
        rec = spectral.BSDF("diffuse", reflectance=tuple(float(x) for x in c)).record()
        assert rec.tex[0].type == 2                                            # MI_TEX_SRGB
        got = np.array(rec.tex[0].v[:3], np.float32)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (c, got, ref)
    black = spectral.BSDF("diffuse", reflectance=(0.0, 0.0, 0.0)).record().tex[0]
    white = spectral.BSDF("diffuse", reflectance=(1.0, 1.0, 1.0)).record().tex[0]
    assert black.v[2] == -np.inf and white.v[2] == np.inf                      # srgb.cpp:30-33
    with pytest.raises(RuntimeError, match="range"):
        spectral.BSDF("diffuse", reflectance=(1.2, 0.5, 0.5))


def test_spectral_property_mapping(spectral):
    """src/libcore/xml.cpp:1073-1100: <rgb> -> srgb / srgb_d65 (emitters), <spectrum value> -> uniform / d65 (emitters)"""
    e = spectral.AreaLight((17.0, 12.0, 4.0))
    import mitsuba2_amd._capi as capi
    d = spectral.BSDF("dielectric").record()
    assert d.tex[0].type == 1 and d.tex[0].v[0] == 1.0 and d.tex[1].type == 1           # defaults: uniform 1
    rc = spectral.BSDF("roughconductor", eta=0.2, k=3.9).record()
    assert rc.tex[0].type == 1 and np.isclose(rc.tex[0].v[0], 0.2) and np.isclose(rc.tex[1].v[0], 3.9)
    rc2 = spectral.BSDF("roughconductor", eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14)).record()
    assert rc2.tex[0].type == 2 and rc2.tex[1].type == 2                                   # unbounded srgb
    assert spectral.host_lib().mih_spectrum_channels() == 4


# ---- the sample loop -----------------------------------------------------------------------------------------------
def _both(spectral, oracle_spectral, scene, sensor, **kw):
    job = spectral.PathIntegrator(**kw).render_job(sensor)
    job.cfg.plan = 2
    o32, o64, st = oracle_spectral.render(scene.desc(), job, threads=4)
    e64, e32, est = oracle_spectral.emu_render(scene.desc(), job)
    return job, o32, o64, st, e64, e32, est


@pytest.mark.parametrize("diffuse_only", [True, False])
def test_spectral_resident_plan_equals_scalar_oracle(spectral, oracle_spectral, diffuse_only):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(40, 32, 8, diffuse_only=diffuse_only, device=-1, ball_level=1)
    job, o32, o64, st, e64, e32, est = _both(spectral, oracle_spectral, scene, sensor)
    assert est[0] == st.samples == 40 * 32 * 8 and est[1] == st.segments
    assert np.array_equal(e32, o32) and np.isfinite(o32).all() and o32[..., 4].min() > 0
    assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))


@pytest.mark.parametrize("which", ["plugin_box", "rect_box", "frosted"])
def test_spectral_plugins_and_rectangles(spectral, oracle_spectral, which):
    """conductor / plastic / twosided / roughdielectric and the analytic rectangles in the scalar_spectral build: the
    resident sample loop == the scalar oracle bit for bit; the plastic's sampling weight uses the spectral
    Texture::mean() (srgb.h:26-35: the sigmoid-polynomial model averaged over 16 wavelengths)."""
    from mitsuba2_amd import scenes
    if which == "plugin_box":
        scene, sensor = scenes.plugin_box(40, 32, 6, device=-1)
    elif which == "rect_box":
        scene, sensor = scenes.rect_box(40, 32, 6, device=-1)
    else:
        scene, sensor = scenes.cornell_box(40, 32, 6, diffuse_only=False, device=-1, ball_level=1,
                                           glass=dict(alpha=0.2, distribution="ggx"))
    job, o32, o64, st, e64, e32, est = _both(spectral, oracle_spectral, scene, sensor)
    assert est[1] == st.segments and np.array_equal(e32, o32) and np.isfinite(o32).all()
    if which == "plugin_box":
        d = scene.desc().contents
        pl = [d.bsdfs[i] for i in range(d.bsdf_count) if d.bsdfs[i].type == 4]
        assert len(pl) == 2 and all(r.tex[0].type == 2 for r in pl)          # MI_TEX_SRGB records
        assert all(0.3 < r.params[3] < 0.8 for r in pl)                       # s_mean / (d_mean + s_mean)


def test_spectral_render_agrees_with_rgb_render(native, spectral, oracle_spectral, oracle):
    """Upsampling RGB to spectra and integrating back against the CIE observer must land near the RGB render
    (same geometry, same sampler): means of X, Y, Z within a few percent."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(48, 36, 32, diffuse_only=True, device=-1)
    job = spectral.PathIntegrator(max_depth=4).render_job(sensor)
    s32, _, _ = oracle_spectral.render(scene.desc(), job, threads=8, want_f64=False)
    native.set_variant("scalar_rgb")
    scene_r, sensor_r = scenes.cornell_box(48, 36, 32, diffuse_only=True, device=-1)
    job_r = native.PathIntegrator(max_depth=4).render_job(sensor_r)
    r32, _, _ = oracle.render(scene_r.desc(), job_r, threads=8, want_f64=False)
    ms = (s32[..., :3] / s32[..., 4:5]).reshape(-1, 3).mean(0); mr = (r32[..., :3] / r32[..., 4:5]).reshape(-1, 3).mean(0)
    assert np.allclose(ms, mr, rtol=0.08), (ms, mr)


# ---- the environment map in the spectral variant (round 4; src/emitters/envmap.cpp:86-116, :269-307) -----------------------------
def _srgb_model_eval64(c, lam):
    """include/mitsuba/render/srgb.h:9-23 in float64"""
    v = (c[..., 0:1] * lam + c[..., 1:2]) * lam + c[..., 2:3]
    return np.maximum(0.0, 0.5 * v / np.sqrt(v * v + 1.0) + 0.5)


def test_spectral_environment_map_against_float64_restatement(spectral, oracle_spectral):
    """The constructor's texel conversion (every texel -> coefficients of the sRGB model of rgb / max(1e-8, 2 hmax(rgb)) + that
    scale; the warp's density from the colour BEFORE the conversion) and eval_spectrum's spectral branch (four texels evaluated at
    the sample's wavelengths, spectra and scales interpolated separately, times D65(lambda) times m_scale), restated here in
    float64 numpy from the reference source; D65 itself through the checker's d65 leaf, which test_d65_spot_check pins on the
    reference's values. sample_direction through its contract: value * pdf == eval at the sampled direction."""
    import math
    from mitsuba2_amd import scenes
    Wd, Hd, scale = 40, 20, 1.3
    img = scenes.sky_envmap(Wd, Hd).astype(np.float32)
    env = spectral.EnvMap(img, scale=scale)
    v = np.array([[-2, 0, -2], [2, 0, -2], [2, 0, 2], [-2, 0, 2], [0, 3, 0]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [0, 1, 4]], np.uint32)
    scene = spectral.Scene([spectral.Mesh("m", v, f, bsdf=spectral.BSDF("diffuse", reflectance=(0.5, 0.5, 0.5)))], envmap=env).build(-1)
    rec = scene.desc().contents.envmap.contents
    data = np.ctypeslib.as_array(rec.rgba, (Hd, Wd, 4)).astype(np.float64)
    dens = np.ctypeslib.as_array(rec.density, (Hd, Wd)).astype(np.float64)
    # the constructor: scale = 2 hmax, coefficients of the normalised colour (host srgb_model_fetch: pinned on the reference's
    # rgb2spec by test_srgb_model_fetch_matches_reference_rgb2spec), density = luminance * sin(theta)
    rgb = img[..., :3].astype(np.float64)
    assert np.allclose(data[..., 3], 2 * rgb.max(-1), rtol=1e-6)
    lum = rgb @ np.array([0.212671, 0.715160, 0.072169])
    assert np.allclose(dens, lum * np.sin(np.arange(Hd) / (Hd - 1) * math.pi)[:, None], rtol=2e-6, atol=1e-7)
    for (y, x) in ((0, 0), (7, 13), (19, 39), (10, 20)):
        nrm = img[y, x, :3] * (np.float32(1) / max(np.float32(1e-8), np.float32(2) * img[y, x, :3].max()))
        c = spectral.BSDF("diffuse", reflectance=tuple(float(q) for q in nrm)).record().tex[0].v[:3]   # the host's srgb_model_fetch
        assert np.allclose(data[y, x, :3], np.array(c, np.float64), rtol=1e-5, atol=1e-9), (y, x)
    rng = np.random.default_rng(17)
    n = 300
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    wl = rng.uniform(365, 825, (n, 4))
    x = np.concatenate([d, rng.uniform(-1, 1, (n, 3)), rng.random((n, 2)), wl], 1).astype(np.float32)
    out = oracle_spectral.eval(9, x, scene.desc()).astype(np.float64)
    assert out.shape[1] == 14

    def d65(lam):
        return float(_spec(oracle_spectral, 1, [1.0 / 10568.0, lam], 1)[0])

    def eval64(uv, lams):
        xx, yy = uv[0] * (Wd - 1), uv[1] * (Hd - 1)
        px, py = min(int(xx), Wd - 2), min(int(yy), Hd - 2)
        w1x, w1y = xx - px, yy - py
        t = data[py:py + 2, px:px + 2]
        s = _srgb_model_eval64(t[..., :3][..., None, :], np.asarray(lams)[:, None])[..., 0]      # [2, 2, 4 wavelengths]
        sv = (1 - w1y) * ((1 - w1x) * s[0, 0] + w1x * s[0, 1]) + w1y * ((1 - w1x) * s[1, 0] + w1x * s[1, 1])
        fv = (1 - w1y) * ((1 - w1x) * t[0, 0, 3] + w1x * t[0, 1, 3]) + w1y * ((1 - w1x) * t[1, 0, 3] + w1x * t[1, 1, 3])
        return sv * np.array([d65(l) for l in lams]) * fv * scale

    def to_uv(dl):
        uv = np.array([math.atan2(dl[0], -dl[2]) / (2 * math.pi), math.acos(max(-1.0, min(1.0, dl[1]))) / math.pi])
        return uv - np.floor(uv)

    checked = 0
    for xi, o in zip(x.astype(np.float64), out):
        uv = to_uv(xi[0:3])
        if min(uv[0], 1 - uv[0]) < 1e-3 or abs(xi[1]) > 0.999:
            continue
        lams = xi[8:12]
        assert np.allclose(o[0:4], eval64(uv, lams), rtol=5e-4, atol=1e-7), (xi[:3], o[:4], eval64(uv, lams))
        sd, pdf, sv = o[5:8], o[9], o[10:14]
        suv = to_uv(sd / np.linalg.norm(sd))
        if min(suv[0], 1 - suv[0]) < 1e-3 or abs(sd[1]) > 0.999 or pdf <= 0:
            continue
        assert np.allclose(sv * pdf, eval64(suv, lams), rtol=3e-3, atol=1e-6)
        checked += 1
    assert checked > 200


@pytest.mark.parametrize("kw", [dict(), dict(with_area_light=False, envmap_after=0, env_scale=0.5)])
def test_spectral_environment_map_resident_plan_equals_scalar_oracle(spectral, oracle_spectral, kw):
    """the open box under the synthetic sky, scalar_spectral: the resident sample loop (CPU run of the device stages) against the
    scalar restatement — misses evaluate the map (path.cpp:126-129), emitter sampling goes through its warp (envmap.cpp:157-190)"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.open_box(36, 28, 6, device=-1, **kw)
    assert scene.desc().contents.envmap.contents.density
    job, o32, o64, st, e64, e32, est = _both(spectral, oracle_spectral, scene, sensor)
    assert est[0] == st.samples == 36 * 28 * 6 and est[1] == st.segments
    assert np.array_equal(e32, o32) and np.isfinite(o32).all() and o32[..., 1].mean() > 0
    assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))


def test_spectral_library_refuses_an_environment_map_without_its_density(spectral, oracle_spectral):
    """mi_envmap::density is required where the texels are coefficients (include/miwave.h): the checker mirrors the refusal"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.open_box(16, 16, 1, device=-1)
    rec = scene.desc().contents.envmap.contents
    keep = rec.density
    rec.density = None
    try:
        job = spectral.PathIntegrator().render_job(sensor)
        with pytest.raises(RuntimeError):
            oracle_spectral.render(scene.desc(), job, threads=1, want_f64=False)
    finally:
        rec.density = keep


# ---- GPU: libmiwave_spectral.so against the spectral oracle --------------------------------------------------------
@pytest.mark.gpu
def test_gpu_spectral_leaf_functions_bit_exact(spectral, oracle_spectral):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(32, 32, 1, diffuse_only=False, device=-1, ball_level=1)
    d = spectral.Device(0)
    assert d.L.mi_spectrum_channels() == 4
    d.upload(scene.desc())
    rng = np.random.default_rng(5)
    n = 4096
    # wavelength sampling, srgb / srgb_d65 textures, spectrum -> XYZ
    x = np.stack([rng.random(n), rng.normal(0, 2e-4, n), rng.normal(0, 0.1, n), rng.normal(0, 20, n),
                  rng.uniform(0.5, 40, n) / 10568.0], 1).astype(np.float32)
    g = d.eval(11, x); o = oracle_spectral.eval(11, x)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32)) and np.isfinite(g).all()
    # BSDFs (diffuse, dielectric, rough conductor) at sampled wavelengths
    wl = g[:, :4]
    nb = scene.desc().contents.bsdf_count
    wi = rng.normal(size=(n, 3)); wi[:, 2] = np.abs(wi[:, 2]); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    wo = rng.normal(size=(n, 3)); wo[:, 2] = np.abs(wo[:, 2]); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    idx = rng.integers(0, nb, n).astype(np.uint32).view(np.float32)
    xb = np.concatenate([idx[:, None], wi, rng.random((n, 3)), wo, wl], 1).astype(np.float32)
    xb[:, 0] = idx
    gb = d.eval(3, xb); ob = oracle_spectral.eval(3, xb, desc=scene.desc())
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    # emitter sampling (srgb_d65 radiance)
    xe = np.concatenate([rng.uniform(50, 500, (n, 3)), rng.random((n, 2)), wl], 1).astype(np.float32)
    ge = d.eval(6, xe); oe = oracle_spectral.eval(6, xe, desc=scene.desc())
    assert np.array_equal(ge.view(np.uint32), oe.view(np.uint32)) and (ge[:, 11:] >= 0).all()
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("diffuse_only", [True, False])
def test_gpu_spectral_render_parity(spectral, oracle_spectral, diffuse_only):
    """BASELINE config 5 class: scalar_spectral Cornell box (+ constant-IOR dielectric and GGX balls)."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 8, diffuse_only=diffuse_only, device=-1, ball_level=2)
    job = spectral.PathIntegrator().render_job(sensor)
    o32, o64, ost = oracle_spectral.render(scene.desc(), job, threads=8)
    d = spectral.Device(0)
    d.upload(scene.desc())
    for spl in (0, 3):
        g32, st = d.render(job, samples_per_launch=spl)
        c = d.counters()
        assert st == 0 and c.plan == 2 and c.film_mode == 1 and c.samples == ost.samples and c.segments == ost.segments
        assert np.array_equal(g32, o32)
    g64, st = d.render(job, f64=True, film_mode=2)
    assert st == 0 and np.array_equal(g64.astype(np.float32), o64.astype(np.float32))
    job.cfg.plan = 1
    film = np.zeros(job.cfg.crop_w * job.cfg.crop_h * 5, np.float32)
    assert d.L.mi_render(d.ctx, C.byref(job.cfg), film.ctypes.data_as(C.c_void_p)) == -1      # queue plan: RGB builds only
    if not diffuse_only:                                            # the (f)-4 plugins and analytic rectangles, spectral build
        for sc_, se_ in (scenes.plugin_box(64, 48, 6, device=-1), scenes.rect_box(64, 48, 6, device=-1)):
            j2 = spectral.PathIntegrator().render_job(se_)
            r32, _, rst = oracle_spectral.render(sc_.desc(), j2, threads=8, want_f64=False)
            d.upload(sc_.desc())
            q32, st = d.render(j2)
            assert st == 0 and d.counters().segments == rst.segments and np.array_equal(q32, r32)
    d.close()


@pytest.mark.gpu
def test_gpu_spectral_environment_map_bit_exact(spectral, oracle_spectral):
    """round 4: the environment map in libmiwave_spectral.so (envmap.cpp:86-116 conversion on the host, :269-307 on the device):
    leaf functions and whole renders against the spectral oracle, bit for bit"""
    from mitsuba2_amd import scenes
    rng = np.random.default_rng(23)
    d = spectral.Device(0)
    for kw in (dict(), dict(with_area_light=False, envmap_after=0, env_scale=0.5)):
        scene, sensor = scenes.open_box(64, 48, 8, device=-1, ball_level=2, **kw)
        d.upload(scene.desc())
        n = 4096
        dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        x = np.concatenate([dirs, rng.uniform(50, 500, (n, 3)), rng.random((n, 2)), rng.uniform(360, 830, (n, 4))], 1).astype(np.float32)
        g = d.eval(9, x); o = oracle_spectral.eval(9, x, desc=scene.desc())
        assert g.shape[1] == 14 and np.array_equal(g.view(np.uint32), o.view(np.uint32)) and (g[:, :4] >= 0).all() and g[:, :4].max() > 0
        job = spectral.PathIntegrator().render_job(sensor)
        o32, _, ost = oracle_spectral.render(scene.desc(), job, threads=8, want_f64=False)
        g32, st = d.render(job)
        c = d.counters()
        assert st == 0 and c.plan == 2 and c.film_mode == 1 and (c.samples, c.segments) == (ost.samples, ost.segments)
        assert np.array_equal(g32, o32) and g32[..., 1].mean() > 0
    # without its density the spectral library refuses the map (its texels are coefficients)
    rec = scene.desc().contents.envmap.contents
    keep = rec.density; rec.density = None
    try:
        with pytest.raises(RuntimeError, match="density"):
            d.upload(scene.desc())
    finally:
        rec.density = keep
    d.close()

