"""The moment integrator (src/integrators/moment.cpp) and what the reference uses it for: the per-pixel z-test of
src/python/python/test/test_renders.py:63-132.

The film of `moment` holds X Y Z A W, nested.XYZ and their squares; every channel is the same filtered sum over the
same samples, so the device delivers it in two renders of the same job (mi_render_cfg::moment_pass = 1: values,
2: squares). Checked here: the scalar restatement (which builds the eleven values of a sample and gives them one
validity verdict) == the staged loop == the device; the algebra of the channels; and the z-test itself between two
estimators of the same image."""
import numpy as np
import pytest
from scipy import stats

from conftest import has_gpu


def _passes(native, oracle, scene, sensor, integ):
    out = []
    for mp in (1, 2):
        job = native.MomentIntegrator(integ).render_job(sensor, moment_pass=mp)
        out.append(oracle.render(scene.desc(), job, threads=8)[0])
    return out


def test_moment_channels_from_the_two_passes(native, oracle):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(32, 24, 1, device=-1, rfilter="box")
    integ = native.PathIntegrator()
    plain, _, _ = oracle.render(scene.desc(), integ.render_job(sensor), threads=4)
    values, squares = _passes(native, oracle, scene, sensor, integ)
    assert np.array_equal(values, plain)                            # pass 1 is the ordinary film
    assert np.array_equal(squares[..., 3:], plain[..., 3:])         # alpha and weight ride along unchanged
    # one sample per pixel under the box filter: every texel holds exactly its own sample, so m2 = value^2
    assert np.array_equal(squares[..., :3], plain[..., :3] * plain[..., :3])
    assert native.MomentIntegrator(integ, name="nested").aov_names() == ["nested.X", "nested.Y", "nested.Z", "m2_nested.X", "m2_nested.Y", "m2_nested.Z"]


@pytest.mark.parametrize("nested", ["path", "direct"])
def test_moment_staged_equals_scalar(native, oracle, nested):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(40, 32, 6, diffuse_only=False, device=-1, ball_level=1)
    integ = native.PathIntegrator() if nested == "path" else native.DirectIntegrator(shading_samples=2)
    for mp in (1, 2):
        job = native.MomentIntegrator(integ).render_job(sensor, moment_pass=mp)
        o32, o64, st = oracle.render(scene.desc(), job, threads=4)
        for plan in ((1, 2) if nested == "path" else (2,)):
            job.cfg.plan = plan
            e64, e32, est = oracle.emu_render(scene.desc(), job)
            assert est[1] == st.segments and np.array_equal(e32, o32)
    # Jensen: E[x^2] >= E[x]^2 texel by texel (same weights in both sums; Cauchy-Schwarz with W)
    values, squares = _passes(native, oracle, scene, sensor, integ)
    w = values[..., 4:5]
    assert (squares[..., :3] * w >= values[..., :3] ** 2 * (1 - 1e-5)).all()


def test_overflowing_square_drops_the_whole_sample(native, oracle):
    """imageblock.cpp:85-109 tests all channels of a sample together: when a square overflows, the reference drops the
    sample from every channel — values, alpha and weight included — in both passes."""
    from mitsuba2_amd import scenes
    from test_textures import textured_quad, quad_sensor
    meshes = textured_quad(native, (0.5, 0.5, 0.5))
    meshes[1] = native.Mesh("light", meshes[1].vertices, meshes[1].faces, bsdf=native.BSDF("diffuse", reflectance=(0, 0, 0)),
                            emitter=native.AreaLight(radiance=(1e21, 1e21, 1e21)))     # lit floor: X ~ 1e20, X^2 = inf
    scene = native.Scene(meshes).build(-1)
    sensor = quad_sensor(native, 32, 24, 2)
    integ = native.DirectIntegrator(emitter_samples=1, bsdf_samples=0, hide_emitters=False)
    plain, _, _ = oracle.render(scene.desc(), integ.render_job(sensor), threads=2)
    values, squares = _passes(native, oracle, scene, sensor, integ)
    assert np.isfinite(values).all() and not np.isnan(squares).any()    # (sums of finite squares may still reach inf)
    assert np.array_equal(values[..., 4], squares[..., 4])
    assert plain[..., 4].sum() > values[..., 4].sum() > 0            # the brightly lit texels lost samples
    job = native.MomentIntegrator(integ).render_job(sensor, moment_pass=2)
    job.cfg.plan = 2
    e64, e32, _ = oracle.emu_render(scene.desc(), job)
    assert np.array_equal(e32, squares)


def test_negative_values_are_kept_under_the_moment_integrator(native, oracle):
    """SamplingIntegrator::render builds its ImageBlocks with warn_negative = !has_aovs (integrator.cpp:110-113): the plain
    path / direct integrators drop a sample with a channel below -1e-5 (imageblock.cpp:88-91), the moment integrator —
    whose block carries AOVs — only drops non-finite ones. A light of negative radiance makes the difference visible."""
    from test_textures import textured_quad, quad_sensor
    meshes = textured_quad(native, (0.5, 0.5, 0.5))
    meshes[1] = native.Mesh("light", meshes[1].vertices, meshes[1].faces, bsdf=native.BSDF("diffuse", reflectance=(0, 0, 0)),
                            emitter=native.AreaLight(radiance=(-2.0, -2.0, -2.0)))
    scene = native.Scene(meshes).build(-1)
    sensor = quad_sensor(native, 32, 24, 2)
    integ = native.DirectIntegrator(emitter_samples=1, bsdf_samples=0, hide_emitters=False)
    plain, _, pst = oracle.render(scene.desc(), integ.render_job(sensor), threads=2)
    values, squares = _passes(native, oracle, scene, sensor, integ)
    assert values[..., 4].sum() > plain[..., 4].sum() > 0              # the plain block rejected the negative samples
    assert values[..., 1].min() < -0.1 and squares[..., 1].max() > 0.005 and np.array_equal(values[..., 4], squares[..., 4])
    for mp, want in ((1, values), (2, squares)):                       # the device stages agree (log sink and splat sink)
        job = native.MomentIntegrator(integ).render_job(sensor, moment_pass=mp)
        job.cfg.plan = 2
        e64, e32, _ = oracle.emu_render(scene.desc(), job)
        assert np.array_equal(e32, want)


def _z_test(a_val, a_sq, b_val, b_sq, spp):
    """test_renders.py:63-132 for two renders: per-pixel means and variances from the moment channels, Welch-style
    z statistic on luminance, Sidak-corrected significance 0.01"""
    def mean_var(val, sq):
        w = val[..., 4]
        m = val[..., 1] / w
        v = np.maximum(sq[..., 1] / w - m * m, 0) / spp
        return m, v
    ma, va = mean_var(a_val, a_sq); mb, vb = mean_var(b_val, b_sq)
    z = (ma - mb) / np.sqrt(np.maximum(va + vb, 1e-12))
    p = 2 * stats.norm.sf(np.abs(z))
    alpha = 1 - (1 - 0.01) ** (1 / p.size)
    return (p > alpha).mean(), p


def test_z_test_between_two_estimators_and_a_wrong_one(native, oracle):
    """direct (4 + 4 samples) and path cut at one bounce estimate the same image; path with one more bounce does not."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(32, 24, 64, device=-1, rfilter="box")
    a = _passes(native, oracle, scene, sensor, native.DirectIntegrator(shading_samples=4))
    b = _passes(native, oracle, scene, sensor, native.PathIntegrator(max_depth=2))
    c = _passes(native, oracle, scene, sensor, native.PathIntegrator(max_depth=3))
    ok_ab, _ = _z_test(*a, *b, 64)
    ok_ac, _ = _z_test(*a, *c, 64)
    assert ok_ab > 0.995                                            # (Sidak: essentially every pixel passes)
    assert ok_ac < 0.9                                              # indirect light is missing from `a`: the test notices


def test_moment_xml_and_errors(native):
    xml = """<scene version="2.0.0">
      <integrator type="moment"><integrator type="direct" name="inner"><integer name="shading_samples" value="2"/></integrator></integrator>
      <shape type="rectangle"/></scene>"""
    scene, sensor, integ = native.load_string(xml)
    assert integ.aov_names()[0] == "inner.X" and integ.aov_names()[-1] == "m2_inner.Z"
    with pytest.raises(RuntimeError, match="nested"):
        native.load_string('<scene version="2.0.0"><integrator type="moment"/><shape type="rectangle"/></scene>')
    with pytest.raises(RuntimeError, match="nested moment"):
        native.MomentIntegrator(native.MomentIntegrator(native.PathIntegrator()))


# ---- device ------------------------------------------------------------------------------------------------
needs_gpu = pytest.mark.skipif(not has_gpu(), reason="needs a GPU")


@pytest.mark.gpu
@needs_gpu
def test_device_moment_passes_and_host_film(native, oracle):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(96, 64, 8, diffuse_only=False, device=0, ball_level=2)
    integ = native.MomentIntegrator(native.PathIntegrator(), name="path")
    dev = native.Device(0)
    try:
        dev.upload(scene.desc())
        films = []
        for mp in (1, 2):
            job = integ.render_job(sensor, moment_pass=mp)
            o32, _, ost = oracle.render(scene.desc(), job, threads=8)
            for plan in (1, 2):
                g32, st = dev.render(job, plan=plan)
                assert st == 0 and dev.counters().segments == ost.segments and np.array_equal(g32, o32)
            films.append(o32)
    finally:
        dev.close()
    assert integ.render(scene, sensor)                              # MomentIntegrator::render through the host layer
    film = sensor.film.data((64, 96, 11))
    assert np.array_equal(film[..., :5], films[0]) and np.array_equal(film[..., 5:8], films[0][..., :3])
    assert np.array_equal(film[..., 8:11], films[1][..., :3])
    rgb = sensor.film.develop()                                     # develop reads X Y Z A W of the wider film
    assert rgb.shape == (64, 96, 3) and np.isfinite(rgb).all()
