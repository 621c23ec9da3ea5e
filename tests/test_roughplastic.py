"""The rough plastic plugin (src/bsdfs/roughplastic.cpp): Beckmann / GGX coating over a diffuse base.

What the reference pins for it is statistical (src/bsdfs/tests/test_rough_plastic.py: chi-square of sample() against
pdf() for smooth / rough coatings at near-normal and grazing incidence); that harness is restated in test_chi2.py and
used here. On top of it: the quadrature rule against numpy's, the transmittance table against the smooth-interface
limit (where it must equal 1 - Fresnel), sample() == eval() / pdf(), energy conservation, and renders three ways
(scalar restatement == staged emulator == device)."""
import numpy as np
import pytest

from conftest import has_gpu
from test_chi2 import _bsdf_chi2, _threshold


def fresnel_dielectric(c, eta):
    s2 = (1 - c * c) / eta ** 2
    ct = np.sqrt(np.maximum(0, 1 - s2))
    a = (eta * c - ct) / (eta * c + ct); b = (c - eta * ct) / (c + eta * ct)
    return .5 * (a * a + b * b)


def test_gauss_legendre_rule(native):
    """src/libcore/quad.cpp:7-64 against numpy.polynomial.legendre.leggauss"""
    for n in (1, 2, 5, 32, 128):
        x, w = native.gauss_legendre(n)
        xr, wr = np.polynomial.legendre.leggauss(n)
        assert np.allclose(x, xr, atol=1e-7) and np.allclose(w, wr, atol=1e-7), n
        assert abs(w.sum() - 2) < 1e-6


@pytest.mark.parametrize("distribution", ["beckmann", "ggx"])
def test_transmittance_table_and_internal_reflectance(native, distribution):
    """roughplastic.cpp:336-371. A nearly smooth coating must reproduce the smooth interface: T(mu) = 1 - F(mu) and the
    internal reflectance the diffuse Fresnel reflectance seen from inside (plastic.cpp:163-165)."""
    eta = 1.49 / 1.000277
    smooth = native.BSDF("roughplastic", alpha=0.001, distribution=distribution)
    t = smooth.table()
    mu = np.maximum(1e-6, np.arange(64) / 63.0)
    assert t.shape == (64,) and np.abs(t[4:] - (1 - fresnel_dielectric(mu[4:], eta))).max() < 2e-4
    r = smooth.record()
    assert abs(r.params[1] - eta) < 1e-6 and abs(r.params[2] - 1 / eta ** 2) < 1e-6
    fdr_int = native.host_lib().mih_fresnel_diffuse_reflectance(1 / eta)
    assert abs(r.params[3] - fdr_int) < 0.03 * fdr_int              # (a 64-point mean of R(mu) mu against the closed form)
    rough = native.BSDF("roughplastic", alpha=0.3, distribution=distribution)
    tr = rough.table()
    assert (np.diff(tr[2:]) > -1e-4).all() and 0.3 < tr[2] < tr[-1] < 1        # more light enters at normal incidence
    assert np.abs(tr[8:] - t[8:]).max() > 5e-3                      # roughness changes the table
    assert rough.record().params[4] == pytest.approx(1 / 1.5)       # s_mean / (d_mean + s_mean) = 1 / (0.5 + 1)


def test_properties_and_errors(native):
    with pytest.raises(RuntimeError, match="anisotropic"):
        native.BSDF("roughplastic", alpha_u=0.1, alpha_v=0.2)
    with pytest.raises(RuntimeError, match="invalid distribution"):
        native.BSDF("roughplastic", distribution="phong")
    with pytest.raises(RuntimeError, match="must be positive and differ"):
        native.BSDF("roughplastic", int_ior=1.0, ext_ior=1.0)
    b = native.BSDF("roughplastic", distribution="ggx", nonlinear=True, specular_reflectance=(0.9, 0.8, 0.7), sample_visible=False)
    r = b.record()
    assert r.type == 6 and r.flags == (1 | 4 | 0x10)
    assert b.flags() == 0x8 | 0x2                                   # GlossyReflection | DiffuseReflection, :176-178
    assert native.BSDF("roughplastic").record().flags == 2          # beckmann, sample_visible


@pytest.mark.parametrize("kw", [dict(alpha=0.05), dict(alpha=0.4, distribution="ggx"), dict(alpha=0.3, nonlinear=True, diffuse_reflectance=(0.8, 0.3, 0.1)),
                                dict(alpha=0.2, sample_visible=False, specular_reflectance=(0.5, 0.6, 0.7))])
def test_sample_is_eval_over_pdf_and_energy_is_conserved(native, oracle, kw):
    bsdf = native.BSDF("roughplastic", **kw)
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    scene = native.Scene([native.Mesh("t", v, np.array([[0, 1, 2]], np.uint32), bsdf=bsdf)]).build(-1)
    rng = np.random.default_rng(3)
    n = 200000
    for wi in ([0, 0, 1], [0.6, 0.1, 0.79], [0.95, 0.2, 0.08]):
        wi = np.asarray(wi, np.float32); wi /= np.linalg.norm(wi)
        q = np.zeros((n, 10), np.float32); q[:, 1:4] = wi; q[:, 4:7] = rng.random((n, 3)); q[:, 7:10] = (0, 0, 1)
        out = oracle.eval(3, q, scene.desc())
        wo, pdf, weight = out[:, 0:3], out[:, 3], out[:, 6:9]
        ok = pdf > 0
        assert ok.mean() > 0.7 and (out[ok, 4] == 1).all()           # bs.eta = 1 (grazing non-visible sampling loses the most)
        q2 = q.copy(); q2[:, 7:10] = wo
        out2 = oracle.eval(3, q2, scene.desc())
        f, p2 = out2[:, 9:12], out2[:, 12]
        assert np.array_equal(p2[ok], pdf[ok])                      # bs.pdf = pdf(wo), :236
        assert np.allclose(weight[ok], f[ok] / pdf[ok, None], rtol=2e-6, atol=1e-7)
        albedo = weight.mean(axis=0)                                # E[f / p] = directional albedo
        assert (albedo < 1.0).all() and (albedo > 0.03).all(), albedo


@pytest.mark.parametrize("kw,wi", [(dict(alpha=0.05), [0.3, 0.2, 0.93]), (dict(alpha=0.5), [0.8, 0.3, 0.05]),
                                   (dict(alpha=0.3, distribution="ggx", sample_visible=False), [0.5, 0.0, 0.5])])
def test_chi2_roughplastic(native, oracle, kw, wi):
    """src/bsdfs/tests/test_rough_plastic.py: smooth / rough coating, near-normal / grazing incidence"""
    res = (256, 192) if kw.get("alpha", 1) < 0.1 else (64, 48)
    p, stat, dof, mass, frac = _bsdf_chi2(native, oracle, "roughplastic", wi, res=res, ires=8 if res[0] > 64 else 16, **kw)
    assert p > _threshold(), (p, stat, dof, mass, frac)
    assert abs(mass - frac) < 0.02


def _scene(native, **kw):
    from mitsuba2_amd import scenes
    meshes = scenes.cornell_box_meshes(False, 1)
    rp = native.BSDF("roughplastic", **kw)
    out = []
    for m in meshes:
        if m.name in ("floor", "back"):
            m = native.Mesh(m.name, m.vertices, m.faces, normals=m.normals, bsdf=rp)
        out.append(m)
    return native.Scene(out).build(-1)


def test_render_staged_equals_scalar(native, oracle):
    from mitsuba2_amd import scenes
    scene = _scene(native, alpha=0.15, distribution="ggx", diffuse_reflectance=(0.2, 0.5, 0.7))
    assert scene.desc().contents.bsdf_table_floats == 64
    sensor = scenes.cornell_sensor(40, 32, 4)
    for integ in (native.PathIntegrator(), native.DirectIntegrator(shading_samples=2)):
        job = integ.render_job(sensor)
        o32, o64, st = oracle.render(scene.desc(), job, threads=4)
        for plan in ((1, 2) if job.cfg.integrator == 0 else (2,)):
            job.cfg.plan = plan
            e64, e32, est = oracle.emu_render(scene.desc(), job)
            assert est[1] == st.segments and np.array_equal(e32, o32)
        assert np.isfinite(o32).all() and o32[..., 1].max() > 0
    plain, sensor2 = scenes.cornell_box(40, 32, 4, diffuse_only=False, device=-1, ball_level=1)
    p32, _, _ = oracle.render(plain.desc(), native.PathIntegrator().render_job(sensor), threads=4)
    o32, _, _ = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=4)
    assert not np.array_equal(p32, o32)


def test_twosided_roughplastic_and_xml(native, oracle):
    xml = """<scene version="2.0.0">
      <sensor type="perspective"><float name="fov" value="45"/>
        <transform name="to_world"><lookat origin="0, -3, 2" target="0, 0, 0" up="0, 0, 1"/></transform>
        <film type="hdrfilm"><integer name="width" value="24"/><integer name="height" value="16"/></film>
        <sampler type="independent"><integer name="sample_count" value="3"/></sampler></sensor>
      <shape type="rectangle"><bsdf type="twosided"><bsdf type="roughplastic"><float name="alpha" value="0.2"/>
        <rgb name="diffuse_reflectance" value="0.7, 0.2, 0.2"/></bsdf><bsdf type="roughplastic"><string name="distribution" value="ggx"/></bsdf></bsdf></shape>
      <shape type="sphere"><point name="center" x="0" y="0" z="1.5"/><float name="radius" value="0.3"/>
        <emitter type="area"><rgb name="radiance" value="30, 30, 30"/></emitter></shape>
    </scene>"""
    scene, sensor, integ = native.load_string(xml)
    scene.build(-1)
    d = scene.desc().contents
    assert d.bsdf_table_floats == 128 and {d.bsdfs[i].params[5] for i in range(d.bsdf_count) if d.bsdfs[i].type == 6} == {0.0, 64.0}
    job = integ.render_job(sensor)
    o32, _, st = oracle.render(scene.desc(), job, threads=2)
    job.cfg.plan = 2
    e64, e32, est = oracle.emu_render(scene.desc(), job)
    assert np.array_equal(e32, o32) and o32[..., 1].max() > 0


def test_spectral_roughplastic(spectral, oracle_spectral):
    """scalar_spectral: the diffuse base is an sRGB-upsampled spectrum, the coating is achromatic"""
    from mitsuba2_amd import scenes
    scene = _scene(spectral, alpha=0.2, diffuse_reflectance=(0.2, 0.5, 0.7), nonlinear=True)
    sensor = scenes.cornell_sensor(24, 16, 3)
    job = spectral.PathIntegrator(max_depth=4).render_job(sensor)
    o32, _, st = oracle_spectral.render(scene.desc(), job, threads=4)
    job.cfg.plan = 2
    e64, e32, est = oracle_spectral.emu_render(scene.desc(), job)
    assert est[1] == st.segments and np.array_equal(e32, o32) and o32[..., 1].max() > 0


# ---- device ------------------------------------------------------------------------------------------------
needs_gpu = pytest.mark.skipif(not has_gpu(), reason="needs a GPU")


@pytest.mark.gpu
@needs_gpu
def test_device_roughplastic_parity(native, oracle):
    from mitsuba2_amd import scenes
    dev = native.Device(0)
    try:
        for big in (False, True):
            scene = _scene(native, alpha=0.15, distribution="ggx", diffuse_reflectance=(0.2, 0.5, 0.7), nonlinear=big)
            if big:                                                # a tree scene: balls at subdivision level 3
                meshes = scenes.cornell_box_meshes(False, 3)
                rp = native.BSDF("roughplastic", alpha=0.3)
                meshes = [native.Mesh(m.name, m.vertices, m.faces, normals=m.normals, bsdf=rp) if m.name == "floor" else m for m in meshes]
                scene = native.Scene(meshes).build(-1)
            sensor = scenes.cornell_sensor(96, 64, 8)
            dev.upload(scene.desc())
            for integ in (native.PathIntegrator(), native.DirectIntegrator()):
                job = integ.render_job(sensor)
                o32, _, ost = oracle.render(scene.desc(), job, threads=8)
                for plan in ((1, 2) if job.cfg.integrator == 0 else (2,)):
                    g32, st = dev.render(job, plan=plan)
                    assert st == 0 and dev.counters().segments == ost.segments and np.array_equal(g32, o32), (big, plan)
        p = scene.desc(); p.contents.bsdfs[[i for i in range(p.contents.bsdf_count) if p.contents.bsdfs[i].type == 6][0]].params[5] = 1.0
        with pytest.raises(RuntimeError, match="transmittance table outside"):
            dev.upload(p)
    finally:
        dev.close()
