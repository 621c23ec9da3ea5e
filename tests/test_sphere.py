"""The analytic `sphere` shape (src/shapes/sphere.cpp): double-precision quadratic, re-projected
surface interaction, cone sampling as an area light. Closed-form checks here (the reference's test_sphere.py needs the
absent enoki); parity as everywhere else."""
import numpy as np
import pytest
import test_chi2 as T


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_sphere_record(native):
    a = np.deg2rad(40.0)
    rot = np.array([[np.cos(a), -np.sin(a), 0, 3.0], [np.sin(a), np.cos(a), 0, -1.0], [0, 0, 1, 2.0], [0, 0, 0, 1]], np.float32)
    scene = native.Scene([native.Mesh.sphere((1, 0, 0), 2.5, to_world=rot, flip_normals=True)]).build(-1)
    d = scene.desc().contents
    assert d.sphere_count == 1 and d.face_count == 1 and d.shapes[0].flags == 4
    r = d.spheres[0]
    assert np.isclose(r.radius, 2.5) and r.flip_normals == 1
    assert np.allclose(r.center[:], (rot @ [1, 0, 0, 1])[:3], atol=1e-6)
    tw = np.array(r.to_world[:]).reshape(4, 4).T; to = np.array(r.to_object[:]).reshape(4, 4).T
    assert np.allclose(tw @ to, np.eye(4), atol=1e-6) and np.allclose(tw[:3, :3] @ tw[:3, :3].T, 2.5 ** 2 * np.eye(3), atol=1e-5)
    with pytest.raises(RuntimeError, match="non-uniform"):
        native.Mesh.sphere(to_world=np.diag([1.0, 2.0, 1.0, 1.0]).astype(np.float32))


def test_sphere_ray_intersect(native, oracle):
    """Sphere::ray_intersect_preliminary (:281-313): nearest root inside [mint, maxt], the far root when the ray starts
    inside, a miss when the segment lies entirely inside; BVH over the bounding triangles == brute force"""
    c, R = np.array([1.0, -2.0, 0.5]), 1.75
    scene = native.Scene([native.Mesh.sphere(c, R)]).build(-1)
    rng = np.random.default_rng(2)
    n = 6000
    o = rng.uniform(-5, 5, (n, 3)); o[: n // 3] = c + rng.normal(size=(n // 3, 3)) * 0.5      # a third start inside
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    h = oracle.trace(scene.desc(), o32, d32)
    oc = o32.astype(np.float64) - c.astype(np.float32).astype(np.float64); dd = d32.astype(np.float64)
    B = 2 * (oc * dd).sum(1); A = (dd * dd).sum(1); C = (oc * oc).sum(1) - np.float64(np.float32(R)) ** 2
    disc = B * B - 4 * A * C
    t0 = np.where(disc >= 0, (-B - np.sqrt(np.maximum(disc, 0))) / (2 * A), np.inf)
    t1 = np.where(disc >= 0, (-B + np.sqrt(np.maximum(disc, 0))) / (2 * A), -np.inf)
    expect = np.where(t0 >= 0, t0, np.where(t1 >= 0, t1, np.inf))
    expect[disc < 0] = np.inf
    hit = np.isfinite(expect)
    assert np.array_equal(np.isfinite(h["t"]), hit) and 0.2 < hit.mean() < 0.9
    assert np.allclose(h["t"][hit], expect[hit], rtol=1e-6)
    e = oracle.emu_trace(scene.desc(), o32, d32)
    for k in ("t", "prim"):
        assert np.array_equal(np.asarray(e[k]).view(np.uint32), np.asarray(h[k]).view(np.uint32))
    inside = np.linalg.norm(oc, axis=1) < R * 0.9
    seg = oracle.trace(scene.desc(), o32[inside], d32[inside], maxt=0.05)              # segment fully inside: no hit
    assert not np.isfinite(seg["t"]).any()
    a = oracle.trace(scene.desc(), o32, d32, any_hit=True)
    assert np.array_equal(np.isfinite(a["t"]), hit)


def test_sphere_surface_interaction(native, oracle):
    c, R = np.array([0.5, 1.0, -2.0], np.float32), np.float32(3.0)
    scene = native.Scene([native.Mesh.sphere(c, R)]).build(-1)
    ok, si = oracle.ray_intersect_full(scene.desc(), [0.5 + 1.2, 1.0 - 0.7, 10.0, 0, 0, -1, 0, np.inf])
    assert ok
    t, p, n = si[0], si[1:4], si[4:7]
    z = np.sqrt(9 - 1.2 ** 2 - 0.7 ** 2)
    assert np.allclose(p, [1.7, 0.3, -2 + z], atol=2e-6) and np.isclose(t, 12 - z, rtol=1e-6)
    assert np.allclose(n, (p - c) / R, atol=1e-6) and np.isclose(np.linalg.norm(p - c), R, rtol=1e-6)


@pytest.mark.parametrize("where", ["outside", "near", "inside"])
def test_chi2_sphere_light(native, oracle, where):
    """Sphere::sample_direction (:169-246) against the density it reports and against the closed form: uniform over the
    subtended cone from outside (Taylor branch for small cones), area sampling from inside"""
    c, R = np.array([0.0, 0.0, 0.0]), 1.0
    light = native.AreaLight((2.0, 3.0, 4.0))
    scene = native.Scene([native.Mesh.sphere(c, R, emitter=light)]).build(-1)
    ref = {"outside": np.array([0.3, -0.2, 2.5]), "near": np.array([40.0, 30.0, 20.0]), "inside": np.array([0.2, 0.1, -0.3])}[where]
    n = 400000
    rng = np.random.default_rng(8)
    inp = np.zeros((n, 5), np.float32); inp[:, 0:3] = ref; inp[:, 3:5] = rng.random((n, 2))
    out = oracle.eval(6, inp, scene.desc())
    d_s, dist, pdf_s, pos, nrm, val = out[:, 0:3].astype(np.float64), out[:, 3], out[:, 4], out[:, 5:8], out[:, 8:11], out[:, 11:14]
    assert np.allclose(np.linalg.norm(pos - c, axis=1), R, rtol=1e-5) and np.allclose(nrm, (pos - c) / R, atol=1e-5)
    assert np.allclose(np.linalg.norm(pos - ref, axis=1), dist, rtol=1e-5)
    dc = np.linalg.norm(c - ref)
    if where != "inside":
        cos_max = np.sqrt(1 - (R / dc) ** 2)
        assert np.allclose(pdf_s, 1 / (2 * np.pi * (1 - cos_max)), rtol=2e-3 if where == "near" else 1e-4)
        axis = (c - ref) / dc
        assert ((d_s @ axis) >= cos_max - 1e-5).all()                    # every sample inside the cone
        # front side only (dot(d, n) < 0): the area light contributes
        assert np.allclose(val * pdf_s[:, None], (2.0, 3.0, 4.0), rtol=1e-3)
        if where == "outside":
            # harness frame: polar axis = cone axis
            zz = axis; xx = np.cross(zz, [0, 1, 0]); xx /= np.linalg.norm(xx); yy = np.cross(zz, xx)
            Rm = np.stack([xx, yy, zz])                                  # scene -> harness
            pdf = lambda dirs: np.where(dirs[:, 2] >= cos_max, 1 / (2 * np.pi * (1 - cos_max)), 0.0)
            p, stat, dof, mass, frac = T.chi2_sphere(d_s @ Rm.T, pdf, n, res=(64, 64), ires=8, z_range=(cos_max, 1.0))
            assert p > T._threshold() and abs(mass - 1) < 2e-3 and frac > 0.9999, (p, stat, dof, mass, frac)
    else:
        # from inside the back faces are seen: dot(d, n) > 0, AreaLight::sample_direction returns zero (area.cpp:131-136)
        assert not val.any()
        area_pdf = (1 / (4 * np.pi * R * R)) * dist.astype(np.float64) ** 2 / np.abs((d_s * nrm).sum(1))
        assert np.allclose(pdf_s, area_pdf, rtol=1e-4)
        p, stat, dof, mass, frac = T.chi2_sphere(d_s, lambda dirs: _inside_pdf(dirs, ref, c, R), n, res=(64, 48), ires=8)
        assert p > T._threshold() and abs(mass - 1) < 2e-3, (p, stat, dof, mass, frac)


def _inside_pdf(dirs, ref, c, R):
    d = dirs.astype(np.float64); oc = ref - c
    b = (d * oc).sum(1); t = -b + np.sqrt(b * b - (oc @ oc - R * R))
    p = ref + d * t[:, None]; nrm = (p - c) / R
    return (1 / (4 * np.pi * R * R)) * t * t / np.abs((d * nrm).sum(1))


def test_sphere_box_emulator_equals_oracle(native, oracle):
    """sphere light + glass and rotated rough-conductor spheres: lane stages (BVH over bounding triangles) == oracle"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.sphere_box(48, 40, 6, device=-1)
    d = scene.desc().contents
    assert d.sphere_count == 3 and d.rectangle_count == 0
    job = native.PathIntegrator().render_job(sensor)
    o32, o64, st = oracle.render(scene.desc(), job, threads=4)
    e64, e32, est = oracle.emu_render(scene.desc(), job)
    assert est[0] == st.samples == 48 * 40 * 6 and est[1] == st.segments
    assert np.array_equal(e32, o32) and np.isfinite(o32).all() and o32[..., 1].sum() > 0
    job.cfg.plan = 2
    r64, r32, rst = oracle.emu_render(scene.desc(), job)
    assert rst[1] == st.segments and np.array_equal(r32, o32)


def test_xml_sphere(native):
    scene, sensor, integ = native.load_string("""<scene version="2.0.0">
        <shape type="sphere"><point name="center" x="1" y="2" z="3"/><float name="radius" value="0.5"/>
            <emitter type="area"><rgb name="radiance" value="5"/></emitter></shape></scene>""")
    scene.build(-1)
    d = scene.desc().contents
    assert d.sphere_count == 1 and d.emitter_count == 1 and np.allclose(d.spheres[0].center[:], (1, 2, 3)) and d.spheres[0].radius == 0.5


@pytest.mark.gpu
def test_sphere_box_device_equals_oracle(native, oracle):
    from mitsuba2_amd import scenes
    dev = native.Device(0)
    scene, sensor = scenes.sphere_box(96, 80, 8, device=-1)
    job = native.PathIntegrator().render_job(sensor)
    dev.upload(scene.desc())
    o32, o64, ost = oracle.render(scene.desc(), job, threads=8)
    for plan in (1, 2):
        g32, st = dev.render(job, plan=plan)
        c = dev.counters()
        assert st == 0 and c.plan == plan and c.samples == ost.samples and c.segments == ost.segments
        assert np.array_equal(g32, o32), "plan %d: rel L2 %g" % (plan, rel_l2(g32, o32))
    rng = np.random.default_rng(2)
    n = 100000
    o = rng.uniform(20, 530, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    ref = oracle.trace(scene.desc(), o, d)
    for quality in (1, 0, 0x40):
        dev.upload(scene.desc(), bvh_quality=quality)
        got = dev.trace(o, d)
        for k in ("t", "prim"):
            assert np.array_equal(np.asarray(got[k]).view(np.uint32), np.asarray(ref[k]).view(np.uint32)), (quality, k)
