"""the analytic `rectangle` shape (src/shapes/rectangle.cpp) — one primitive that the
BVH builders see through two bounding triangles and that every scene query, the surface interaction and the
area-light sampling treat analytically. Pinned against closed forms (the reference's test_rectangle.py needs the
absent enoki); parity as everywhere else."""
import numpy as np
import pytest


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _scene(native, to_world, **kw):
    return native.Scene([native.Mesh.rectangle(to_world, **kw)]).build(-1)


def test_rectangle_record(native):
    """one shape, one primitive, MI_SHAPE_RECTANGLE, to_world and its inverse (flip_normals folded in)"""
    m = native.quad_to_world((1, 2, 3), (4, 0, 0), (0, 0, 2))
    scene = _scene(native, m)
    d = scene.desc().contents
    assert d.shape_count == 1 and d.face_count == 1 and d.rectangle_count == 1
    sh = d.shapes[0]
    assert sh.flags == 2 and sh.first_face == 0 and sh.face_count == 1
    r = d.rectangles[0]
    tw = np.array(r.to_world[:]).reshape(4, 4).T; to = np.array(r.to_object[:]).reshape(4, 4).T
    assert np.allclose(tw, m) and np.allclose(tw @ to, np.eye(4), atol=1e-6)
    assert np.allclose(tw @ [-1, -1, 0, 1], [1, 2, 3, 1]) and np.allclose(tw @ [1, 1, 0, 1], [5, 2, 5, 1])
    flipped = _scene(native, m, flip_normals=True)
    f = np.array(flipped.desc().contents.rectangles[0].to_world[:]).reshape(4, 4).T
    assert np.allclose(f[:, 2], -m[:, 2]) and np.allclose(f[:, [0, 1, 3]], m[:, [0, 1, 3]])


def test_rectangle_ray_intersect(native, oracle):
    """Rectangle::ray_intersect_preliminary / ray_test (rectangle.cpp:139-173): t, prim_uv = object-space (x, y),
    hits exactly the parallelogram, both sides, respects [mint, maxt]; BVH result == brute force"""
    m = native.quad_to_world((0, 0, 5), (4, 0, 0), (1, 3, 0))            # sheared quad in the z = 5 plane
    scene = _scene(native, m)
    rng = np.random.default_rng(1)
    n = 4000
    o = np.zeros((n, 3), np.float32); o[:, :2] = rng.uniform(-2, 7, (n, 2)); o[:, 2] = rng.choice([-1.0, 9.0], n)
    d = np.zeros((n, 3), np.float32); d[:, :2] = rng.normal(0, 0.15, (n, 2)); d[:, 2] = np.where(o[:, 2] < 5, 1, -1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    h = oracle.trace(scene.desc(), o, d)
    t_plane = (5 - o[:, 2]) / d[:, 2]
    p = o + d * t_plane[:, None]
    # object coordinates: p = corner + s * u + t * v  ->  x = 2 s - 1, y = 2 t - 1
    st = np.linalg.solve(np.array([[4, 1], [0, 3]], np.float64), (p[:, :2]).T).T
    inside = (st >= 0).all(1) & (st <= 1).all(1)
    margin = np.minimum(st, 1 - st).min(1)
    sure = np.abs(margin) > 1e-4
    hit = np.isfinite(h["t"])
    assert np.array_equal(hit[sure], inside[sure]) and 0.05 < inside.mean() < 0.8
    assert np.allclose(h["t"][hit], t_plane[hit], rtol=1e-5)
    assert np.allclose(h["u"][hit], 2 * st[hit, 0] - 1, atol=2e-5) and np.allclose(h["v"][hit], 2 * st[hit, 1] - 1, atol=2e-5)
    assert (h["prim"][hit] == 0).all()
    e = oracle.emu_trace(scene.desc(), o, d)
    for k in ("t", "u", "v", "prim"):
        assert np.array_equal(np.asarray(e[k]).view(np.uint32), np.asarray(h[k]).view(np.uint32))
    # segment limits and the any-hit form
    far = oracle.trace(scene.desc(), o, d, maxt=t_plane * 0.999)
    assert not np.isfinite(far["t"]).any()
    near = oracle.trace(scene.desc(), o, d, mint=t_plane * 1.001)
    assert not np.isfinite(near["t"]).any()
    a = oracle.trace(scene.desc(), o, d, any_hit=True)
    assert np.array_equal(np.isfinite(a["t"]) & (a["t"] == 0), hit)


def test_rectangle_surface_interaction_and_sampling(native, oracle):
    """compute_surface_interaction (:175-208): p = ray(t), n = frame normal, uv = (prim_uv + 1) / 2; area light:
    sample_position uniform over the quad with pdf 1 / area (:111-130) turned into a solid-angle density"""
    m = native.quad_to_world((0, 0, 5), (4, 0, 0), (1, 3, 0))
    scene = native.Scene([native.Mesh.rectangle(m, emitter=native.AreaLight((3, 2, 1)))]).build(-1)
    ok, si = oracle.ray_intersect_full(scene.desc(), [1.5, 1.0, 9.0, 0, 0, -1, 0, np.inf])
    assert ok
    t, p, n = si[0], si[1:4], si[4:7]
    assert np.isclose(t, 4) and np.allclose(p, [1.5, 1.0, 5.0]) and np.allclose(n, [0, 0, 1])
    # emitter sampling from a reference point above the quad: MI_EVAL_EMITTER_SAMPLE
    rng = np.random.default_rng(3)
    k = 2000
    inp = np.zeros((k, 5), np.float32); inp[:, 0:3] = (2.0, 1.0, 8.0); inp[:, 3:5] = rng.random((k, 2))
    out = oracle.eval(6, inp, scene.desc())
    d, dist, pdf, pos, nrm, val = out[:, 0:3], out[:, 3], out[:, 4], out[:, 5:8], out[:, 8:11], out[:, 11:14]
    s, tt = inp[:, 3], inp[:, 4]
    expect = np.array([0, 0, 5]) + s[:, None] * np.array([4, 0, 0]) + tt[:, None] * np.array([1, 3, 0])
    assert np.allclose(pos, expect, atol=1e-5) and np.allclose(nrm, [0, 0, 1])
    dd = expect - np.array([2.0, 1.0, 8.0]); r = np.linalg.norm(dd, axis=1)
    assert np.allclose(dist, r, rtol=1e-5) and np.allclose(d, dd / r[:, None], atol=1e-5)
    area = 12.0
    assert np.allclose(pdf, (1 / area) * r * r / np.abs(dd[:, 2] / r), rtol=1e-4)
    assert np.allclose(val, np.array([3, 2, 1]) / pdf[:, None], rtol=1e-4)
    # from below the quad (its back side) the light contributes nothing (area.cpp:131-136)
    inp[:, 0:3] = (2.0, 1.0, 2.0)
    assert not oracle.eval(6, inp, scene.desc())[:, 11:14].any()


def test_rect_box_emulator_equals_oracle(native, oracle):
    """Walls and light as analytic rectangles, blocks as meshes: wavefront lane stages and resident sample loop (BVH
    over the bounding triangles) == scalar oracle (brute force over the primitives), bit for bit"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.rect_box(48, 40, 6, device=-1)
    d = scene.desc().contents
    assert d.rectangle_count == 7 and d.face_count == 7 + 20
    job = native.PathIntegrator().render_job(sensor)
    o32, o64, st = oracle.render(scene.desc(), job, threads=4)
    e64, e32, est = oracle.emu_render(scene.desc(), job)
    assert est[0] == st.samples == 48 * 40 * 6 and est[1] == st.segments
    assert np.array_equal(e32, o32) and np.isfinite(o32).all()
    assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))
    job.cfg.plan = 2
    r64, r32, rst = oracle.emu_render(scene.desc(), job)
    assert rst[1] == st.segments and np.array_equal(r32, o32)
    assert st.segments / st.samples > 2.0 and o32[..., 1].sum() > 0


def test_xml_rectangle_is_analytic(native):
    scene, sensor, integ = native.load_string("""<scene version="2.0.0">
        <shape type="rectangle"><transform name="to_world"><scale x="2" y="3"/><translate z="1"/></transform>
            <boolean name="flip_normals" value="true"/></shape></scene>""")
    scene.build(-1)
    d = scene.desc().contents
    assert d.rectangle_count == 1 and d.face_count == 1 and d.shapes[0].flags == 2
    tw = np.array(d.rectangles[0].to_world[:]).reshape(4, 4).T
    assert np.allclose(tw, [[2, 0, 0, 0], [0, 3, 0, 0], [0, 0, -1, 1], [0, 0, 0, 1]])


@pytest.mark.gpu
def test_rect_box_device_equals_oracle(native, oracle):
    from mitsuba2_amd import scenes
    dev = native.Device(0)
    scene, sensor = scenes.rect_box(96, 80, 8, device=-1)
    job = native.PathIntegrator().render_job(sensor)
    dev.upload(scene.desc())
    o32, o64, ost = oracle.render(scene.desc(), job, threads=8)
    for plan in (1, 2):
        g32, st = dev.render(job, plan=plan)
        c = dev.counters()
        assert st == 0 and c.plan == plan and c.samples == ost.samples and c.segments == ost.segments
        assert np.array_equal(g32, o32), "plan %d: rel L2 %g" % (plan, rel_l2(g32, o32))
    # scene queries through mi_trace: closest hit and any hit == brute force, device LBVH too
    rng = np.random.default_rng(2)
    n = 100000
    o = rng.uniform(20, 530, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    ref = oracle.trace(scene.desc(), o, d)
    for quality in (1, 0, 0x40):
        dev.upload(scene.desc(), bvh_quality=quality)
        got = dev.trace(o, d)
        for k in ("t", "u", "v", "prim"):
            assert np.array_equal(np.asarray(got[k]).view(np.uint32), np.asarray(ref[k]).view(np.uint32)), (quality, k)
    assert np.isfinite(ref["t"]).mean() > 0.8
