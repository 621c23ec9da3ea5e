"""The Scene query surface beyond ray_intersect_preliminary (include/mitsuba/render/scene.h:38-128):
Scene::ray_intersect -> SurfaceInteraction3f, Scene::sample_emitter_direction (with its visibility test),
Scene::pdf_emitter_direction, Endpoint::{sample_direction, pdf_direction, eval} — as named C-ABI entry points
(mi_ray_intersect, mi_sample_emitter_direction, mi_pdf_emitter_direction, mi_emitter_eval), as methods of the host
classes, and against the scalar oracle: bit-exact, every field.
"""
import numpy as np
import pytest


def _scenes(scenes):
    """(name, scene, sensor): triangle meshes with shading normals, texture coordinates + every (f)-4 plugin, analytic
    rectangles, analytic spheres (one of them an emitter), area light + environment map"""
    return [("balls", *scenes.cornell_box(64, 48, 1, diffuse_only=False, device=-1, ball_level=2)),
            ("plugins", *scenes.plugin_box(64, 48, 1, device=-1)),
            ("rects", *scenes.rect_box(64, 48, 1, device=-1)),
            ("spheres", *scenes.sphere_box(64, 48, 1, device=-1)),
            ("envmap", *scenes.open_box(64, 48, 1, device=-1, ball_level=2))]


def _camera_rays(sensor, n, seed):
    rng = np.random.default_rng(seed)
    rays = np.array([sensor.sample_ray(x, y) for x, y in rng.uniform(0, 1, (n, 2)).astype(np.float32)])
    return rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7]


def _bits_equal(a, b):
    """structured records: every field the same bits (NaN == NaN)"""
    for name in a.dtype.names:
        x, y = np.ascontiguousarray(a[name]), np.ascontiguousarray(b[name])
        if x.dtype.kind == "f":
            same = (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))
        else:
            same = x == y
        assert same.all(), (name, np.argwhere(~same)[:4], x[~same][:4], y[~same][:4])


def _second_bounce_rays(oracle, desc, o, d, mint, maxt, seed):
    """rays leaving the first hit points in random directions (incoherent, start on surfaces) + the camera rays"""
    si = oracle.ray_intersect(desc, o, d, mint, maxt)
    hit = np.isfinite(si["t"])
    rng = np.random.default_rng(seed)
    d2 = rng.normal(size=(hit.sum(), 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = si["p"][hit]
    m2 = (1 + np.abs(o2).max(1)) * np.float32(8.940697e-05)
    return (np.concatenate([o, o2]), np.concatenate([d, d2]), np.concatenate([mint, m2]).astype(np.float32),
            np.concatenate([maxt, np.full(len(o2), np.inf, np.float32)]).astype(np.float32)), si


# ---- CPU tier: the checker against itself and the host-side error behaviour ---------------------------------------

def test_oracle_query_surface_is_self_consistent(native, oracle):
    from mitsuba2_amd import scenes
    for name, scene, sensor in _scenes(scenes):
        desc = scene.desc()
        o, d, mint, maxt = _camera_rays(sensor, 96, 3)
        si = oracle.ray_intersect(desc, o, d, mint, maxt)
        pre = oracle.trace(desc, o, d, mint, maxt)
        assert np.array_equal(si["t"].view(np.uint32), pre["t"].view(np.uint32)) and np.array_equal(si["prim_index"], pre["prim"])
        for i in range(0, 96, 7):                                  # the older single-ray entry point
            ok, full = oracle.ray_intersect_full(desc, np.concatenate([o[i], d[i], [mint[i], maxt[i]]]))
            if ok:
                assert full[0] == si["t"][i] and np.array_equal(full[1:4], si["p"][i]) and np.array_equal(full[16:19], si["wi"][i])
        hit = np.isfinite(si["t"])
        assert hit.any() and np.allclose(np.linalg.norm(si["sh_n"][hit], axis=1), 1, atol=1e-5)
        n_em = scene.emitter_count()
        ref = si["p"][hit]; smp = np.random.default_rng(4).uniform(0, 1, (len(ref), 2)).astype(np.float32)
        for em in [-1] + list(range(n_em)):
            ds_v, sp_v = oracle.sample_emitter_direction(desc, ref, smp, True, em)
            ds_u, sp_u = oracle.sample_emitter_direction(desc, ref, smp, False, em)
            _bits_equal(ds_v, ds_u)                                # visibility only zeroes the spectrum
            shadow = oracle.trace(desc, ref, ds_u["d"], (1 + np.abs(ref).max(1)) * np.float32(8.940697e-05),
                                  ds_u["dist"] * np.float32(1 - 8.940697e-04), any_hit=True)
            occluded = np.isfinite(shadow["t"]) & (ds_u["pdf"] != 0)
            assert np.array_equal(sp_v[~occluded], sp_u[~occluded]) and not sp_v[occluded].any()
            # pdf of the sampled direction == the density the sample reports (area.cpp / envmap.cpp, both ways round)
            pdf = oracle.pdf_emitter_direction(desc, ref, ds_u, em)
            # (reference points ON an emitter are left out: a sphere light samples its surface by area from there but
            # reports the cone density, sphere.cpp:224-236 vs :248-262)
            ok = (ds_u["pdf"] > 0) & (sp_u.sum(1) > 0) & (si["emitter_index"][hit] < 0)
            assert ok.any() and np.allclose(pdf[ok], ds_u["pdf"][ok], rtol=2e-4), (name, em)
        # si.emitter(scene)->eval(si): non-zero exactly on front-facing emitter hits and environment misses
        ev = oracle.emitter_eval(desc, si)
        sees = si["emitter_index"] >= 0
        assert not ev[~sees].any() and (ev[sees].sum(1) > 0).sum() >= 0


def test_emitters_know_their_scene_and_contexts_are_checked(native):
    from mitsuba2_amd import scenes, api
    scene, sensor = scenes.open_box(32, 32, 1, device=-1, ball_level=1)
    assert scene.emitter_count() == 2                              # area light + environment map
    with pytest.raises(RuntimeError, match="not built on a device"):
        scene.ray_intersect(np.zeros((1, 3), np.float32), np.array([[0, 0, 1]], np.float32))
    with pytest.raises(RuntimeError, match="not built on a device"):
        scene.sample_emitter_direction([278, 100, 100], [0.3, 0.4])
    # BSDF::sample(ctx, si, ...) — bsdf.h:328-394: the full context is served (== the local-frame helper), others refused
    b = api.BSDF("diffuse", reflectance=(0.5, 0.4, 0.3))
    H = api.host_lib()
    wi = np.array([0.3, 0.2, 0.9], np.float32); wi /= np.linalg.norm(wi)
    s2 = np.array([0.25, 0.75], np.float32); out = np.zeros(9, np.float32)
    fp = lambda a: a.ctypes.data_as(api._capi.c_float_p)
    assert H.mih_bsdf_sample_ctx(b.h, 0, 0x1ff, 0xffffffff, fp(wi), 0.5, fp(s2), fp(out)) == 0
    bs = b.sample(wi, 0.5, s2)
    assert np.array_equal(out[0:3], bs["wo"]) and np.array_equal(out[6:9], bs["weight"]) and out[3] == bs["pdf"]
    for mode, mask, comp in ((1, 0x1ff, 0xffffffff), (0, 0x004, 0xffffffff), (0, 0x1ff, 0)):
        assert H.mih_bsdf_sample_ctx(b.h, mode, mask, comp, fp(wi), 0.5, fp(s2), fp(out)) == -1
        assert b"BSDFContext" in H.mih_last_error()


# ---- GPU tier: the C ABI and the host classes against the oracle ---------------------------------------------------

@pytest.mark.gpu
def test_ray_intersect_c_abi_equals_oracle(native, oracle):
    """mi_ray_intersect: every field of the SurfaceInteraction3f records, camera rays and incoherent second-bounce rays,
    packet sweep (<= 64 triangles) and tree walks, meshes with shading normals / texture coordinates and analytic shapes"""
    from mitsuba2_amd import scenes
    dev = native.Device(0)
    for name, scene, sensor in _scenes(scenes):
        desc = scene.desc()
        (o, d, mint, maxt), _ = _second_bounce_rays(oracle, desc, *_camera_rays(sensor, 3000, 5), seed=6)
        want = oracle.ray_intersect(desc, o, d, mint, maxt)
        for quality in (1, 0, 0x40, 1 | 0x10):                     # host / device SAH builder, radix tree, forced tree walk
            dev.upload(desc, bvh_quality=quality)
            _bits_equal(dev.ray_intersect(o, d, mint, maxt), want)
        assert np.isfinite(want["t"]).sum() > 2000 and (want["emitter_index"] >= 0).any(), name
    dev.close()


@pytest.mark.gpu
def test_emitter_queries_c_abi_equal_oracle(native, oracle):
    """mi_sample_emitter_direction (scene-level and per emitter, with and without the visibility test),
    mi_pdf_emitter_direction and mi_emitter_eval: bit-exact"""
    from mitsuba2_amd import scenes
    dev = native.Device(0)
    for name, scene, sensor in _scenes(scenes):
        desc = scene.desc()
        dev.upload(desc)
        o, d, mint, maxt = _camera_rays(sensor, 3000, 7)
        si = oracle.ray_intersect(desc, o, d, mint, maxt)
        hit = np.isfinite(si["t"])
        ref = si["p"][hit]; smp = np.random.default_rng(8).uniform(0, 1, (len(ref), 2)).astype(np.float32)
        for em in [-1] + list(range(scene.emitter_count())):
            for vis in (False, True):
                g_ds, g_sp = dev.sample_emitter_direction(ref, smp, vis, em)
                o_ds, o_sp = oracle.sample_emitter_direction(desc, ref, smp, vis, em)
                _bits_equal(g_ds, o_ds)
                assert np.array_equal(g_sp.view(np.uint32), o_sp.view(np.uint32)), (name, em, vis)
            assert np.array_equal(dev.pdf_emitter_direction(ref, o_ds, em).view(np.uint32),
                                  oracle.pdf_emitter_direction(desc, ref, o_ds, em).view(np.uint32))
        assert np.array_equal(dev.emitter_eval(si).view(np.uint32), oracle.emitter_eval(desc, si).view(np.uint32))
        assert (o_sp.sum(1) > 0).any() and (g_sp.sum(1) == 0).any(), name          # some visible, some occluded / back-facing
    dev.close()


@pytest.mark.gpu
def test_host_classes_serve_the_same_queries(native, oracle):
    """Scene::ray_intersect (batch and the Ray3f overload), Scene::sample_emitter_direction / pdf_emitter_direction,
    Emitter::sample_direction / pdf_direction / eval of the host layer == the oracle"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.open_box(64, 48, 1, device=0, ball_level=2)
    desc = scene.desc()
    o, d, mint, maxt = _camera_rays(sensor, 400, 9)
    want = oracle.ray_intersect(desc, o, d, mint, maxt)
    _bits_equal(scene.ray_intersect(o, d, mint, maxt), want)
    n_em = scene.emitter_count()
    assert n_em == 2
    rng = np.random.default_rng(10)
    for i in range(0, 400, 9):
        valid, rec, has_bsdf, em = scene.ray_intersect_one(np.concatenate([o[i], d[i], [mint[i], maxt[i]]]))
        _bits_equal(np.array([rec]), want[i:i + 1])
        assert valid == bool(np.isfinite(want["t"][i])) == has_bsdf and em == want["emitter_index"][i]
        found, spec = scene.emitter_eval(rec)
        assert found == (em >= 0) and np.array_equal(spec, oracle.emitter_eval(desc, want[i:i + 1])[0])
        if not valid:
            continue
        u = rng.uniform(0, 1, 2).astype(np.float32)
        for emitter in (-1, 0, 1):
            ds, spec = scene.sample_emitter_direction(rec["p"], u, test_visibility=True, emitter=emitter)
            o_ds, o_sp = oracle.sample_emitter_direction(desc, rec["p"], u, emitter < 0, emitter)   # Endpoint::sample_direction: no occluders
            _bits_equal(np.array([ds]), o_ds)
            assert np.array_equal(spec, o_sp[0])
            assert scene.pdf_emitter_direction(rec["p"], ds, emitter) == oracle.pdf_emitter_direction(desc, rec["p"], o_ds, emitter)[0]
