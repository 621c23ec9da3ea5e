"""Mesh texture coordinates (src/librender/mesh.cpp:492-511) and the bitmap texture (src/textures/bitmap.cpp).

The reference pins neither with exact vectors (its test_bitmap.py is a chi-square test of sample_position over an
image of the absent data repository); here the lookup is checked against an independent numpy restatement of
BitmapTextureImpl::interpolate, the surface interaction against the formulas of mesh.cpp, and renders three ways:
scalar restatement == staged emulator == device."""
import os

import numpy as np
import pytest

from conftest import has_gpu


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


# ---- an independent statement of bitmap.cpp:385-460 in float32 numpy ------------------------------------------
def np_wrap(v, res, mode):
    v = np.asarray(v, np.int64)
    if mode == "clamp":
        return np.clip(v, 0, res - 1)
    div = np.trunc(v / res).astype(np.int64)           # enoki::divisor: C division
    mod = v - div * res
    mod = np.where(mod < 0, mod + res, mod)
    if mode == "mirror":
        keep = ((div & 1) == 0) ^ (v < 0)
        mod = np.where(keep, mod, res - 1 - mod)
    return mod


def np_lookup(img, uv, filter_type="bilinear", wrap_mode="repeat", to_uv=None):
    f32 = np.float32
    img = np.asarray(img, f32)
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=2)
    h, w = img.shape[:2]
    uv = np.asarray(uv, f32)
    u, v = uv[:, 0], uv[:, 1]
    if to_uv is not None:
        m = np.asarray(to_uv, np.float64); u64, v64 = u.astype(np.float64), v.astype(np.float64)
        fma = lambda a, b, c: (a * b + c.astype(np.float64)).astype(f32)     # one rounding (float64 holds the product exactly)
        u, v = fma(m[0, 1], v64, fma(m[0, 0], u64, np.full_like(u, m[0, 3]))), fma(m[1, 1], v64, fma(m[1, 0], u64, np.full_like(v, m[1, 3])))
    if filter_type == "nearest":
        xi = np.floor(u * f32(w)).astype(np.int64); yi = np.floor(v * f32(h)).astype(np.int64)
        return img[np_wrap(yi, h, wrap_mode), np_wrap(xi, w, wrap_mode)]
    x = (u.astype(np.float64) * w - .5).astype(f32); y = (v.astype(np.float64) * h - .5).astype(f32)    # one rounding: fmadd
    xi = np.floor(x).astype(np.int64); yi = np.floor(y).astype(np.int64)
    w1x = (x - xi.astype(f32))[:, None]; w1y = (y - yi.astype(f32))[:, None]
    w0x = f32(1) - w1x; w0y = f32(1) - w1y
    x0, x1 = np_wrap(xi, w, wrap_mode), np_wrap(xi + 1, w, wrap_mode)
    y0, y1 = np_wrap(yi, h, wrap_mode), np_wrap(yi + 1, h, wrap_mode)
    v00, v10, v01, v11 = img[y0, x0], img[y0, x1], img[y1, x0], img[y1, x1]
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
    v0 = fma(w0x, v00, w1x * v10); v1 = fma(w0x, v01, w1x * v11)
    return fma(w0y, v0, w1y * v1)


def textured_quad(native, tex, **bsdf_kw):
    """a unit floor quad with texture coordinates, a textured diffuse BSDF and a light above it"""
    P = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    F = np.array([[0, 2, 1], [0, 3, 2]], np.uint32)
    T = np.array([[0, 0], [2, 0], [2, 2], [0, 2]], np.float32)          # the texture repeats twice across the quad
    floor = native.Mesh("floor", P, F, texcoords=T, bsdf=native.BSDF("diffuse", reflectance=tex, **bsdf_kw))
    L = np.array([[-.3, 1.5, -.3], [.3, 1.5, -.3], [.3, 1.5, .3], [-.3, 1.5, .3]], np.float32)
    light = native.Mesh("light", L, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), bsdf=native.BSDF("diffuse", reflectance=(0, 0, 0)),
                        emitter=native.AreaLight(radiance=(20, 20, 20)))
    return [floor, light]


def quad_sensor(native, w=48, h=32, spp=4):
    from mitsuba2_amd import scenes
    film = native.Film(rfilter="gaussian", width=w, height=h)
    return native.Sensor(film, native.Sampler(sample_count=spp, seed=3), fov=50.0,
                         to_world=dict(origin=(0, 2.2, 2.6), target=(0, 0, 0), up=(0, 1, 0)))


def checker(n=8, rgb=True, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.random((n, n, 3 if rgb else 1)).astype(np.float32) * 0.8 + 0.1
    return img if rgb else img[..., 0]


@pytest.mark.parametrize("filter_type", ["bilinear", "nearest"])
@pytest.mark.parametrize("wrap_mode", ["repeat", "mirror", "clamp"])
@pytest.mark.parametrize("rgb", [True, False])
def test_bitmap_lookup_equals_numpy_restatement(native, oracle, filter_type, wrap_mode, rgb):
    img = checker(6 if rgb else 5, rgb)
    to_uv = np.array([[1.5, .25, 0, .1], [-.5, 2.0, 0, -.3], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    tex = native.BitmapTexture(img, filter_type=filter_type, wrap_mode=wrap_mode, to_uv=to_uv)
    scene = native.Scene(textured_quad(native, tex)).build(-1)
    rng = np.random.default_rng(1)
    uv = (rng.random((4000, 2)) * 6 - 3).astype(np.float32)                # well outside [0, 1]: every wrap branch
    uv[:8] = [[0, 0], [1, 1], [.5, .5], [-1, 2], [1e-7, 1 - 1e-7], [-.25, .75], [2.5, -2.5], [.999999, 0]]
    q = np.zeros((len(uv), 4), np.float32); q[:, :2] = uv
    got = oracle.eval(12, q, desc=scene.desc())
    want = np_lookup(img, uv, filter_type, wrap_mode, to_uv)
    assert got.shape == want.shape
    assert np.array_equal(got, want)                 # same operations, same roundings: bit for bit


def test_bitmap_texture_properties_and_errors(native, tmp_path):
    with pytest.raises(RuntimeError, match="Invalid filter type"):
        native.BitmapTexture(checker(), filter_type="trilinear")
    with pytest.raises(RuntimeError, match="Invalid wrap mode"):
        native.BitmapTexture(checker(), wrap_mode="border")
    with pytest.raises(ValueError):
        native.BitmapTexture(np.zeros((4, 4, 2), np.float32))
    (w, h, c), mean = native.BitmapTexture(np.full((1, 1, 3), .25, np.float32)).info()
    assert (w, h, c) == (2, 2, 3) and np.allclose(mean, .25)              # bitmap.cpp:137-143: at least 2 x 2
    # PFM files: RGB and greyscale, either byte order, bottom-to-top scanlines
    img = checker(5)
    for name, arr, magic, scale, dt in (("a.pfm", img, "PF", -1.0, "<f4"), ("b.pfm", img, "PF", 1.0, ">f4"), ("c.pfm", img[..., 0], "Pf", -2.0, "<f4")):
        with open(tmp_path / name, "wb") as f:
            f.write(("%s\n%d %d\n%g\n" % (magic, arr.shape[1], arr.shape[0], scale)).encode())
            f.write((arr[::-1] / abs(scale)).astype(dt).tobytes())
        t = native.BitmapTexture(filename=str(tmp_path / name))
        (w, h, c), mean = t.info()
        assert (w, h, c) == (5, 5, arr.shape[2] if arr.ndim == 3 else 1)
        assert np.allclose(mean, arr.reshape(25, -1).mean(axis=0), atol=1e-6)
    with pytest.raises(RuntimeError, match="not a Portable Float Map"):
        (tmp_path / "x.pfm").write_bytes(b"P6\n1 1\n255\n")
        native.BitmapTexture(filename=str(tmp_path / "x.pfm"))


def test_texcoords_drive_uv_and_shading_frame(native, oracle):
    """mesh.cpp:492-511 against its formulas: si.uv interpolates the vertex uvs and dp_du solves the 2 x 2 system,
    which turns the shading frame's tangent (interaction.h:153-156)."""
    P = np.array([[0, 0, 0], [2, 0, 0], [0, 0, -1]], np.float32)
    F = np.array([[0, 1, 2]], np.uint32)
    T = np.array([[.1, .2], [.4, .9], [.8, .3]], np.float32)
    with_uv = native.Scene([native.Mesh("t", P, F, texcoords=T)]).build(-1)
    without = native.Scene([native.Mesh("t", P, F)]).build(-1)
    ray = np.array([[.5, 1, -.25, 0, -1, 0, 0, 10]], np.float32)
    def si_of(scene):
        ok, o = oracle.ray_intersect_full(scene.desc(), ray[0])
        assert ok == 1
        return dict(p=o[1:4], n=o[4:7], sh_n=o[7:10], sh_s=o[10:13], uv=o[19:21])
    a = si_of(with_uv); b = si_of(without)
    b1, b2 = b["uv"]                                                       # barycentrics without texcoords
    assert np.allclose(b["uv"], [.25, .25], atol=1e-6)
    want_uv = T[0] * (1 - b1 - b2) + T[1] * b1 + T[2] * b2
    assert np.allclose(a["uv"], want_uv, atol=1e-6)
    dp0, dp1 = P[1] - P[0], P[2] - P[0]; duv0, duv1 = T[1] - T[0], T[2] - T[0]
    det = duv0[0] * duv1[1] - duv0[1] * duv1[0]
    dp_du = (duv1[1] * dp0 - duv0[1] * dp1) / det
    n = a["sh_n"]
    s = dp_du - n * np.dot(n, dp_du); s /= np.linalg.norm(s)
    assert np.allclose(a["sh_s"], s, atol=1e-5) and not np.allclose(a["sh_s"], b["sh_s"], atol=1e-2)
    assert np.allclose(a["p"], b["p"]) and np.allclose(a["n"], b["n"])
    # a degenerate parameterisation keeps the coordinate_system() tangents (:506)
    flat = native.Scene([native.Mesh("t", P, F, texcoords=np.zeros((3, 2), np.float32))]).build(-1)
    c = si_of(flat)
    assert np.allclose(c["sh_s"], b["sh_s"]) and np.allclose(c["uv"], 0)


def test_obj_and_ply_texcoords(native, tmp_path):
    """test_mesh.py:136-175 (test06_load_various_features): uv present, OBJ flips v (obj.cpp:99,199), PLY does not"""
    (tmp_path / "q.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 0 1\nv 0 0 1\nvt 0.95 0.98\nvt 0.02 0.98\nvt 0.02 0.68\nvt 0.95 0.68\n"
                                    "f 1/1 2/2 3/3\nf 1/1 3/3 4/4\n")
    m = native.Mesh.load(str(tmp_path / "q.obj"))
    assert m.texcoords is not None and np.allclose(m.texcoords[0], [.95, 1 - .98]) and np.allclose(m.texcoords[2], [.02, 1 - .68])
    keep = native.Mesh.load(str(tmp_path / "q.obj"), flip_tex_coords=False)
    assert np.allclose(keep.texcoords[0], [.95, .98])
    (tmp_path / "q.ply").write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                                    "property float s\nproperty float t\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n"
                                    "0 0 0 0.1 0.2\n1 0 0 0.3 0.4\n0 0 1 0.5 0.6\n3 0 1 2\n")
    p = native.Mesh.load(str(tmp_path / "q.ply"))
    assert np.allclose(p.texcoords, [[.1, .2], [.3, .4], [.5, .6]])
    assert native.Mesh("n", m.vertices, m.faces).texcoords is None


def test_upload_validation_of_the_new_tables(native, oracle):
    tex = native.BitmapTexture(checker())
    scene = native.Scene(textured_quad(native, tex)).build(-1)
    d = scene.desc().contents
    assert d.bitmap_count == 1 and d.bitmaps[0].width == 8 and d.bsdfs[0].tex[0].type == 5 and d.vertex_texcoords
    assert d.shapes[0].flags & 8 and not d.shapes[1].flags & 8


@pytest.mark.parametrize("case", ["rgb", "grey_nearest_mirror", "conductor", "twosided"])
def test_textured_render_staged_equals_scalar(native, oracle, case):
    if case == "rgb":
        meshes = textured_quad(native, native.BitmapTexture(checker()))
    elif case == "grey_nearest_mirror":
        meshes = textured_quad(native, native.BitmapTexture(checker(7, rgb=False), filter_type="nearest", wrap_mode="mirror"))
    else:
        meshes = textured_quad(native, native.BitmapTexture(checker()))
        tex = native.BitmapTexture(checker(5, seed=2), wrap_mode="clamp")
        metal = native.BSDF("roughconductor", alpha=0.2, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), specular_reflectance=tex)
        if case == "twosided":
            metal = native.TwoSided(metal)
        meshes[0] = native.Mesh("floor", meshes[0].vertices, meshes[0].faces, texcoords=meshes[0].texcoords, bsdf=metal)
    scene = native.Scene(meshes).build(-1)
    sensor = quad_sensor(native)
    for integ in (native.PathIntegrator(max_depth=4), native.DirectIntegrator(shading_samples=2)):
        job = integ.render_job(sensor)
        o32, o64, st = oracle.render(scene.desc(), job, threads=4)
        for plan in ((1, 2) if job.cfg.integrator == 0 else (2,)):
            job.cfg.plan = plan
            e64, e32, est = oracle.emu_render(scene.desc(), job)
            assert est[1] == st.segments and np.array_equal(e32, o32)
        assert np.isfinite(o32).all() and o32[..., 1].max() > 0
    # the texture is really in the image: a constant BSDF of the mean colour renders differently
    plain = textured_quad(native, tuple(float(x) for x in checker().reshape(-1, 3).mean(axis=0)))
    plain_scene = native.Scene(plain).build(-1)                     # (desc() points into the scene object: keep it)
    p32, _, _ = oracle.render(plain_scene.desc(), native.PathIntegrator(max_depth=4).render_job(sensor), threads=4)
    if case in ("rgb",):
        o32, _, _ = oracle.render(scene.desc(), native.PathIntegrator(max_depth=4).render_job(sensor), threads=4)
        assert rel_l2(o32, p32) > 1e-2


def test_spectral_bitmap_texture(spectral, oracle_spectral):
    """scalar_spectral: texels become sRGB-model coefficients on the host (bitmap.cpp:156-165), the lookup evaluates
    the model at the four corner texels and blends the spectra (:439-450); a uniform image must equal the constant
    `srgb` texture of that colour."""
    col = (0.3, 0.6, 0.2)
    flat = np.broadcast_to(np.array(col, np.float32), (4, 4, 3)).copy()
    a = spectral.Scene(textured_quad(spectral, spectral.BitmapTexture(flat))).build(-1)
    b = spectral.Scene(textured_quad(spectral, col)).build(-1)
    sensor = quad_sensor(spectral, 32, 24, 3)
    job = spectral.PathIntegrator(max_depth=3).render_job(sensor)
    fa, _, sa = oracle_spectral.render(a.desc(), job, threads=4)
    fb, _, sb = oracle_spectral.render(b.desc(), job, threads=4)
    assert sa.segments == sb.segments and rel_l2(fa, fb) < 1e-6
    tex = spectral.BitmapTexture(checker())
    scene = spectral.Scene(textured_quad(spectral, tex)).build(-1)
    o32, _, st = oracle_spectral.render(scene.desc(), job, threads=4)
    job.cfg.plan = 2                                               # spectral builds run the resident plan only
    e64, e32, est = oracle_spectral.emu_render(scene.desc(), job)
    assert est[1] == st.segments and np.array_equal(e32, o32) and o32[..., 1].max() > 0
    with pytest.raises(RuntimeError, match="raw=true"):
        spectral.Scene(textured_quad(spectral, spectral.BitmapTexture(checker(), raw=True))).build(-1)


# ---- device ------------------------------------------------------------------------------------------------
needs_gpu = pytest.mark.skipif(not has_gpu(), reason="needs a GPU")


@pytest.fixture()
def dev(native):
    d = native.Device(0)
    yield d
    d.close()


@pytest.mark.gpu
@needs_gpu
@pytest.mark.parametrize("filter_type,wrap_mode", [("bilinear", "repeat"), ("bilinear", "mirror"), ("nearest", "clamp")])
def test_device_bitmap_lookup_bit_identical(native, oracle, dev, filter_type, wrap_mode):
    img = checker(6)
    to_uv = np.array([[1.5, .25, 0, .1], [-.5, 2.0, 0, -.3], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    tex = native.BitmapTexture(img, filter_type=filter_type, wrap_mode=wrap_mode, to_uv=to_uv)
    scene = native.Scene(textured_quad(native, tex)).build(-1)
    dev.upload(scene.desc())
    rng = np.random.default_rng(5)
    q = np.zeros((200000, 4), np.float32); q[:, :2] = (rng.random((200000, 2)) * 8 - 4).astype(np.float32)
    assert np.array_equal(dev.eval(12, q), oracle.eval(12, q, desc=scene.desc()))


@pytest.mark.gpu
@needs_gpu
@pytest.mark.parametrize("big", [False, True])
def test_device_textured_render_parity(native, oracle, dev, big):
    """packet scene (4 triangles) and tree scene (textured quad + icosphere clutter): every plan, both integrators"""
    from mitsuba2_amd import scenes
    meshes = textured_quad(native, native.BitmapTexture(checker()))
    tex = native.BitmapTexture(checker(5, seed=2), wrap_mode="clamp")
    metal = native.BSDF("roughconductor", alpha=0.2, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), specular_reflectance=tex)
    p, f, v = scenes.icosphere((0, .36, 0), .35, 3 if big else 0)
    uv = np.stack([np.arctan2(v[:, 2], v[:, 0]) / (2 * np.pi) + .5, np.arccos(np.clip(v[:, 1], -1, 1)) / np.pi], axis=1)
    meshes.append(native.Mesh("ball", p, f, normals=v, texcoords=uv, bsdf=metal))
    scene = native.Scene(meshes).build(-1)
    sensor = quad_sensor(native, 96, 64, 8)
    dev.upload(scene.desc())
    for integ in (native.PathIntegrator(max_depth=5), native.DirectIntegrator()):
        job = integ.render_job(sensor)
        o32, o64, ost = oracle.render(scene.desc(), job, threads=8)
        for plan in ((1, 2) if job.cfg.integrator == 0 else (2,)):
            g32, st = dev.render(job, plan=plan)
            c = dev.counters()
            assert st == 0 and c.plan == plan and c.segments == ost.segments
            assert np.array_equal(g32, o32), "plan %d rel L2 %g" % (plan, rel_l2(g32, o32))


@pytest.mark.gpu
@needs_gpu
def test_device_rejects_bad_texture_tables(native, dev):
    tex = native.BitmapTexture(checker())
    scene = native.Scene(textured_quad(native, tex)).build(-1)
    p = scene.desc(); d = p.contents
    d.bsdfs[0].tex[0].v[0] = 3.0
    with pytest.raises(RuntimeError, match="bitmap index out of range"):
        dev.upload(p)
    d.bsdfs[0].tex[0].v[0] = 0.0
    import ctypes
    keep = ctypes.cast(d.vertex_texcoords, ctypes.c_void_p).value          # (a pointer field read from a struct aliases it)
    d.vertex_texcoords = None
    with pytest.raises(RuntimeError, match="HAS_TEXCOORDS without vertex_texcoords"):
        dev.upload(p)
    d.vertex_texcoords = ctypes.cast(ctypes.c_void_p(keep), ctypes.POINTER(ctypes.c_float))
    d.bitmaps[0].channels = 2
    with pytest.raises(RuntimeError, match="Unsupported channel count"):
        dev.upload(p)
    d.bitmaps[0].channels = 3
    dev.upload(p)


@pytest.mark.gpu
@needs_gpu
def test_device_spectral_textured_render(spectral, oracle_spectral):
    scene = spectral.Scene(textured_quad(spectral, spectral.BitmapTexture(checker()))).build(-1)
    sensor = quad_sensor(spectral, 64, 48, 4)
    job = spectral.PathIntegrator(max_depth=4).render_job(sensor)
    d = spectral.Device(0)
    try:
        d.upload(scene.desc())
        o32, _, ost = oracle_spectral.render(scene.desc(), job, threads=8)
        g32, st = d.render(job)
        assert st == 0 and d.counters().segments == ost.segments and np.array_equal(g32, o32)
    finally:
        d.close()
