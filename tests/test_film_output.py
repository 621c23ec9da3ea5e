"""HDRFilm::develop — W-normalised XYZ -> linear sRGB, written as OpenEXR (uncompressed scanline,
float16 / float32) or PFM; read back here with independent minimal readers."""
import struct

import numpy as np
import pytest


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(x) for x in f.readline().split()]
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data[::-1]


def read_exr(path):
    """single-part, scanline, NO_COMPRESSION OpenEXR -> dict(channel -> H x W array)"""
    b = open(path, "rb").read()
    assert struct.unpack_from("<II", b, 0) == (20000630, 2)
    pos, attrs = 8, {}
    def cstr(p):
        e = b.index(b"\0", p); return b[p:e].decode(), e + 1
    while b[pos] != 0:
        name, pos = cstr(pos); typ, pos = cstr(pos); size = struct.unpack_from("<i", b, pos)[0]; pos += 4
        attrs[name] = (typ, b[pos:pos + size]); pos += size
    pos += 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    ch, p, raw = [], 0, attrs["channels"][1]
    while raw[p] != 0:
        e = raw.index(b"\0", p); nm = raw[p:e].decode(); p = e + 1
        ptype, = struct.unpack_from("<i", raw, p); p += 16
        ch.append((nm, {1: "<f2", 2: "<f4"}[ptype]))
    assert [c[0] for c in ch] == sorted(c[0] for c in ch)
    offsets = struct.unpack_from("<%dQ" % h, b, pos)
    out = {nm: np.zeros((h, w), np.float32) for nm, _ in ch}
    for y in range(h):
        o = offsets[y]
        yy, size = struct.unpack_from("<ii", b, o); o += 8
        assert yy == y
        for nm, dt in ch:
            n = w * np.dtype(dt).itemsize
            out[nm][y] = np.frombuffer(b[o:o + n], dt).astype(np.float32); o += n
    return out


@pytest.mark.parametrize("kw,ext", [(dict(), ".exr"), (dict(component_format="float32", pixel_format="rgba"), ".exr"),
                                    (dict(file_format="pfm"), ".pfm")])
def test_film_develop_round_trip(native, oracle, tmp_path, kw, ext):
    from mitsuba2_amd import api, scenes
    film = api.Film(width=40, height=24, **kw)
    sampler = api.Sampler(sample_count=4, seed=0)
    sensor = api.Sensor(film, sampler, fov=39.3, to_world=dict(origin=(278, 273, -800), target=(278, 273, 0), up=(0, 1, 0)))
    scene = api.Scene(scenes.cornell_box_meshes(True)).build(-1)
    job = native.PathIntegrator().render_job(sensor)
    o32, _, _ = oracle.render(scene.desc(), job, threads=4, want_f64=False)
    film.set_data(o32)                                              # what mi_render leaves in the film's storage
    path = film.develop_to(tmp_path / "out.png")                    # the extension is replaced (hdrfilm.cpp:336-338)
    assert path.endswith("out" + ext)
    xyz = o32[..., :3] / o32[..., 4:5]
    M = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], np.float32)
    rgb = xyz @ M.T
    assert np.allclose(film.develop(), rgb, rtol=1e-5, atol=1e-6)
    if ext == ".pfm":
        assert np.allclose(read_pfm(path), rgb, rtol=1e-5, atol=1e-6)
    else:
        img = read_exr(path)
        got = np.stack([img["R"], img["G"], img["B"]], 2)
        half = kw.get("component_format", "float16") == "float16"
        assert np.allclose(got, rgb, rtol=1e-3 if half else 1e-5, atol=1e-4 if half else 1e-6)
        if half:                                                    # float16 = round-to-nearest-even of the float32 image
            assert np.array_equal(got, film.develop().astype(np.float16).astype(np.float32))
        assert ("A" in img) == (kw.get("pixel_format") == "rgba")
        if "A" in img:
            assert np.allclose(img["A"], o32[..., 3] / o32[..., 4], rtol=1e-5)


def test_film_output_property_errors(native):
    with pytest.raises(RuntimeError, match="file_format"):
        native.Film(file_format="rgbe")
    with pytest.raises(RuntimeError, match="component_format"):
        native.Film(component_format="uint32")
    with pytest.raises(RuntimeError, match="Destination file"):
        native.Film(width=4, height=4).develop_to("")


def test_reconstruction_filters_spot_checks(native):
    """src/rfilters/tests/test_rfilter.py:8-52 (test01 .. test06), value for value"""
    def f(name):
        film = native.Film(rfilter=name, width=4, height=4)
        return lambda x: (film.rfilter(x)["eval"], film.rfilter(x)["eval_discretized"]), film.rfilter(0.0)
    ev, info = f("box")
    assert ev(0.49) == (1, 1) and ev(0.51) == (0, 0)
    ev, info = f("gaussian")
    assert np.allclose(ev(0.2), 0.9227, atol=8e-3) and ev(2.1) == (0, 0) and info["radius"] == 2.0 and info["border_size"] == 2
    ev, info = f("lanczos")
    assert np.allclose(ev(1.4), -0.14668, atol=1e-2) and ev(3.1) == (0, 0) and info["radius"] == 3.0
    ev, info = f("mitchell")
    assert np.allclose(ev(0), 0.8888, atol=1e-3) and ev(2.1) == (0, 0)
    ev, info = f("catmullrom")
    assert np.allclose(ev(0), 0.9765, atol=5e-2) and ev(2.1) == (0, 0)
    ev, info = f("tent")
    assert np.allclose(ev(0.1), 0.903, atol=5e-2) and ev(1.1) == (0, 0) and info["radius"] == 1.0
    with pytest.raises(RuntimeError, match="not found"):
        native.Film(rfilter="sinc", width=4, height=4)


@pytest.mark.parametrize("name", ["tent", "mitchell", "catmullrom", "lanczos"])
def test_every_filter_through_the_film_stages(native, oracle, name):
    """ImageBlock::put with each filter's table (negative lobes included): the scalar restatement's own put() ==
    the staged sample log + ordered replay the device runs (film_gather.h), both film modes"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(40, 32, 4, device=-1, rfilter=name)
    job = native.PathIntegrator().render_job(sensor)
    assert job.cfg.filter_radius == {"tent": 1.0, "mitchell": 2.0, "catmullrom": 2.0, "lanczos": 3.0}[name]
    o32, o64, st = oracle.render(scene.desc(), job, threads=4)
    for plan in (1, 2):
        job.cfg.plan = plan
        e64, e32, est = oracle.emu_render(scene.desc(), job)
        assert np.array_equal(e32, o32) and np.array_equal(e64.astype(np.float32), o64.astype(np.float32))
    if name != "tent":
        assert (np.asarray(job.cfg.filter_lut) < 0).any()           # the lobes are really in the table
    assert np.isfinite(o32).all() and (o32[..., 4] > 0).all()


# ---- device ------------------------------------------------------------------------------------------------
from conftest import has_gpu  # noqa: E402


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs a GPU")
@pytest.mark.parametrize("name", ["tent", "mitchell", "catmullrom", "lanczos"])
def test_device_film_with_every_filter(native, oracle, name):
    """the replay kernels take the filter as tables: k_film_groups over the 16-byte class records (weights per phase class)
    where the filter has them, k_film_blocks over the 24-byte position log otherwise (lanczos) — bit-identical to
    ImageBlock::put's order for every filter"""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(96, 64, 8, device=-1, rfilter=name)
    job = native.PathIntegrator().render_job(sensor)
    o32, o64, ost = oracle.render(scene.desc(), job, threads=8)
    dev = native.Device(0)
    try:
        dev.upload(scene.desc())
        for plan in (1, 2):
            g32, st = dev.render(job, plan=plan)
            assert st == 0 and dev.counters().film_mode == 1 and np.array_equal(g32, o32), (name, plan)
        g64, st = dev.render(job, f64=True, film_mode=2)
        assert st == 0 and np.array_equal(g64.astype(np.float32), o64.astype(np.float32))
    finally:
        dev.close()


def test_hdrfilm_construct_and_crops(native):
    """src/films/tests/test_hdrfilm.py:7-71 (test01_construct, test02_crops)"""
    assert native.Film().rfilter(0)["radius"] == 2.0                 # default reconstruction filter: gaussian
    film = native.Film()
    props = native.Properties("gaussian", stddev=18.5)
    assert native.host_lib().mih_film_set_filter(film.h, b"gaussian", props.h) == 0 and film.rfilter(0)["radius"] == 4 * 18.5
    for bad in (dict(component_format="uint8"), dict(pixel_format="brga")):
        with pytest.raises(RuntimeError):
            native.Film(**bad)
    film = native.Film(width=32, height=21, crop_width=11, crop_height=5, crop_offset_x=2, crop_offset_y=3,
                       high_quality_edges=True, pixel_format="rgba")
    assert film.crop_size() == (11, 5)
    incomplete = dict(width=32, height=21, crop_offset_x=30, crop_offset_y=20)
    with pytest.raises(RuntimeError, match="Invalid crop window"):
        native.Film(**incomplete)                                   # the crop size does not adjust itself
    assert native.Film(crop_width=2, crop_height=1, **incomplete).crop_size() == (2, 1)
