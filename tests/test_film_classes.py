"""Phase classes (csrc/miw/film.h, csrc/film_classes.h): the 16-byte sample record must reproduce ImageBlock::put.

The device logs, per finished sample, X Y Z and ONE BYTE per axis — the class of the sample's phase inside its pixel — instead of
the 8-byte position, and the film replay turns the byte back into the filter weights of the 5 x 5 texels around the pixel
(src/librender/imageblock.cpp:114-161). The table behind that byte is enumerated on the host over all 2^23 + 1 phases; this
file checks it against block_splat() — the restatement of put() the oracle uses — sample by sample: random positions, the edge
phases (samples on a pixel's boundary and exactly at its centre, where lo / hi jump), pixels at the origin (where float32
positions keep all 23 jitter bits) and far out (where they keep 13), every block corner, clipped edge blocks, crop windows,
block sizes down to 1. Films of whole renders through the records are compared in test_cpu_pipeline.py / test_fuzz.py (the
staged emulator uses the same record and replay as the device: oracle/wavefront_emu.cpp)."""
import ctypes as C

import numpy as np
import pytest


def _weights(native, oracle, sensor, n_threads, pos, pix):
    from mitsuba2_amd._capi import mi_render_cfg, c_float_p, c_i32_p
    job = native.PathIntegrator().render_job(sensor, n_threads=n_threads)
    L = oracle.L
    L.emu_film_weights.argtypes = [C.POINTER(mi_render_cfg), C.c_int, c_float_p, c_i32_p, c_float_p, c_float_p, c_i32_p]
    L.emu_film_weights.restype = C.c_int
    pos = np.ascontiguousarray(pos, np.float32); pix = np.ascontiguousarray(pix, np.int32)
    n = len(pos)
    a = np.zeros((n, 64), np.float32); b = np.zeros((n, 64), np.float32); reach = np.zeros(1, np.int32)
    count = L.emu_film_weights(C.byref(job.cfg), n, pos.ctypes.data_as(c_float_p), pix.ctypes.data_as(c_i32_p),
                               a.ctypes.data_as(c_float_p), b.ctypes.data_as(c_float_p), reach.ctypes.data_as(c_i32_p))
    return count, a, b, job.cfg


def _samples(cfg, n, seed):
    """positions as render_sample forms them (float32 pixel + jitter, integrator.cpp:242) + the edge phases"""
    g = np.random.default_rng(seed)
    x0, y0, w, h = int(cfg.crop_x), int(cfg.crop_y), int(cfg.crop_w), int(cfg.crop_h)
    px = g.integers(x0, x0 + w, n); py = g.integers(y0, y0 + h, n)
    bs = int(cfg.block_size)
    k = n // 6                                              # block corners and the film's corners
    px[:k] = x0 + np.minimum(w - 1, (g.integers(0, max(w // bs, 1) + 1, k) * bs - g.integers(0, 2, k)).clip(0))
    py[:k] = y0 + np.minimum(h - 1, (g.integers(0, max(h // bs, 1) + 1, k) * bs - g.integers(0, 2, k)).clip(0))
    jit = (g.integers(0, 1 << 23, (n, 2)).astype(np.float64) / (1 << 23))                  # pcg32_next_f32: multiples of 2^-23
    edge = np.array([0.0, 0.5, 1.0 - 2.0 ** -23, 0.5 - 2.0 ** -23, 0.5 + 2.0 ** -23, 2.0 ** -23, 0.25, 0.75])
    jit[k:3 * k] = edge[g.integers(0, len(edge), (2 * k, 2))]
    pos = np.stack([(px.astype(np.float32) + jit[:, 0].astype(np.float32)), (py.astype(np.float32) + jit[:, 1].astype(np.float32))], 1)
    return pos.astype(np.float32), np.stack([px, py], 1).astype(np.int32)


@pytest.mark.parametrize("rfilter", ["gaussian", "tent", "box", "mitchell", "catmullrom"])
@pytest.mark.parametrize("film", [dict(width=1920, height=1080), dict(width=77, height=45),
                                  dict(width=4000, height=3000, crop_offset_x=3000, crop_offset_y=2500, crop_width=150, crop_height=70),
                                  dict(width=300, height=200, crop_offset_x=1, crop_offset_y=2, crop_width=37, crop_height=19)])
def test_class_weights_equal_imageblock_put(native, oracle, rfilter, film):
    sensor = native.Sensor(native.Film(rfilter=rfilter, **film), native.Sampler(sample_count=1), fov=40.0)
    for n_threads in (1, 4096):                             # 32 x 32 blocks; blocks halved down to a pixel or two
        count, a, b, cfg = _weights(native, oracle, sensor, n_threads, *_samples(native.PathIntegrator().render_job(sensor, n_threads=n_threads).cfg, 6000, 5))
        assert 2 <= count <= 255
        assert not (b[:, 63] == -1.0).any()                 # put() never left the window the replay's lanes cover
        # texels outside the footprint get weight (w * 0): -0 under a negative lobe (mitchell, catmullrom) — value * -0 added to a
        # float32 sum leaves it as it is, like the +0 of a texel put() never touches; everywhere else: the same bits
        assert np.array_equal(a, b) and np.array_equal(a.view(np.uint32)[b != 0], b.view(np.uint32)[b != 0])
        # (box filter, border 0: a sample exactly on its block's left / top edge falls into texel -1 and is dropped, :163-170)
        assert (b.sum(1) > 0).all() if rfilter != "box" else (b.sum(1) > 0).mean() > 0.8


def test_filters_without_class_tables_keep_the_position_log(native, oracle):
    """lanczos (radius 3) is outside the enumeration: emu_film_weights reports -1 and the render logs positions (24 bytes)"""
    sensor = native.Sensor(native.Film(rfilter="lanczos", width=64, height=48), native.Sampler(sample_count=1), fov=40.0)
    count, _, _, _ = _weights(native, oracle, sensor, 1, np.zeros((1, 2), np.float32), np.zeros((1, 2), np.int32))
    assert count == -1
