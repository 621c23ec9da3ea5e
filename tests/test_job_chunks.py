"""Pixel jobs cut into chunks of samples (device/resident_kernel.h: QueueWork::fetch; round 6).

A full frame of a packet kernel runs in rounds of one pixel per resident lane — every lane starts at once and a pixel's samples are one
serial job (integrator.cpp:196-209: one PCG32 stream per pixel) — so a frame of 6.33 pixels per lane lasts seven rounds. With chunk jobs
the queue holds (chunk, pixel) pairs, chunk-major, and a pixel changes lanes between two chunks: its 16-byte state word goes through
memory inside one launch. The film must not notice: the arithmetic per sample, the order of the samples of a pixel and the log index
stay what they were. GPU tier: forced on small frames (fewer pixels than lanes: all chunks of a pixel are drawn at once by different
lanes — often of the same wavefront — and every chunk but the first has to WAIT for the one before it: the hand-over path), several
chunk sizes, the kernel classes (plain diffuse 32-bit / 64-bit masks, BSDF dispatch, the phase machine over a tree) and the switch."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counters_carry_the_chunk_fields():
    from mitsuba2_amd import _capi
    names = [f[0] for f in _capi.mi_counters._fields_]
    assert names[-2:] == ["job_chunk", "job_chunks"]
    src = open(os.path.join(ROOT, "mitsuba2_amd", "csrc", "device", "resident_kernel.h")).read()
    # the hand-over protocol: the counter word is published after the RNG words have landed, and read before them, with agent-scope accesses
    assert src.index("__hip_atomic_store(words,") < src.index('asm volatile("s_waitcnt vmcnt(0)"') < src.index("__hip_atomic_store(words + 1,")
    assert "global_load_dwordx4 %0, %1, off sc1" in src              # ... and the reader takes the four words in one 16-byte load


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("kind", ["diffuse32", "diffuse64", "dispatch", "tree"])
def test_chunk_jobs_render_the_oracles_film(native, oracle, kind):
    from mitsuba2_amd import scenes
    from mitsuba2_amd import api
    if kind == "diffuse32":
        scene, sensor = scenes.cornell_box(64, 48, 32, device=-1)                                    # 32 triangles, plain diffuse: the headline's kernel
    elif kind == "diffuse64":                                                                       # 36 triangles: its 64-bit-mask twin
        v = np.array([(150, 330, 150), (250, 330, 150), (200, 330, 250), (200, 420, 190)], np.float32)
        f = np.array([(0, 1, 2), (0, 3, 1), (1, 3, 2), (2, 3, 0)], np.int32)
        meshes = scenes.cornell_box_meshes() + [api.Mesh("pyramid", v, f, bsdf=api.BSDF("diffuse", reflectance=(0.4, 0.5, 0.7)))]
        scene = api.Scene(meshes).build(-1)
        sensor = scenes.cornell_sensor(64, 48, 32, 0, "gaussian")
    elif kind == "tree":
        scene, sensor = scenes.cornell_box(64, 48, 32, device=-1, diffuse_only=False, ball_level=3)  # 2 572 triangles: the phase machine over the 8-wide tree (C3's kernel)
    else:
        scene, sensor = scenes.cornell_box(64, 48, 32, device=-1, diffuse_only=False, ball_level=0)  # BSDF dispatch (conductor + dielectric), still a packet scene
    job = native.PathIntegrator().render_job(sensor, n_threads=8)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    dev = native.Device(0)
    try:
        dev.upload(scene.desc())
        g, st = dev.render(job)                                       # a small frame: no chunks by default
        c = dev.counters()
        assert st == 0 and c.job_chunk == 0 and c.path_kernel == (1 if kind == "tree" else 0) and np.array_equal(g, o32)
        dev.set_option("MIW_JOB_CHUNK_FORCE", "1")
        g, st = dev.render(job)                                       # default chunk: spp / 8 but at least 64 samples — 32 spp is one chunk: a job = a pixel
        assert st == 0 and dev.counters().job_chunk == 0 and np.array_equal(g, o32)
        for chunk, chunks in ((8, 3), (16, 2), (1, 6)):               # halving chunks down to `chunk` samples: 16 + 8 + 8, 16 + 16, 16 + 8 + 4 + 2 + 1 + 1
            dev.set_option("MIW_JOB_CHUNK", str(chunk))
            g, st = dev.render(job)
            c = dev.counters()
            assert st == 0 and (c.job_chunk, c.job_chunks) == (chunk, chunks), (chunk, c.job_chunk, c.job_chunks)
            assert (c.samples, c.segments) == (ost.samples, ost.segments) and np.array_equal(g, o32), chunk
        dev.set_option("MIW_JOB_CHUNK", "0")                          # the switch
        g, st = dev.render(job)
        assert st == 0 and dev.counters().job_chunk == 0 and np.array_equal(g, o32)
        dev.set_option("MIW_JOB_CHUNK", "12")                         # not a power of two: refused (a job = a pixel)
        g, st = dev.render(job)
        assert st == 0 and dev.counters().job_chunk == 0 and np.array_equal(g, o32)
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_chunk_jobs_with_a_ragged_last_chunk_and_clipped_blocks(native, oracle):
    """spp not a power of two (the first chunk takes the odd part), a frame whose blocks are clipped (slots without a pixel are skipped in
    every chunk), a crop window."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(75, 53, 22, device=-1)
    job = native.PathIntegrator().render_job(sensor, n_threads=8)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    dev = native.Device(0)
    try:
        dev.upload(scene.desc())
        dev.set_option("MIW_JOB_CHUNK_FORCE", "1")
        for chunk, chunks in ((0, 0), (8, 3), (16, 2), (4, 4)):       # 22 spp: 6 + 8 + 8, 6 + 16, 6 + 8 + 4 + 4
            if chunk:
                dev.set_option("MIW_JOB_CHUNK", str(chunk))
            g, st = dev.render(job)
            c = dev.counters()
            if chunk:
                assert (c.job_chunk, c.job_chunks) == (chunk, chunks)
            else:
                assert c.job_chunk == 0                             # fewer than 128 samples per pixel: a job = a pixel
            assert st == 0 and (c.samples, c.segments) == (ost.samples, ost.segments) and np.array_equal(g, o32), chunk
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_chunk_jobs_of_the_direct_integrator(native, oracle):
    """direct.h: pixel_stream_render_direct draws the same queue (its packet kernels; 1 / 3+2 samples per shading point)."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 16, device=-1)
    dev = native.Device(0)
    try:
        dev.upload(scene.desc())
        dev.set_option("MIW_JOB_CHUNK_FORCE", "1")
        for kw in (dict(), dict(emitter_samples=3, bsdf_samples=2)):
            job = native.DirectIntegrator(**kw).render_job(sensor)
            o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
            for chunk, chunks in ((4, 3), (1, 5)):                    # 16 spp: 8 + 4 + 4; 8 + 4 + 2 + 1 + 1
                dev.set_option("MIW_JOB_CHUNK", str(chunk))
                g, st = dev.render(job)
                c = dev.counters()
                assert st == 0 and (c.job_chunk, c.job_chunks) == (chunk, chunks), (c.job_chunk, c.job_chunks)
                assert (c.samples, c.segments) == (ost.samples, ost.segments) and np.array_equal(g, o32), (kw, chunk)
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_chunk_jobs_in_the_spectral_variant(spectral, oracle_spectral):
    """the scalar_spectral library's packet kernel (C5's: glass block, 4 wavelengths per sample) and its phase machine."""
    from mitsuba2_amd import scenes
    for kw in (dict(glass_block=True), dict(diffuse_only=False, ball_level=2)):
        scene, sensor = scenes.cornell_box(64, 48, 16, device=-1, **kw)
        job = spectral.PathIntegrator().render_job(sensor, n_threads=8)
        o32, _, ost = oracle_spectral.render(scene.desc(), job, threads=8, want_f64=False)
        d = spectral.Device(0)
        try:
            d.upload(scene.desc())
            d.set_option("MIW_JOB_CHUNK_FORCE", "1")
            d.set_option("MIW_JOB_CHUNK", "2")
            g, st = d.render(job)
            c = d.counters()
            assert st == 0 and (c.job_chunk, c.job_chunks) == (2, 4), (c.job_chunk, c.job_chunks)      # 8 + 4 + 2 + 2
            assert (c.samples, c.segments) == (ost.samples, ost.segments) and np.array_equal(g, o32), kw
        finally:
            d.close()
