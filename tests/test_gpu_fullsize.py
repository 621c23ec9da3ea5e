"""Oracle parity at the CONFIGURED sizes of BASELINE configs 3, 4 and 5 (tests/test_gpu_parity.py holds config 2's).

The full frames are hours of work for the brute-force oracle, so each test renders a crop window of the real job —
the config's own 1920x1080 sensor, its own triangle count, its own spp (C4: 64 of the 2048, the rest of the job being
more of the same sample streams) — on the device and requires the film to be BIT-IDENTICAL to the oracle's
(`oracle.render`: scalar control flow of path.cpp:100-211 / integrator.cpp:181-288, brute-force scene queries) and
the segment counts to be equal. Whole-frame properties of the same jobs (plans and trees agreeing with each other)
are in test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 1920, 1080
THREADS = os.cpu_count() or 8


def _crop_job(native, scenes, spp, x, y, w, h, n_threads):
    """The job of a crop window. `n_threads` is what SamplingIntegrator::render passes on as the worker count: the block
    size halves until there are at least that many blocks (integrator.cpp:88-97), which is also what lets the oracle —
    one worker per block, like the reference — spread a small window over the host's cores."""
    sensor = scenes.cornell_sensor(W, H, spp, crop_offset_x=x, crop_offset_y=y, crop_width=w, crop_height=h)
    return native.PathIntegrator().render_job(sensor, n_threads=n_threads)


def test_c3_crop_at_1024spp_equals_oracle(native, oracle):
    """Config 3: material balls (GGX rough conductor + bk7 dielectric, 40 972 triangles, shading normals), 1920x1080
    sensor, a 16x16 window at the full 1024 spp straddling the silhouettes of the glass and the metal ball
    (262 144 samples, ~1e11 brute-force triangle tests on the host): SAH tree, device LBVH, phase-machine and lock-step
    kernels, wavefront plan — every film equal to the oracle's."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(W, H, 1024, diffuse_only=False, device=-1)
    job = _crop_job(native, scenes, 1024, 968, 1000, 16, 16, n_threads=64)        # -> 64 blocks of 2 x 2 pixels
    o32, _, ost = oracle.render(scene.desc(), job, threads=THREADS, want_f64=False)
    assert ost.samples == 16 * 16 * 1024
    dev = native.Device(0)
    for quality in (0, 0x40):                                           # the SAH tree built on the device, the radix tree
        dev.upload(scene.desc(), bvh_quality=quality)
        c = dev.counters()
        assert c.bvh_tris == 40972 and c.bvh_builder == (1 if quality else 3)
        g, st = dev.render(job)
        c = dev.counters()
        assert st == 0 and c.plan == 2 and c.film_mode == 1
        assert c.samples == ost.samples and c.segments == ost.segments
        assert 0 < c.shadow_rays <= ost.shadow_rays          # the device does not trace shadow rays that carry a zero contribution
        assert np.array_equal(g, o32)
    p1, st = dev.render(job, plan=1)
    assert st == 0 and dev.counters().plan == 1 and np.array_equal(p1, o32)
    dev.close()


@pytest.mark.parametrize("quality,spp", [(0, 64), (0x40, 16)])
def test_c4_crop_on_the_1080p_sensor_equals_oracle(native, oracle, quality, spp):
    """Config 4 class: 911 362 triangles, area light + 1024x512 environment map, all three BSDFs with shading normals,
    the 1920x1080 sensor, a 24x16 window at 64 spp on the SAH tree (24 576 samples against brute force over 0.9 M
    triangles) and at 16 spp on the device LBVH (the same tree-independent answers, a quarter of the host time)."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.interior_scene(W, H, spp, device=-1)
    job = _crop_job(native, scenes, spp, 948, 700, 24, 16, n_threads=96)           # -> 96 blocks of 2 x 2 pixels
    o32, _, ost = oracle.render(scene.desc(), job, threads=THREADS, want_f64=False)
    dev = native.Device(0)
    dev.upload(scene.desc(), bvh_quality=quality)
    c = dev.counters()
    assert c.bvh_tris == 911362 and c.bvh_builder == (1 if quality else 3)
    g, st = dev.render(job)
    c = dev.counters()
    assert st == 0 and c.samples == ost.samples == 24 * 16 * spp and c.segments == ost.segments
    assert c.path_kernel == 1                                          # the wave-level phase machine (k_path_phased)
    assert np.array_equal(g, o32)
    dev.close()


def test_c5_spectral_crop_at_512spp_equals_oracle(spectral, oracle_spectral):
    """Config 5: scalar_spectral Cornell box with a (constant-IOR) dielectric block, 1920x1080 sensor, a 64x64 window
    over the block's edge at the full 512 spp (2.1 M samples, 4 wavelengths each)."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(W, H, 512, diffuse_only=True, glass_block=True, device=-1)
    job = _crop_job(spectral, scenes, 512, 930, 810, 64, 64, n_threads=64)       # -> 64 blocks of 8 x 8 pixels
    o32, _, ost = oracle_spectral.render(scene.desc(), job, threads=THREADS, want_f64=False)
    assert ost.samples == 64 * 64 * 512
    dev = spectral.Device(0)
    dev.upload(scene.desc())
    g, st = dev.render(job)
    c = dev.counters()
    assert st == 0 and c.plan == 2 and c.film_mode == 1 and c.samples == ost.samples and c.segments == ost.segments
    assert np.array_equal(g, o32)
    # the glass block is in the window: a good share of the paths refract (longer than the all-diffuse box's 3.4)
    assert c.segments / c.samples > 3.4
    dev.close()
