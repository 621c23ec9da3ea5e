"""Random scenes (tools/fuzz_cpu.py: every BSDF plugin, meshes with and without shading normals / texture coordinates, analytic
spheres and rectangles, one to three area lights, optional environment map, crop windows, every reconstruction filter, depth
limits, passes). CPU tier: the staged emulator of the device kernels == the scalar restatement of the reference, bit for bit
(runs of the tool over 3 700 seeds found no mismatch; a sample runs here). The device-side counterpart on the same
recipes is tools/fuzz_gpu.py — to be run by hand on a GPU box first: a test that has never seen hardware does not belong in the
tier the driver runs unattended."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("first", [0, 1000])
def test_random_scenes_staged_emulator_equals_scalar_restatement(native, oracle, first):
    import fuzz_cpu
    from mitsuba2_amd import scenes
    for seed in range(first, first + 8):
        ok, recipe, segs = fuzz_cpu.run_case(native, scenes, oracle, seed)
        assert ok, (seed, segs, recipe)


@pytest.mark.parametrize("first", [7000, 8000])
def test_random_scenes_in_the_spectral_variant(spectral, oracle_spectral, first):
    """the same recipes under scalar_spectral (round 4: every plugin, bitmaps and the environment map exist there too): the CPU run
    of the resident sample loop == the spectral restatement, bit for bit"""
    import fuzz_cpu
    from mitsuba2_amd import scenes
    for seed in range(first, first + 6):
        ok, recipe, segs = fuzz_cpu.run_case(spectral, scenes, oracle_spectral, seed, resident_only=True)
        assert ok, (seed, segs, recipe)

