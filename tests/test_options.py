"""The library's switches (include/miwave.h: mi_set_option) and the per-render debug overrides of mi_render_cfg.

VERDICT r05 (hygiene): a drop-in library whose kernel choice depends on its CALLER's environment at every render is fragile. The
environment is read ONCE, by mi_create, into the context; afterwards only mi_set_option / mi_render_cfg::debug_* change a choice.

CPU tier: the entry points exist and the option table is what INTEGRATION.md lists. GPU tier: an environment variable set after
mi_create changes nothing; the context's option and the per-render field both do; unknown names are refused; the pooled phase
machine (device/pooled_kernel.h, opt-in) renders the oracle's film in both of its workgroup shapes.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def test_option_entry_points_and_the_documented_table():
    from mitsuba2_amd import _capi
    lib = C.CDLL(os.path.join(_capi.LIB_DIR, "libmiwave.so"))
    lib.mi_option_count.restype = C.c_int32
    lib.mi_option_name.restype = C.c_char_p; lib.mi_option_name.argtypes = [C.c_int32]
    lib.mi_option_help.restype = C.c_char_p; lib.mi_option_help.argtypes = [C.c_int32]
    names = [lib.mi_option_name(i).decode() for i in range(lib.mi_option_count())]
    assert len(names) == len(set(names)) >= 30 and all(n.startswith("MIW_") for n in names)
    assert lib.mi_option_name(-1) is None and lib.mi_option_name(len(names)) is None
    assert all(lib.mi_option_help(i) for i in range(len(names)))
    # every switch the library knows is in INTEGRATION.md's table, and the table names nothing else
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    listed = set(re.findall(r"^\| `(MIW_[A-Z0-9_]+)`", doc, re.M))
    assert listed == set(names), (sorted(set(names) - listed), sorted(listed - set(names)))
    # ... and the library itself reads the environment in one place only
    src = open(os.path.join(ROOT, "mitsuba2_amd", "csrc", "miwave.hip")).read()
    assert src.count("getenv(") == 1 and "void from_env()" in src
    for extra in ("film_reduce.h", "device/trace.h", "device/phased_kernel.h", "device/pooled_kernel.h", "device/film_kernels.h"):
        assert "getenv(" not in open(os.path.join(ROOT, "mitsuba2_amd", "csrc", extra)).read(), extra


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_environment_is_read_once_and_options_belong_to_the_context(native, oracle, monkeypatch):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(48, 40, 4, diffuse_only=False, ball_level=3, device=-1)
    job = native.PathIntegrator().render_job(sensor, n_threads=8)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    monkeypatch.delenv("MIW_BVH8", raising=False)
    dev = native.Device(0)
    try:
        dev.upload(scene.desc())
        monkeypatch.setenv("MIW_BVH8", "0")                         # after mi_create: must not matter
        g, st = dev.render(job)
        assert st == 0 and dev.counters().tree_width == 8 and np.array_equal(g, o32)
        assert dev.get_option("MIW_BVH8") is None
        dev.set_option("MIW_BVH8", "0")
        assert dev.get_option("MIW_BVH8") == "0"
        g, st = dev.render(job)
        assert st == 0 and dev.counters().tree_width == 4 and np.array_equal(g, o32)
        g, st = dev.render(job, tree_width=8)                        # the per-render field wins over the context's option
        assert st == 0 and dev.counters().tree_width == 8 and np.array_equal(g, o32)
        dev.set_option("MIW_BVH8", None)
        with pytest.raises(RuntimeError, match="unknown option"):
            dev.set_option("MIW_NO_SUCH_SWITCH", "1")
        job.cfg.debug_film_replay = 99
        assert dev.L.mi_render(dev.ctx, C.byref(job.cfg), np.zeros(48 * 40 * 5, np.float32).ctypes.data_as(C.c_void_p)) != 0
        job.cfg.debug_film_replay = 0
        # a context created while the variable is set does take it (mi_create copies the environment)
        dev2 = native.Device(0)
        try:
            assert dev2.get_option("MIW_BVH8") == "0"
        finally:
            dev2.close()
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("shape", ["12x1", "8x2"])
def test_pooled_phase_machine_renders_the_oracles_film(native, oracle, shape):
    """device/pooled_kernel.h (round 6, opt-in): the walks of a workgroup's pixels as job records in LDS, advanced by whichever lane
    of the column is free — one pixel per lane (12 wavefronts) and two (8 wavefronts). Per-job arithmetic and order are the phase
    machine's, so the film is the oracle's bit for bit, with and without an environment map / analytic shapes / textures (the fuzz
    recipes cover those classes on the default kernel; here: the three kernel classes the pooled launch instantiates)."""
    import hashlib
    import json
    import sys
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(96, 64, 8, device=-1, diffuse_only=False, ball_level=3)
    job = native.PathIntegrator().render_job(sensor, n_threads=8)
    o32, _, ost = oracle.render(scene.desc(), job, threads=8, want_f64=False)
    dev = native.Device(0)
    try:
        dev.set_option("MIW_POOL_SHAPE", shape)
        dev.upload(scene.desc())
        g, st = dev.render(job, path_kernel=3)                       # MI_PATH_KERNEL_POOLED
        c = dev.counters()
        assert st == 0 and c.pooled == 1 and c.tree_width == 8 and c.pool_waves == int(shape.split("x")[0])
        assert (c.samples, c.segments) == (ost.samples, ost.segments)
        assert np.array_equal(g, o32)
        g2, st = dev.render(job, path_kernel=2)                      # ... and k_path_phased on the same context
        assert st == 0 and dev.counters().pooled == 0 and np.array_equal(g2, o32)
        # the fuzz recipes (random rooms: every plugin, analytic shapes, textures, environment maps, crop windows, depth limits) whose
        # default render runs the phase machine over the 8-wide tree: the pooled launch must reproduce the oracle's committed digest
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import fuzz_cpu
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "round3.json")))["fuzz"]
        pooled_runs = 0
        for seed in sorted(int(k) for k, v in gold.items() if "sha256" in v)[:24]:
            scene, sensor, ikw, recipe, keep = fuzz_cpu.make_case(native, scenes, seed)
            ikw = dict(ikw); ikw.pop("samples_per_pass", None)
            if ikw.pop("integrator", "path") == "direct":
                continue
            job = native.PathIntegrator(**ikw).render_job(sensor)
            dev.upload(scene.desc())
            film, st = dev.render(job, path_kernel=3)
            c = dev.counters()
            assert st == 0 and (c.samples, c.segments) == (gold[str(seed)]["samples"], gold[str(seed)]["segments"]), (seed, recipe)
            assert hashlib.sha256(np.ascontiguousarray(film, np.float32).tobytes()).hexdigest() == gold[str(seed)]["sha256"], (seed, c.pooled, recipe)
            pooled_runs += int(c.pooled)
        assert pooled_runs >= 6, pooled_runs                         # (scenes without a tree, or whose tables do not fit, keep their own kernels)
    finally:
        dev.close()
