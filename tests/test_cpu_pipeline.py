"""CPU-side tests: the wavefront decomposition vs the scalar oracle (through the CPU
emulator of the product's lane stages), host logic, the C-ABI surface without a GPU,
golden fixtures, and the world_size-2 (gloo) sharding path."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel_l2(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _both(native, oracle, scene, sensor, **kw):
    job = native.PathIntegrator(**kw).render_job(sensor)
    o32, o64, st = oracle.render(scene.desc(), job, threads=4)
    e64, e32, est = oracle.emu_render(scene.desc(), job)
    return job, o32, o64, st, e64, e32, est


@pytest.mark.parametrize("diffuse_only", [True, False])
def test_wavefront_stages_equal_scalar_path_integrator(native, oracle, diffuse_only):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(48, 40, 6, diffuse_only=diffuse_only, device=-1, ball_level=1)
    job, o32, o64, st, e64, e32, est = _both(native, oracle, scene, sensor)
    assert est[0] == st.samples == 48 * 40 * 6 and est[1] == st.segments
    assert np.array_equal(e32, o32)                                 # ordered gather == reference-order float32 film
    assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))   # immediate splat == exact-sum film
    assert rel_l2(o32, o64) < 1e-5
    assert st.segments / st.samples > 2.0 and np.isfinite(o32).all() and o32[..., 4].min() > 0


@pytest.mark.parametrize("metal", [dict(distribution="beckmann"), dict(distribution="beckmann", sample_visible=False),
                                   dict(distribution="beckmann", alpha_u=0.05, alpha_v=0.3),
                                   dict(distribution="ggx", alpha_u=0.05, alpha_v=0.3, sample_visible=False)])
def test_rough_conductor_variants(native, oracle, metal):
    """roughconductor.cpp defaults to Beckmann (:167-169): both distributions, (an)isotropic, with and
    without visible-normal sampling, through the wavefront stages == the scalar oracle."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(40, 32, 6, diffuse_only=False, device=-1, ball_level=1, metal=metal)
    job, o32, o64, st, e64, e32, est = _both(native, oracle, scene, sensor)
    assert est[1] == st.segments and np.array_equal(e32, o32) and np.isfinite(o32).all()
    ggx, _ = scenes.cornell_box(40, 32, 6, diffuse_only=False, device=-1, ball_level=1)
    g32, _, _ = oracle.render(ggx.desc(), job, threads=4)
    assert not np.array_equal(g32, o32)                             # the variant really changes the image


@pytest.mark.parametrize("kw", [dict(), dict(with_area_light=False), dict(envmap_after=0), dict(env_scale=0.25)])
def test_environment_map_scene(native, oracle, kw):
    """Environment emitter (src/emitters/envmap.cpp): misses see the map (scene.h:248-249), it is sampled through
    the hierarchical warp with MIS against BSDF sampling, alone or mixed with an area light, at either position
    in the emitter order — wavefront stages and the resident sample loop both == the scalar oracle."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.open_box(40, 32, 6, device=-1, **kw)
    job, o32, o64, st, e64, e32, est = _both(native, oracle, scene, sensor)
    assert est[1] == st.segments and np.array_equal(e32, o32) and np.isfinite(o32).all()
    assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))
    job.cfg.plan = 2
    r64, r32, rst = oracle.emu_render(scene.desc(), job)
    assert rst[1] == st.segments and np.array_equal(r32, o32)
    top = o32[:8, :, 1] / o32[:8, :, 4]                             # rays leaving through the open top see sky
    assert top.mean() > 0.05
    if kw.get("env_scale"):
        full, _ = scenes.open_box(40, 32, 6, device=-1)
        f32, _, _ = oracle.render(full.desc(), native.PathIntegrator().render_job(sensor), threads=4)
        assert not np.array_equal(f32, o32)


@pytest.mark.parametrize("per_launch", [1, 4, 5, 64])
def test_resident_plan_equals_scalar_path_integrator(native, oracle, per_launch):
    """pixel_render (the resident plan's per-pixel sample loop), advanced in passes of `per_launch`
    samples with only the PCG32 state carried between passes, == the scalar oracle, bit for bit."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(40, 36, 10, diffuse_only=False, device=-1, ball_level=1)
    job = native.PathIntegrator().render_job(sensor)
    job.cfg.plan = 2; job.cfg.samples_per_launch = per_launch
    o32, o64, st = oracle.render(scene.desc(), job, threads=4)
    e64, e32, est = oracle.emu_render(scene.desc(), job)
    assert est[0] == st.samples == 40 * 36 * 10 and est[1] == st.segments
    assert est[3] == -(-10 // per_launch)                           # launches
    assert np.array_equal(e32, o32)
    assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))


def test_phantom_grazing_hit_is_structure_independent(native, oracle):
    """A shadow ray found by a 1.3e8-sample render (tools/diag_plans.py): it leaves a side face of the short block at
    grazing angle, 0.007 below the top edge; Moeller-Trumbore (ill-conditioned: det ~ 1e-4 |e1||e2|) "re-hits" that
    face at t = 0.035 > mint although the ray left the face's bounds at t = 0.013. Brute force used to report the
    hit, every spatial structure (leaf filter, both tree walks) to miss it. The accept rule of shape.h (the hit
    point must lie inside the triangle's bounds grown by accept_pad) makes all of them agree."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 1, device=-1)
    o = np.array([[110.197144, 164.992737, 233.387756]], np.float32)
    d = np.array([[0.28888461, 0.953493118, 0.0860041305]], np.float32)
    mint, maxt = np.float32(0.0209558979), np.float32(401.329437)
    dsc = scene.desc().contents
    tri = np.array([[dsc.vertex_positions[3 * dsc.faces[3 * 20 + k] + a] for a in range(3)] for k in range(3)], np.float32)
    import ctypes as C
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    out = np.zeros(4, np.float32)
    oracle.L.orc_ray_triangle(fp(np.ascontiguousarray(tri.reshape(-1))), fp(np.concatenate([o[0], d[0], [mint, maxt]]).astype(np.float32)), fp(out))
    hit, t, u, v = out
    assert hit and abs(t - 0.0352677) < 1e-5 and u + v > 0.9999          # the raw triangle test accepts the phantom
    assert (o[0] + t * d[0])[1] > 165.02                                # ... 0.026 above the face's top edge (pad: 0.0056)
    for any_hit in (False, True):
        b = oracle.trace(scene.desc(), o, d, mint=mint, maxt=maxt, any_hit=any_hit)
        for leaf in (2, 4):
            e = oracle.emu_trace(scene.desc(), o, d, mint=mint, maxt=maxt, any_hit=any_hit, max_leaf=leaf)
            assert np.array_equal(np.asarray(e["t"]).view(np.uint32), np.asarray(b["t"]).view(np.uint32))
        assert not np.isfinite(b["t"]).any()                            # unoccluded under the rule


def test_samples_per_pass(native, oracle):
    """samples_per_pass < sample_count (integrator.cpp:75-86): every pass re-seeds the pixels from its own block ids
    (spiral.cpp:41: counter + (remaining_passes - 1) * block_count) and adds its blocks onto the film after the passes already there."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(40, 36, 6, device=-1)
    integ = native.PathIntegrator(samples_per_pass=2)
    assert integ.pass_count(sensor) == 3
    one = native.PathIntegrator().render_job(sensor)
    assert one.cfg.spp == 6 and one.cfg.accumulate == 0
    nblocks = one.cfg.block_count
    o32 = o64 = e32 = e64 = None
    segments = 0
    for p in range(3):
        job = integ.render_job(sensor, pass_index=p)
        assert job.cfg.spp == 2 and job.cfg.accumulate == (1 if p else 0) and job.cfg.block_count == nblocks
        # spiral.cpp:41: the first pass rendered carries the highest id offset (remaining_passes - 1), the last one 0
        assert np.array_equal(job.block_ids[:nblocks], one.block_ids[:nblocks] + (2 - p) * nblocks)
        o32, o64, st = oracle.render(scene.desc(), job, threads=4, onto=(o32, o64))
        e64, e32, est = oracle.emu_render(scene.desc(), job, onto=(e64, e32))
        assert est[1] == st.segments and np.array_equal(e32, o32)
        assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))
        segments += st.segments
    w1 = oracle.render(scene.desc(), integ.render_job(sensor, pass_index=0), threads=4)[0][..., 4]
    assert abs(o32[..., 4].sum() / (3 * w1.sum()) - 1) < 0.02      # three passes' worth of filter weights
    s32, _, sst = oracle.render(scene.desc(), one, threads=4)       # the single-pass render: other seeds, same estimator
    assert not np.array_equal(s32, o32)
    assert abs(o32[..., 1].sum() / s32[..., 1].sum() - 1) < 0.1 and abs(segments / sst.segments - 1) < 0.05
    with pytest.raises(RuntimeError, match="multiple of samples_per_pass"):
        native.PathIntegrator(samples_per_pass=4).render_job(sensor)


def test_pass_shards_sum_to_the_multi_pass_film(native, oracle):
    """bench.py --shard passes: rank r renders pass r of the samples_per_pass = spp / N run into its own film
    (accumulate = 0), one reduce adds them. Sum of the pass films == the film the passes build one onto the other
    (the reference's order), up to the rounding of a different float32 summation order; exactly in float64."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(40, 36, 8, device=-1)
    integ = native.PathIntegrator(samples_per_pass=2)
    seq32 = seq64 = None
    parts32, parts64 = [], []
    for p in range(4):
        job = integ.render_job(sensor, pass_index=p)
        seq32, seq64, _ = oracle.render(scene.desc(), job, threads=4, onto=(seq32, seq64))
        job.cfg.accumulate = 0
        a32, a64, st = oracle.render(scene.desc(), job, threads=4)
        assert st.samples == 40 * 36 * 2
        parts32.append(a32); parts64.append(a64)
    total = np.sum(np.asarray(parts32, np.float64), axis=0)
    assert rel_l2(total, seq32) < 1e-6 and np.allclose(total, seq32, rtol=2e-6, atol=1e-7)
    assert np.allclose(np.sum(parts64, axis=0), seq64, rtol=1e-12)
    assert not np.array_equal(parts32[0], parts32[1])               # different seeds per pass (spiral.cpp:41)


@pytest.mark.parametrize("kw", [dict(max_depth=1), dict(max_depth=2), dict(max_depth=3, rr_depth=1), dict(rr_depth=2)])
def test_depth_and_rr_variants(native, oracle, kw):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(32, 24, 4, device=-1)
    job, o32, o64, st, e64, e32, est = _both(native, oracle, scene, sensor, **kw)
    assert est[1] == st.segments and np.array_equal(e32, o32)
    if kw.get("max_depth") == 1:
        assert st.segments == 0 and o32[..., :3].max() > 0           # only directly visible emitters
        assert st.shadow_rays == 0


def test_ragged_film_crop_box_filter(native, oracle):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(77, 45, 3, device=-1, seed=99, rfilter="box",
                                       crop_offset_x=5, crop_offset_y=3, crop_width=50, crop_height=37)
    job, o32, o64, st, e64, e32, est = _both(native, oracle, scene, sensor)
    assert st.samples == 50 * 37 * 3 and np.array_equal(e32, o32)
    assert np.array_equal(e64.astype(np.float32), o64.astype(np.float32))
    assert np.allclose(o32[..., 4], 3.0)                             # box filter: every pixel owns exactly its spp


def test_seed_changes_image_and_is_reproducible(native, oracle):
    from mitsuba2_amd import scenes
    films = []
    for seed in (0, 0, 1):
        scene, sensor = scenes.cornell_box(32, 32, 2, device=-1, seed=seed)
        films.append(oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=3)[0])
    assert np.array_equal(films[0], films[1]) and not np.array_equal(films[0], films[2])


def test_oracle_is_thread_count_invariant(native, oracle):
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(96, 64, 2, device=-1)
    job = native.PathIntegrator().render_job(sensor)
    a = oracle.render(scene.desc(), job, threads=1)[0]
    b = oracle.render(scene.desc(), job, threads=7)[0]
    assert np.array_equal(a, b)


def test_image_mean_sanity(native, oracle):
    """In the spirit of src/python/python/test/scenes.py:261-286: a closed-form-ish sanity bound, not a golden."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 64, 16, device=-1)
    o32 = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=8)[0]
    rgb_over_w = o32[..., :3] / o32[..., 4:5]
    assert 0.05 < rgb_over_w.mean() < 2.0 and (o32[..., 3] / o32[..., 4]).mean() > 0.9


# ---- golden fixtures (tests/golden/make_golden.py wrote them from the oracle) ---------------------------
def test_golden_film_fixture(native, oracle):
    g = np.load(os.path.join(GOLDEN, "cornell_48x32_4spp.npz"))
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(48, 32, 4, device=-1)
    o32, _, st = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=2, want_f64=False)
    assert np.array_equal(o32, g["film"]) and st.segments == int(g["segments"])
    scene, sensor = scenes.cornell_box(32, 24, 4, diffuse_only=False, device=-1, ball_level=1)
    o32, _, st = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=2, want_f64=False)
    assert np.array_equal(o32, g["film_materials"]) and st.segments == int(g["segments_materials"])


def _round1_plugin_jobs(native):
    """the three jobs of tests/golden/round1_plugins.npz -> [(key, scene, job)]"""
    from mitsuba2_amd import scenes
    from test_textures import textured_quad, quad_sensor, checker
    scene, sensor = scenes.cornell_box(32, 24, 4, diffuse_only=False, device=-1, ball_level=1)
    tscene = native.Scene(textured_quad(native, native.BitmapTexture(checker(), wrap_mode="mirror"))).build(-1)
    return [("direct", scene, native.DirectIntegrator(emitter_samples=2, bsdf_samples=1).render_job(sensor)),
            ("squares", scene, native.MomentIntegrator(native.PathIntegrator(max_depth=3)).render_job(sensor, moment_pass=2)),
            ("textured", tscene, native.PathIntegrator(max_depth=4).render_job(quad_sensor(native, 32, 24, 4)))]


def test_golden_fixture_of_the_round1_plugins(native, oracle):
    """direct integrator, moment squares pass, texture coordinates + bitmap texture: committed films"""
    g = np.load(os.path.join(GOLDEN, "round1_plugins.npz"))
    for key, scene, job in _round1_plugin_jobs(native):
        o32, _, st = oracle.render(scene.desc(), job, threads=2, want_f64=False)
        assert np.array_equal(o32, g["film_" + key]) and st.segments == int(g["segments_" + key]), key


# ---- host logic --------------------------------------------------------------------------------------------
def test_render_cfg_block_size_and_seeds(native):
    from mitsuba2_amd import scenes
    sensor = scenes.cornell_sensor(256, 256, 64, seed=7)
    job = native.PathIntegrator().render_job(sensor, n_threads=8)
    assert job.cfg.block_size == 32 and job.cfg.block_count == 64 and job.cfg.spp == 64 and job.cfg.base_seed == 7
    assert job.cfg.max_depth == -1 and job.cfg.rr_depth == 5       # integrator.cpp:305-314
    job = native.PathIntegrator().render_job(sensor, n_threads=256)
    assert job.cfg.block_size == 16                                 # halved until #blocks >= #threads (integrator.cpp:88-97)
    job = native.PathIntegrator(block_size=20).render_job(sensor)
    assert job.cfg.block_size == 32                                 # rounded up to a power of two (:27-32)
    ids = job.block_ids[:job.cfg.block_count]
    assert sorted(ids) == list(range(64)) and ids[4 * 8 + 4] == 0   # the spiral starts at block (bx/2, by/2)
    sensor = scenes.cornell_sensor(1920, 1080, 512)
    job = native.PathIntegrator().render_job(sensor, n_threads=8)
    assert job.cfg.block_count == 60 * 34 == 2040


def test_shard_tiles_partition_the_blocks(native):
    from mitsuba2_amd import scenes, dist
    sensor = scenes.cornell_sensor(200, 120, 1)
    seen = []
    for r in range(3):
        integ = native.PathIntegrator(); integ.set_shard(r, 3)
        job = integ.render_job(sensor)
        tiles = list(job.tiles[:job.cfg.tile_count])
        ids = [int(job.block_ids[t]) for t in tiles]
        assert ids == dist.shard_blocks(job.cfg.block_count, r, 3)
        seen += tiles
    assert sorted(seen) == list(range(7 * 4))


def test_property_errors(native):
    with pytest.raises(RuntimeError, match="rr_depth"):
        native.PathIntegrator(rr_depth=0)
    with pytest.raises(RuntimeError, match="max_depth"):
        native.PathIntegrator(max_depth=-2)
    with pytest.raises(RuntimeError, match="not found"):
        native.BSDF("blendbsdf")
    with pytest.raises(RuntimeError, match="invalid distribution"):
        native.BSDF("roughconductor", distribution="phong")
    with pytest.raises(RuntimeError, match="alpha_u"):
        native.BSDF("roughconductor", alpha_u=0.1)
    with pytest.raises(RuntimeError, match="range"):
        native.BSDF("diffuse", reflectance=(1.5, 0.2, 0.2))
    with pytest.raises(RuntimeError, match="crop"):
        native.Film(width=10, height=10, crop_width=20)
    r = native.BSDF("roughconductor", distribution="ggx", alpha=0.2, eta=(0.2, 0.9, 1.1), k=(3.9, 2.4, 2.1)).record()
    assert r.type == 2 and r.flags == 3 and np.allclose(list(r.params)[:2], [0.2, 0.2])
    d = native.BSDF("dielectric").record()
    assert np.isclose(d.params[0], 1.5046 / 1.000277)               # bk7 / air defaults (dielectric.cpp:177-180)


def test_film_defaults_and_develop(native):
    from mitsuba2_amd import api
    import ctypes
    f = api.Film()
    s = api.Sensor(f, api.Sampler(), fov=45.0)
    job = native.PathIntegrator().render_job(s)
    assert (job.cfg.crop_w, job.cfg.crop_h) == (768, 576)           # film.cpp:11-14


# ---- the C ABI without a GPU ------------------------------------------------------------------------------------
def test_c_abi_exports_every_declared_symbol(native):
    from mitsuba2_amd import _capi
    lib = C.CDLL(os.path.join(_capi.LIB_DIR, "libmiwave.so"))
    header = open(os.path.join(ROOT, "include", "miwave.h")).read()
    import re
    declared = set(re.findall(r"\b(mi_[a-z_]+)\s*\(", header))
    assert declared == set(_capi.MI_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_no_gpu_means_loud_failure(native):
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    from mitsuba2_amd import api, scenes
    with pytest.raises(RuntimeError, match="mi_create failed"):
        api.Device(0)
    with pytest.raises(RuntimeError):
        scenes.cornell_box(16, 16, 1, device=0)
    scene, sensor = scenes.cornell_box(16, 16, 1, device=-1)
    with pytest.raises(RuntimeError, match="no device context"):
        api.PathIntegrator().render(scene, sensor)


def test_package_does_not_touch_the_oracle():
    """The product must never import / link / exec anything under oracle/."""
    pkg = os.path.join(ROOT, "mitsuba2_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f == "build.py":            # the build recipe may COMPILE the checker (building is not using)
                continue
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle_py" not in text and "libmiw_oracle" not in text and "../oracle" not in text, f
    out = subprocess.run(["ldd", os.path.join(pkg, "lib", "libmiwave_host.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


# ---- world_size 2 over gloo: tile shards + one film reduce == the single-process film ------------------------------
_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
from mitsuba2_amd import api, scenes, dist
import oracle_py
rank, world, _ = dist.init("gloo")
scene, sensor = scenes.cornell_box(96, 64, 3, device=-1)
integ = api.PathIntegrator(); integ.set_shard(rank, world)
job = integ.render_job(sensor)
e64, e32, stats = oracle_py.load().emu_render(scene.desc(), job)     # CPU stand-in for mi_render on this rank's shard
film = torch.from_numpy(e64.copy())
dist.reduce_film(film)
t = dist.max_over_ranks(float(rank))
if rank == 0:
    np.save(%(out)r, film.numpy()); assert t == world - 1
dist.finalize()
'''


@pytest.mark.parametrize("world", [2, 3])        # 3: the 6 spiral blocks of the 96 x 64 job split 2 / 2 / 2 round-robin, an odd rank count
def test_gloo_world_size_2_sharded_render(native, oracle, tmp_path, world):
    from mitsuba2_amd import scenes
    out = str(tmp_path / "film.npy")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT, out=out))
    import socket
    with socket.socket() as so:                                  # a free port, as bench.spawn_command picks one (a fixed one collides with a parallel run)
        so.bind(("127.0.0.1", 0)); port = str(so.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr",
                        "127.0.0.1", "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    scene, sensor = scenes.cornell_box(96, 64, 3, device=-1)
    full = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=4)[1]
    got = np.load(out)
    assert np.array_equal(got.astype(np.float32), full.astype(np.float32))


def test_accept_rule_is_a_build_switch_and_silent_on_the_bench_scenes(native, oracle, tmp_path):
    """The bounding-box accept rule of triangle hits (miw/shape.h; a documented departure from mesh.h:194-226) can be
    compiled out with -DMIW_ACCEPT_RULE=0. With well-conditioned rays it never fires: the checker built without it gives
    the same hits on 60 000 camera + bounce rays and the same films on the benchmark scene classes."""
    import ctypes as C
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from mitsuba2_amd import scenes, build
    so = str(tmp_path / "libmiw_oracle_norule.so")
    subprocess.check_call([build.CXX] + build.CXX_FLAGS + ["-DMIW_ACCEPT_RULE=0", os.path.join(ROOT, "oracle", "miw_oracle.cpp"),
                                                           os.path.join(ROOT, "oracle", "wavefront_emu.cpp"), "-o", so, "-lpthread"])
    bare = oracle_py.Oracle(C.CDLL(so), 3)
    for scene, sensor in (scenes.cornell_box(64, 48, 8, device=-1), scenes.cornell_box(64, 48, 4, diffuse_only=False, device=-1, ball_level=2)):
        job = native.PathIntegrator().render_job(sensor)
        a, _, sa = oracle.render(scene.desc(), job, threads=4, want_f64=False)
        b, _, sb = bare.render(scene.desc(), job, threads=4, want_f64=False)
        assert sa.segments == sb.segments and np.array_equal(a, b)
        rng = np.random.default_rng(12)
        rays = np.array([sensor.sample_ray(x, y) for x, y in rng.uniform(0, 1, (20000, 2)).astype(np.float32)])
        h = oracle.trace(scene.desc(), rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7])
        hit = np.isfinite(h["t"])
        p = rays[hit, 0:3] + rays[hit, 3:6] * h["t"][hit, None]
        d2 = rng.normal(size=p.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        o = np.concatenate([rays[:, 0:3], p]); d = np.concatenate([rays[:, 3:6], d2])
        mint = np.concatenate([rays[:, 6], np.full(len(p), 1e-2, np.float32)]); maxt = np.concatenate([rays[:, 7], np.full(len(p), np.inf, np.float32)])
        x, y = oracle.trace(scene.desc(), o, d, mint, maxt), bare.trace(scene.desc(), o, d, mint, maxt)
        assert np.array_equal(x["prim"], y["prim"]) and np.array_equal(x["t"].view(np.uint32), y["t"].view(np.uint32))


@pytest.mark.parametrize("rfilter", ["gaussian", "box", "lanczos"])
def test_tiny_blocks_of_a_many_threaded_render(native, oracle, rfilter):
    """integrator.cpp:88-97 halves the block size until there are as many blocks as worker threads — down to 1 pixel for
    a small window. A film texel then lies under the borders of up to (2 * border + 1)^2 blocks, not 2 x 2: the ordered
    film merge (miw/film_gather.h, the device's k_film_merge) must still add every covering block, in ascending id."""
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(192, 108, 4, device=-1)
    sensor = scenes.cornell_sensor(192, 108, 4, rfilter=rfilter, crop_offset_x=90, crop_offset_y=60, crop_width=16, crop_height=16)
    sizes = []
    for n_threads in (1, 16, 64, 256):
        job = native.PathIntegrator().render_job(sensor, n_threads=n_threads)
        sizes.append((job.cfg.block_size, job.cfg.block_count))
        o32, o64, st = oracle.render(scene.desc(), job, threads=4)
        e64, e32, est = oracle.emu_render(scene.desc(), job)
        assert st.samples == 16 * 16 * 4 and est[1] == st.segments
        assert np.array_equal(e32, o32) and np.array_equal(e64.astype(np.float32), o64.astype(np.float32))
    assert sizes == [(32, 1), (4, 16), (2, 64), (1, 256)]


@pytest.mark.parametrize("shape", [(150, 83, 32), (16, 16, 2), (700, 40, 8)])
def test_tile_interleaved_sample_log_is_the_same_film(native, oracle, monkeypatch, shape):
    """The device writes the 16-byte sample log interleaved over groups of 64 tiles when k_film_lanes will replay it (miw/film.h:
    log_index, [tile / 64][pixel][sample][tile % 64], padded to whole groups). The CPU twin of the lane stages writes and replays
    the same layout through the same header (MIW_EMU_LOG_IL=1): log_index is a bijection into log_capacity and the film is the
    [lane][sample] film and the oracle's — with fewer than 64 tiles, with exactly one tile per pixel, with more than one group."""
    from mitsuba2_amd import scenes
    w, h, bs = shape
    scene, sensor = scenes.cornell_box(w, h, 3, device=-1, seed=5)
    job = native.PathIntegrator().render_job(sensor, n_threads=1 if bs == 32 else (64 if bs == 2 else 400))
    assert job.cfg.block_size == bs, job.cfg.block_size
    o32, o64, st = oracle.render(scene.desc(), job, threads=4)
    monkeypatch.delenv("MIW_EMU_LOG_IL", raising=False)
    a64, a32, ast = oracle.emu_render(scene.desc(), job)
    monkeypatch.setenv("MIW_EMU_LOG_IL", "1")
    b64, b32, bst = oracle.emu_render(scene.desc(), job)
    assert np.array_equal(a32, o32) and np.array_equal(b32, o32) and list(ast) == list(bst)


def test_more_ranks_than_blocks_leaves_empty_shards_empty(native, oracle):
    """A 64x48 frame has 4 spiral blocks; with 8 ranks, ranks 4..7 hold no block. Their job must say so (non-NULL
    tile_list, tile_count == 0 — a NULL list means "all blocks" to mi_render) and render nothing, so that the film
    reduce still adds up to the 1-rank film."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(64, 48, 2, device=-1)
    full32, full64, fst = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=4)
    acc = np.zeros_like(full64); samples = 0
    for rank in range(8):
        integ = native.PathIntegrator(); integ.set_shard(rank, 8)
        job = integ.render_job(sensor)
        assert job.cfg.block_count == 4 and job.cfg.tile_count == (1 if rank < 4 else 0) and bool(job.cfg.tile_list)
        _, p64, st = oracle.render(scene.desc(), job, threads=2)
        e64, _, est = oracle.emu_render(scene.desc(), job)
        assert st.samples == (fst.samples // 4 if rank < 4 else 0) or rank < 4      # clipped edge blocks differ in size
        if rank >= 4:
            assert st.samples == 0 and est[0] == 0 and not p64.any() and not e64.any()
        acc += p64; samples += st.samples
    assert samples == fst.samples and np.array_equal(acc.astype(np.float32), full64.astype(np.float32))


def test_shard_mode_selection_and_pass_jobs(native):
    """mitsuba2_amd/dist.py: what bench.py --shard auto picks, and the per-rank jobs of the pass partition"""
    from mitsuba2_amd import dist as mdist, scenes
    pick = mdist.choose_shard
    # the headline partition is the north star's: tiles, at every rank count (the N-GPU film == the 1-GPU film)
    assert [pick("auto", n, 1920, 1080, 512) for n in (1, 2, 3, 4, 8)] == ["tiles"] * 5
    assert [pick("auto", n, 3840, 2160, 512) for n in (2, 4, 8)] == ["tiles"] * 3
    assert pick("tiles", 8, 1920, 1080, 512) == "tiles" and pick("passes", 2, 7680, 4320, 512) == "passes"
    assert pick("passes", 8, 1920, 1080, 100) == "tiles"                                             # spp not divisible
    with pytest.raises(ValueError):
        pick("rows", 2, 64, 64, 4)
    scene, sensor = scenes.cornell_box(64, 48, 8, device=-1)
    one = native.PathIntegrator().render_job(sensor)
    for make in (native.PathIntegrator, native.DirectIntegrator):
        for rank in range(4):
            integ, job = mdist.pass_job(make, sensor, rank, 4, 8)
            c = job.cfg
            assert (c.spp, c.accumulate, c.tile_count, c.block_count) == (2, 0, 0, one.cfg.block_count)
            assert np.array_equal(job.block_ids[:c.block_count], one.block_ids[:c.block_count] + (3 - rank) * c.block_count)   # spiral.cpp:41


def test_golden_digests_of_round2(native, oracle):
    """tests/golden/round2.json: film digests + segment counts of the scene classes and plugins the .npz fixtures do not cover
    and of twelve fuzz recipes (tests/golden/make_golden_r2.py). The oracle and the device share their leaf headers: an
    accidental leaf edit moves both, and only a committed answer notices."""
    import json
    sys.path.insert(0, GOLDEN)
    import make_golden_r2 as mk
    from mitsuba2_amd import scenes
    want = json.load(open(os.path.join(GOLDEN, "round2.json")))
    got = mk.compute(native, scenes, oracle, mk.rgb_cases(native, scenes))
    assert len(got) == 21
    for key, rec in got.items():
        assert rec["sha256"] == want[key]["sha256"] and rec["segments"] == want[key]["segments"], (key, rec, want[key])


def test_golden_digest_of_the_spectral_glass_block(spectral, oracle_spectral):
    import json
    sys.path.insert(0, GOLDEN)
    import make_golden_r2 as mk
    from mitsuba2_amd import scenes
    want = json.load(open(os.path.join(GOLDEN, "round2.json")))
    got = mk.compute(spectral, scenes, oracle_spectral, mk.spectral_cases(spectral, scenes))
    for key, rec in got.items():
        assert rec["sha256"] == want[key]["sha256"] and rec["segments"] == want[key]["segments"], (key, rec, want[key])


def test_oracle_render_in_chunks_of_blocks_is_the_one_call_render(native, oracle):
    """tests/golden/make_golden_r5.py renders the 4.25e9-sample frame of config 4 in checkpointed chunks of consecutive spiral ids:
    every chunk one orc_render(only_blocks = ids, accumulate = 1) onto the film of the chunks before it. The blocks are merged in
    ascending id inside a call (miw_oracle.cpp: film_put in job order), so the chunked film goes through exactly the float32
    additions of the one-call film — bit for bit, with the same sample / segment / shadow-ray totals."""
    from mitsuba2_amd import scenes
    scene, sensor = scenes.cornell_box(200, 120, 3, diffuse_only=False, ball_level=1, device=-1)
    job = native.PathIntegrator().render_job(sensor)
    n = int(job.cfg.block_count)
    one, _, st1 = oracle.render(scene.desc(), job, threads=4, want_f64=False)
    job.cfg.accumulate = 1
    film = np.zeros_like(one); seg = sam = sha = 0
    for lo in range(0, n, 5):
        ids = np.arange(lo, min(lo + 5, n), dtype=np.uint32)
        film, _, st = oracle.render(scene.desc(), job, threads=3, want_f64=False, only_blocks=ids, onto=(film, None))
        film = np.array(film); seg += st.segments; sam += st.samples; sha += st.shadow_rays
    assert np.array_equal(film, one) and (sam, seg, sha) == (st1.samples, st1.segments, st1.shadow_rays)
